cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prep
python tools/prep_timing.py > gpurun_out/prep/prep_timing.json 2> gpurun_out/prep/err.txt; cat gpurun_out/prep/prep_timing.json
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prep/trace -o prep -- python3 $GRAFT_REPO_ROOT/tools/prep_timing.py > $GRAFT_REPO_ROOT/gpurun_out/prep/trace.log 2>&1
