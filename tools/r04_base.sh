cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_base; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_cpp_facade.py -m gpu -x -q > $OUT/pytest_facade.log 2>&1; echo "pytest facade rc=$?"; tail -3 $OUT/pytest_facade.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench rc=$?"
timeout -k 10 200 python tools/icp_timing.py > $OUT/icp_timing.json 2>$OUT/icp_timing.err; echo "icp rc=$?"; cat $OUT/icp_timing.json
timeout -k 10 200 python tools/pipeline_timing.py > $OUT/pipeline_timing.json 2>$OUT/pipeline_timing.err; echo "pipe rc=$?"; cat $OUT/pipeline_timing.json
cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/icp_prof -o icp -- python3 $GRAFT_REPO_ROOT/tools/icp_timing.py > $GRAFT_REPO_ROOT/$OUT/icp_prof.log 2>&1; echo "icp prof rc=$?"
