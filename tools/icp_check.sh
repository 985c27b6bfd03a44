# ICP: parity tests, C2 top-5 timing, frame -> pose, the tail-phase clocks build (build_var/icpclk.so, -DPPF_ICP_CLOCKS) and a kernel trace:  gpurun --timeout 1100 -- bash tools/icp_check.sh
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/icp_check; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_robustness.py::test_icp_schedules_agree tests/test_c1_pipeline_golden.py tests/test_gpu_prep.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
timeout -k 10 200 python tools/icp_timing.py --repeat 5 > $OUT/icp_timing.json 2>$OUT/icp_timing.err; echo "icp rc=$?"; cat $OUT/icp_timing.json
timeout -k 10 200 python tools/pipeline_timing.py > $OUT/pipeline_timing.json 2>$OUT/pipeline_timing.err; echo "pipe rc=$?"; cat $OUT/pipeline_timing.json
if [ -f build_var/icpclk.so ]; then PPF_HIP_LIB=$PWD/build_var/icpclk.so timeout -k 10 200 python tools/icp_timing.py --repeat 1 > $OUT/icp_clk.json 2>$OUT/icp_clk.err; tail -5 $OUT/icp_clk.err; fi
cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/icp_prof -o icp -- python3 $GRAFT_REPO_ROOT/tools/icp_timing.py > $GRAFT_REPO_ROOT/$OUT/icp_prof.log 2>&1; echo "icp prof rc=$?"
