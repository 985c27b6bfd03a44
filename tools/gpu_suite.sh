# the whole GPU suite in one process:  gpurun --timeout 1100 -- bash tools/gpu_suite.sh
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/gpu_suite; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log
