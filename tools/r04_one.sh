cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_one
timeout -k 10 600 python -m pytest tests/test_gpu_icp.py -m gpu -x -q > gpurun_out/r04_one/pytest.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/r04_one/pytest.log
