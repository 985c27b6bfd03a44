#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_e3
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_configs.py -m gpu -x -q -s > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?"; grep -c "fixtures\]" $OUT/pytest_new.log; tail -4 $OUT/pytest_new.log
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench rc=$?"; tail -3 $OUT/bench_full.err
tools/vote_variants.sh $OUT q0_b32 a_counted0 a_owncell0 a_build2 a_dsmall0 a_dbig0 a_dbig2 a_aggonly a_directonly a_none
