#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_e9
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
NO_PMC=1 tools/vote_variants.sh $OUT product rs512 rs640
BENCH_ARGS="--config c4 --cells 32" NO_PMC=1 tools/vote_variants.sh $OUT/c4 product rs512
