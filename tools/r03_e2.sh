#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_e2
mkdir -p $ROOT/$OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
tools/vote_variants.sh $OUT q0_b64 q1_b64 q0_b32 q1_b32
