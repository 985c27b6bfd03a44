#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_e6
mkdir -p $OUT
tools/vote_variants.sh $OUT g128 g256 g512 g1024 g256_b64
python tools/pmc_table.py $OUT/pmc_*
