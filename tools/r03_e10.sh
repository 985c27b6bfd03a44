#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
NO_PMC=1 tools/vote_variants.sh gpurun_out/r03_e10 product mlp6 mlp8 mlp12
