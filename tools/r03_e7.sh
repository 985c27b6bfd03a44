#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
NO_PMC=1 tools/vote_variants.sh gpurun_out/r03_e7 product t_h16 t_h20 t_h32 t_c1024 t_c4096 t_r16 t_r64 t_q1
