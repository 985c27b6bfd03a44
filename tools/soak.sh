#!/bin/bash
# Soak run of the seeded sweeps against the oracle with many more draws than the suite's:
#   gpurun --timeout 1150 -- bash tools/soak.sh [match draws] [icp draws] [prep draws] [policy draws] [degenerate draws] [batch draws]
# One pytest process per sweep, logs under gpurun_out/soak/ (publish the summary lines in profiles/rNN_soak.md).
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/soak; mkdir -p $OUT
# PPF_SOAK_OFFSET=N in the environment starts the extra draws at seed N (fresh seeds for a further run)
export PPF_SOAK_MATCH=${1:-300} PPF_SOAK_ICP=${2:-150} PPF_SOAK_PREP=${3:-60} PPF_SOAK_POLICY=${4:-100} PPF_SOAK_DEGENERATE=${5:-100} PPF_SOAK_BATCH=${6:-40}
timeout -k 10 650 python -m pytest tests/test_gpu_random_sweep.py -m gpu -q -p no:cacheprovider > $OUT/match.log 2>&1; echo "match rc=$? $(tail -1 $OUT/match.log)"
timeout -k 10 350 python -m pytest tests/test_gpu_icp.py -k random_draw -m gpu -q -p no:cacheprovider > $OUT/icp.log 2>&1; echo "icp rc=$? $(tail -1 $OUT/icp.log)"
timeout -k 10 250 python -m pytest tests/test_gpu_prep.py -k random_draw -m gpu -q -p no:cacheprovider > $OUT/prep.log 2>&1; echo "prep rc=$? $(tail -1 $OUT/prep.log)"
