#!/bin/bash
# Everything the round's profiles/ directory is made from, in one gpurun call (about 6 minutes on the box):
#   bench lines (c2 with the CPU baseline, c4, c5, 2-rank rehearsals), rocprofv3 kernel traces, PMC passes.
# Usage (from the repo root, through gpurun):  tools/collect_profiles.sh gpurun_out/r02_final
set -e
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd "$ROOT"
python bench.py --steps 20 --warmup 3 > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
python bench.py --config c4 --steps 4 --warmup 1 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
python bench.py --config c5 --steps 6 --warmup 2 > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
echo "bench lines done"
( export PPF_BENCH_ONE_DEVICE=1 PPF_BENCH_BACKEND=gloo
  python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline > "$OUT/bench_c3_2ranks_one_device_gloo.json" 2> "$OUT/bench_n2.err"
  python bench.py --gpus 2 --config c4 --shard refs --steps 2 --warmup 1 > "$OUT/bench_c4_refs_2ranks_one_device_gloo.json" 2>> "$OUT/bench_n2.err"
  python bench.py --gpus 2 --config c5 --steps 3 --warmup 1 > "$OUT/bench_c5_2ranks_one_device_gloo.json" 2>> "$OUT/bench_n2.err" )
echo "2-rank rehearsals done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/trace_c2" -o c2 -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$ROOT/$OUT/trace_c2.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/trace_c4" -o c4 -- python3 "$ROOT/bench.py" --config c4 --cells 32 --steps 3 --warmup 1 > "$ROOT/$OUT/trace_c4.log" 2>&1
echo "traces done"
cd "$ROOT"
tools/pmc_vote.sh "$OUT/pmc_c2"
BENCH_ARGS="--config c4 --cells 32" tools/pmc_vote.sh "$OUT/pmc_c4"   # c4 overflows 16-bit cells: profile its steady state (32-bit cells from the first call)
echo "pmc done"
