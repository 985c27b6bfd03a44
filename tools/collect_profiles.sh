#!/bin/bash
# Everything the round's profiles/ directory is made from, in two gpurun calls (about 6 + 7 minutes on the box):
#   tools/collect_profiles.sh gpurun_out/r03_final lines    bench lines (c2 with host entry, C4 / C5 legs and the CPU baseline; c4; c5),
#                                                           2-rank rehearsals, rocprofv3 kernel traces, the attribution builds
#   tools/collect_profiles.sh gpurun_out/r03_final pmc      PMC passes of c2, c4, c5 (never combined with traces)
#   tools/collect_profiles.sh gpurun_out/r03_final benchlines   only the bench lines and rehearsals again (publish with --lines)
# then, in the repo:  python tools/publish_profiles.py gpurun_out/r03_final r03
# The attribution builds (build_var/a_*.so) are made beforehand with tools/build_attribution.sh.
set -e
OUT=$1
WHAT=${2:-lines}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd "$ROOT"
if [ "$WHAT" = lines ] || [ "$WHAT" = benchlines ]; then
  python bench.py --steps 20 --warmup 5 > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
  python bench.py --config c4 --cells 32 --steps 3 --warmup 1 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
  python bench.py --config c5 --steps 3 --warmup 1 > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
  echo "bench lines done"
  ( export PPF_BENCH_ONE_DEVICE=1 PPF_BENCH_BACKEND=gloo
    python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline > "$OUT/bench_c3_2ranks_one_device_gloo.json" 2> "$OUT/bench_n2.err"
    python bench.py --gpus 2 --config c4 --shard refs --steps 2 --warmup 1 > "$OUT/bench_c4_refs_2ranks_one_device_gloo.json" 2>> "$OUT/bench_n2.err"
    python bench.py --gpus 2 --config c5 --steps 3 --warmup 1 > "$OUT/bench_c5_2ranks_one_device_gloo.json" 2>> "$OUT/bench_n2.err" )
  echo "2-rank rehearsals done"
  [ "$WHAT" = benchlines ] && exit 0   # after publishing the PMC summaries: the lines again, so that their roofline objects read them
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/trace_c2" -o c2 -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > "$ROOT/$OUT/trace_c2.log" 2>&1
    rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/trace_c4" -o c4 -- python3 "$ROOT/bench.py" --config c4 --cells 32 --steps 3 --warmup 1 > "$ROOT/$OUT/trace_c4.log" 2>&1
    rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/trace_c5" -o c5 -- python3 "$ROOT/bench.py" --config c5 --steps 3 --warmup 1 > "$ROOT/$OUT/trace_c5.log" 2>&1 )
  echo "traces done"
  # the reference's real call and its ICP: timings and the ICP kernels' trace
  python tools/pipeline_timing.py > "$OUT/pipeline_timing.json" 2> "$OUT/pipeline_timing.err"
  python tools/icp_timing.py --repeat 5 > "$OUT/icp_timing.json" 2> "$OUT/icp_timing.err"
  python tools/icp_timing.py --repeat 5 --legacy > "$OUT/icp_timing_legacy.json" 2>> "$OUT/icp_timing.err"
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace_icp" -o icp -- python3 "$ROOT/tools/icp_timing.py" --repeat 3 > "$ROOT/$OUT/trace_icp.log" 2>&1 )
  echo "pipeline and ICP done"
  if [ -f build_var/a_none.so ]; then
    tools/vote_variants.sh "$OUT/classes" product a_counted0 a_owncell0 a_dsmall0 a_dbig0 a_dbig2 a_aggonly a_directonly a_none
  fi
else
  tools/pmc_vote.sh "$OUT/pmc_c2"
  echo "pmc c2 done"
  BENCH_ARGS="--config c4 --cells 32" tools/pmc_vote.sh "$OUT/pmc_c4"   # c4 overflows 16-bit cells: its steady state is 32-bit cells from the first call
  echo "pmc c4 done"
  # one lane: the counters do not depend on how the matches overlap, and bench.py then runs no extra single-lane step (the
  # summary divides by the 3 steps of the run)
  PPF_BATCH_LANES=1 BENCH_ARGS="--config c5" tools/pmc_vote.sh "$OUT/pmc_c5"
  echo "pmc c5 done"
fi
