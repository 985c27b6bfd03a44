#!/bin/bash
# Time and count k_vote in several builds of the library (run on the GPU box through gpurun).
#   tools/vote_variants.sh OUTDIR name[=lib.so] ...      name "product" = the in-tree library, otherwise build_var/<name>.so
# Per build: one bench line (HIP-event kernel times) and one rocprofv3 --pmc pass over the LDS / wait counters.
# BENCH_ARGS selects the workload (default C2).  The program after `--` is python3 itself.
set -e
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
for v in "$@"; do
  name=${v%%=*}
  if [ "$name" = product ]; then lib=$ROOT/yolo_ppf_pose_estimation_amd/csrc/libppf_hip.so; else lib=$ROOT/build_var/$name.so; fi
  export PPF_HIP_LIB=$lib
  cd "$ROOT"
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs $BENCH_ARGS > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "$name: bench failed"; tail -5 "$OUT/$name.err"; exit 1; }
  python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d.get("kernel_ms", {})
print(sys.argv[2], "step %.3f ms" % d["ms_per_step"], {a: round(b, 3) for a, b in k.items()}, flush=True)
PY
  if [ -z "$NO_PMC" ]; then
    cd /tmp && export TMPDIR=/tmp
    timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS \
      --output-format csv -d "$ROOT/$OUT/pmc_$name" -o "$name" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs $BENCH_ARGS > "$ROOT/$OUT/pmc_$name.log" 2>&1 || { echo "$name: pmc failed"; tail -5 "$ROOT/$OUT/pmc_$name.log"; exit 1; }
  fi
done
echo done
