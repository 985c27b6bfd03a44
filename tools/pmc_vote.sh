#!/bin/bash
# SQ / LDS / TCC counter passes of the bench step (rocprofv3 --pmc, one pass per counter set; never combined with traces).
#   tools/pmc_vote.sh OUTDIR [bench args...]        (run on the GPU box through gpurun; OUTDIR under gpurun_out/)
# The program after `--` is python3 itself (no env/bash wrappers: the profiler's library initialises the GPU first).
set -e
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -o "$name" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs ${BENCH_ARGS} > "$ROOT/$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
run sq3 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_WAVES SQ_INSTS_FLAT_LDS_ONLY
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
