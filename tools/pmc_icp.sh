#!/bin/bash
# SQ counter passes of the frame -> pose pipeline (tools/pipeline_timing.py) for the ICP kernels: rocprofv3 --pmc, one pass per
# counter set, never combined with traces.   gpurun --timeout 900 -- bash tools/pmc_icp.sh gpurun_out/pmc_icp
set -e
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -o "$name" -- python3 "$ROOT/tools/pipeline_timing.py" > "$ROOT/$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
echo done
