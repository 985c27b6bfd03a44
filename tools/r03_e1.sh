#!/bin/bash
# round 3, first GPU call: baseline suite + bench, and what the LDS counters report for the microbenchmark's known patterns
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_e1
mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/ubench_pmc -o ub -- $ROOT/build_var/ubench_lds_ops > $OUT/ubench_pmc.log 2>&1
echo done
