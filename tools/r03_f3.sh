#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_final; mkdir -p $OUT
python bench.py --config c5 --steps 3 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "c5 rc=$?"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --pipeline-depth 2 > $OUT/bench_c2_depth2.json 2> $OUT/bench_c2_depth2.err; echo "depth2 rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_final/bench_c5.json').read().strip().splitlines()[-1]); print('c5', d['ms_per_step'], d['kernel_ms'])
d=json.loads(open('gpurun_out/r03_final/bench_c2_depth2.json').read().strip().splitlines()[-1]); print('depth2', d['ms_per_step'], d['kernel_ms'])
PY
