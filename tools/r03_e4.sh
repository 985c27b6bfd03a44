#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_e4
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --pipeline-depth 2 > $OUT/bench_depth2.json 2> $OUT/bench_depth2.err; echo "depth2 rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_e4/bench_depth2.json')); print('depth2 step', d['ms_per_step'], d['kernel_ms'])
PY
