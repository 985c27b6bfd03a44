#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_final; mkdir -p $OUT
python bench.py --steps 20 --warmup 5 > $OUT/bench_c2_check.json 2> $OUT/bench_c2_check.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_final/bench_c2_check.json').read().strip().splitlines()[-1])
print('c2', d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'], d['roofline']['traffic_stale'], d['roofline']['traffic_stale_reason'])
print('host', d['host_entry']['ms_per_match'], 'pipelined', d['pipelined']['ms_per_step'])
for k,v in d['other_configs'].items(): print(k, v['ms_per_step'], v['kernel_ms'].get('k_vote'), v['roofline']['frac'], v['roofline']['traffic_stale'], v['roofline']['traffic_stale_reason'])
PY
