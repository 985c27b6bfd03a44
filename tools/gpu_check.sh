#!/bin/bash
# One gpurun call that says whether the tree is releasable: the GPU suite, smoke(), and the full bench line with the staleness flags of its
# three counter-based rooflines.   gpurun --timeout 1150 -- tools/gpu_check.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/final; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_final.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_final.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $OUT/bench_c2_check.json 2> $OUT/bench_c2_check.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/final/bench_c2_check.json').read().strip().splitlines()[-1])
print('c2', d['ms_per_step'], d['kernel_ms']['k_vote'], d['roofline']['frac'], d['roofline']['traffic_stale'], d['roofline']['traffic_stale_reason'])
print('host', d['host_entry']['ms_per_match'], 'pipelined', d['pipelined']['ms_per_step'])
for k,v in d['other_configs'].items():
    if 'ms_per_step' in v: print(k, v['ms_per_step'], v['kernel_ms'].get('k_vote'), v['roofline']['frac'], v['roofline']['traffic_stale'], v['roofline']['traffic_stale_reason'])
    else: print(k, 'frame->pose', v.get('frame_to_pose_ms'), 'prep', v.get('prep_ms'), 'match', v.get('match_ms'), 'icp', v.get('icp_ms'), 'golden', v.get('equals_oracle_golden'))
PY
