cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_bench; mkdir -p $OUT
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench rc=$?"; tail -3 $OUT/bench_c2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench/bench_c2.json').read().strip().splitlines()[-1])
print('c2', d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'], d['roofline']['traffic_stale_reason'])
print(json.dumps(d['other_configs']['c1_pipeline'], indent=1))
PY
