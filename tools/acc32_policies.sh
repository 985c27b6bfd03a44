# C4 under the four PPF_OPT_ACC32 policies (profiles/r04_c4_acc32_policies.md):  gpurun --timeout 1100 -- bash tools/acc32_policies.sh
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/acc32_policies; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_robustness.py tests/test_gpu_parity.py tests/test_gpu_policy.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for c in auto 32 16 limit; do
  timeout -k 10 300 python bench.py --config c4 --steps 3 --warmup 2 --cells $c --no-cpu-baseline --no-other-configs > $OUT/c4_$c.json 2> $OUT/c4_$c.err; echo "c4 $c rc=$?"
  python - $c <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/acc32_policies/c4_{sys.argv[1]}.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step %.1f'%d['ms_per_step'], d['kernel_ms'], 'acc32 items', d.get('acc32_items_per_step'))
PY
done
