#!/bin/bash
# bench every diagnostic build under build_var/ (and the product build) on the default workload; one line each
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
one() {
  local tag=$1 lib=$2
  PPF_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernel_ms',{})
l=d.get('lds_roofline') or {}
print('$tag', 'step %.3f ms' % d['ms_per_step'], {a: round(b,3) for a,b in k.items()}, 'atomics/launch %.3g' % (l.get('lds_atomics_per_launch') or 0), 'frac %.3f' % (l.get('frac_ubench') or 0))
"
}
one product "$ROOT/yolo_ppf_pose_estimation_amd/csrc/libppf_hip.so"
for f in build_var/*.so; do one "$(basename $f .so)" "$ROOT/$f"; done
