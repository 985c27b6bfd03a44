cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_icp_var; mkdir -p $OUT
for v in product icpb1 icpb3 icpb4; do
  if [ $v = product ]; then unset PPF_HIP_LIB; else export PPF_HIP_LIB=$PWD/build_var/$v.so; fi
  timeout -k 10 200 python tools/icp_timing.py --repeat 7 > $OUT/icp_$v.json 2>$OUT/icp_$v.err
  timeout -k 10 200 python tools/pipeline_timing.py > $OUT/pipe_$v.json 2>$OUT/pipe_$v.err
  python - $v <<'PY'
import json,sys
v=sys.argv[1]
a=json.load(open(f'gpurun_out/r04_icp_var/icp_{v}.json')); b=json.load(open(f'gpurun_out/r04_icp_var/pipe_{v}.json'))
print(v, 'c2 icp ms %.3f'%(a['gpu_seconds']*1e3), 'c1 frame', min(b['frame_to_pose_ms']), b['of_which_ms'])
PY
done
