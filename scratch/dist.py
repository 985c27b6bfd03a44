import sys; sys.path.insert(0,'.')
import numpy as np, torch
from yolo_ppf_pose_estimation_amd import synth
from yolo_ppf_pose_estimation_amd.detector import PPF3DDetector
from yolo_ppf_pose_estimation_amd.device import Workspace
bottle = np.load('tests/golden/bottle_model_xyzn.npy')
det = PPF3DDetector(0.036, 0.05).trainModel(bottle)
scene, _ = synth.make_scene(bottle, n_points=50000, seed=12345)
ws = Workspace(timing=True)
d = torch.from_numpy(scene).cuda()
ws.match_device(det, d.data_ptr(), 50000, 6, 1/20., 0.05)
res = ws.results(2500)
v, p = ws.ref_counters(2500)
v = v.astype(np.float64)
print('total', v.sum(), 'mean', v.mean(), 'max', v.max(), 'p50', np.median(v), 'p90', np.percentile(v,90), 'p99', np.percentile(v,99))
print('top10', np.sort(v)[-10:])
print('max/mean', v.max()/v.mean(), ' sum/256', v.sum()/256, ' => tail bound ratio max/(sum/256)=', v.max()/(v.sum()/256))
print(res['stats'])
