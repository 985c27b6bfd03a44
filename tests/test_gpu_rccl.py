"""RCCL on the path's only collective, exercised on the one GPU a test box has: a fresh child process forms a process
group of ONE rank over backend "nccl" (= RCCL on ROCm) and all_gathers the engine's device-side result blocks straight
from HBM -- the top-5 clustered poses (C3 / C5) and the per-reference pose block (C4 sharded over reference points) --
then checks the bytes and tears the group down.  Multi-rank behaviour is covered on the CPU with gloo
(tests/test_parallel_gloo.py); this test makes sure that the first RCCL call this code ever issues is not the driver's
8-GPU run."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import datetime, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["PPF_ROOT"])
from yolo_ppf_pose_estimation_amd import parallel, synth, workloads as W
from yolo_ppf_pose_estimation_amd.detector import PPF3DDetector
from yolo_ppf_pose_estimation_amd.device import Workspace

torch.cuda.set_device(0)
dist.init_process_group("nccl", world_size=1, rank=0, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=120))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
bottle = W.bottle()
det = PPF3DDetector(0.06, 0.05).trainModel(bottle)
assert det.device() == torch.cuda.current_device() == 0
scene, _ = synth.make_scene(bottle, n_points=6000, seed=3)
d_scene = torch.from_numpy(scene).cuda()
ws = Workspace()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ws.match_device(det, d_scene.data_ptr(), scene.shape[0], 6, 1.0 / 20.0, 0.05, presampled=True, stream=s.cuda_stream)
    st = ws.stats()
    top = ws.device_top_block(W.TOP_K, s.cuda_stream)
    raw = ws.device_pose_block(st["n_ref"] + 3, s.cuda_stream)
    # the collective itself, on the engine's blocks, device to device
    out_top = torch.empty_like(top)
    dist.all_gather_into_tensor(out_top, top)
    out_raw = parallel.gather_device(raw, dist, force=True)
s.synchronize()
torch.cuda.synchronize()
assert out_top.is_cuda and out_raw.is_cuda
assert torch.equal(out_top, top) and torch.equal(out_raw, raw)
res = ws.results(st["n_ref"])
host_top = np.stack([p.pose for p in res["poses"][: W.TOP_K]])
np.testing.assert_array_equal(out_top.cpu().numpy()[: len(host_top), :16].reshape(-1, 4, 4), host_top)
host_raw = np.stack([p.pose for p in res["raw_poses"]])
np.testing.assert_array_equal(out_raw.cpu().numpy()[: st["n_ref"], :16].reshape(-1, 4, 4), host_raw)
assert (out_raw.cpu().numpy()[st["n_ref"]:] == 0).all()
# the max-over-ranks / sum-over-ranks reductions bench.py makes
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.5
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK", st["n_ref"], int(res["poses"][0].numVotes))
"""


def test_rccl_all_gather_of_the_device_result_blocks_world_of_one():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               PPF_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=420)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "RCCL_OK" in r.stdout
