"""N>1 path on CPU: world_size-2 gloo run of the sharding + pose-gather glue (the GPU box runs the same
code over RCCL).  No compute here: the engine has no CPU path; pose blocks are synthetic records."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from yolo_ppf_pose_estimation_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharding_covers_everything_once():
    for world in (1, 2, 3, 8):
        crops = sorted(c for r in range(world) for c in parallel.shard_crops(64, r, world))
        assert crops == list(range(64))
    shards = [np.arange(r, 10, 3)[:, None].astype(np.float64) for r in range(3)]
    np.testing.assert_array_equal(parallel.merge_reference_shards(shards)[:, 0], np.arange(10))
    assert parallel.POSE_WORDS == 27


def test_pose_block_roundtrip():
    from yolo_ppf_pose_estimation_amd._capi import Pose
    recs = (Pose * 4)()
    for i in range(3):
        recs[i].num_votes = 100 - i
        recs[i].pose[5] = 1.5 + i
        recs[i].t[2] = -0.25 * i
    arr = parallel.poses_to_array(recs, 3, 5)
    assert arr.shape == (5, 27) and (arr[3:] == 0).all()
    back = parallel.array_to_poses(arr)
    assert [p.numVotes for p in back] == [100, 99, 98]
    assert back[1].pose[1, 1] == 2.5 and back[2].t[2] == -0.5


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, %r)
    from yolo_ppf_pose_estimation_amd import parallel
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = parallel.shard_crops(5, rank, world)
    local = np.zeros((3, parallel.POSE_WORDS))
    local[:, 0] = [rank * 10 + k for k in range(3)]
    local[:, -1] = 1.0
    allp = parallel.gather_poses(local)
    assert allp.shape == (world, 3, parallel.POSE_WORDS)
    for r in range(world):
        assert list(allp[r, :, 0]) == [r * 10 + k for k in range(3)]
    # reference-point shards of one crop: rank r holds points r, r+world, ...
    per_ref = np.arange(rank, 7, world, dtype=np.float64)[:, None]
    padded = np.full((4, 1), -1.0); padded[: per_ref.shape[0]] = per_ref
    g = parallel.gather_poses(padded)
    merged = parallel.merge_reference_shards([g[r][g[r][:, 0] >= 0] for r in range(world)])
    assert list(merged[:, 0]) == list(range(7))
    # the tensor-block forms bench.py uses (device tensors over RCCL on the GPU box; CPU tensors over gloo here)
    import torch
    blk = torch.zeros((4, parallel.POSE_WORDS), dtype=torch.float64)
    blk[: per_ref.shape[0], 0] = torch.from_numpy(per_ref[:, 0])
    blk[per_ref.shape[0]:, 0] = -1.0
    allb = parallel.gather_device(blk)
    assert tuple(allb.shape) == (world * 4, parallel.POSE_WORDS)
    merged_t = parallel.merge_reference_shards_device(allb, world, 7)
    assert merged_t[:, 0].tolist() == [float(k) for k in range(7)] and merged_t.is_contiguous()
    dist.barrier()
    if rank == 0:
        print("GLOO_OK", world, mine)
    dist.destroy_process_group()
""")


def _free_port() -> str:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "GLOO_OK 2 [0, 2, 4]" in r.stdout
