"""One-process-per-GPU glue for the path.

The path shards without any exchange during voting: independent crops go to different ranks, or the
reference points of one crop are strided over ranks (``ref_offset`` / ``ref_stride`` of the C-ABI).
The only collective is the final gather of pose records (RCCL ``all_gather`` on the GPU box, gloo
in the CPU tests).  torch.distributed is plumbing here; nothing in this module computes poses.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from ._capi import Pose

POSE_WORDS = C.sizeof(Pose) // 8  # a ppf_pose record as float64 words (27)


def shard_crops(n_crops: int, rank: int, world: int) -> List[int]:
    """Crop c is matched by rank c mod world (BASELINE configs C3 / C5)."""
    return [c for c in range(n_crops) if c % world == rank]


def shard_reference_points(rank: int, world: int) -> dict:
    """ppf_match_params fields that give this rank every world-th reference point of one crop (C4)."""
    return {"ref_offset": rank, "ref_stride": world}


def poses_to_array(records, count: int, k: int) -> np.ndarray:
    """First `k` ppf_pose records of a ctypes array as a (k, POSE_WORDS) float64 array; rows past `count`
    are zero (num_votes == 0 marks them empty)."""
    out = np.zeros((k, POSE_WORDS), dtype=np.float64)
    n = min(count, k)
    if n:
        out[:n] = np.frombuffer(records, dtype=np.float64)[: n * POSE_WORDS].reshape(n, POSE_WORDS)
    return out


def array_to_poses(arr: np.ndarray):
    """Inverse of poses_to_array: list of Pose3D, empty rows dropped."""
    from .detector import Pose3D
    out = []
    arr = np.ascontiguousarray(arr, dtype=np.float64)
    for row in arr:
        rec = Pose.from_buffer_copy(row.tobytes())
        if rec.num_votes:
            out.append(Pose3D(rec))
    return out


def gather_poses(local: np.ndarray, device=None):
    """all_gather of every rank's (k, POSE_WORDS) pose block -> (world, k, POSE_WORDS) on every rank.
    One collective, fixed size, latency-bound (k * 216 B per rank)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return local[None].copy()
    world = dist.get_world_size()
    t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64))
    if device is not None:
        t = t.to(device)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)  # concatenation along dim 0
    return out.cpu().numpy().reshape((world,) + tuple(t.shape))


def gather_device(block, dist=None, force=False):
    """all_gather of a device tensor block (rows = pose records as float64 words) without leaving HBM: every rank
    contributes the same shape; returns (world * rows, words) on the device (RCCL under backend "nccl").  With the
    gloo rehearsal backend the block travels through the host.  A world of one returns the block itself unless `force`
    asks for the collective anyway (tests/test_gpu_rccl.py: RCCL's first call must not be the 8-GPU run)."""
    import torch
    import torch.distributed as tdist

    d = dist or tdist
    if not (d.is_available() and d.is_initialized()) or (d.get_world_size() == 1 and not force):
        return block
    world = d.get_world_size()
    if d.get_backend() != "nccl":
        host = block.cpu()
        out = torch.empty((world * host.shape[0],) + tuple(host.shape[1:]), dtype=host.dtype)
        d.all_gather_into_tensor(out, host)
        return out.to(block.device)
    out = torch.empty((world * block.shape[0],) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
    d.all_gather_into_tensor(out, block)
    return out


def merge_reference_shards_device(gathered, world: int, n_total: int):
    """Device twin of merge_reference_shards: `gathered` is (world * per_rank, words) from gather_device, rank r's block
    holding reference points r, r + world, ...; returns the first n_total rows in reference-point order, contiguous."""
    per = gathered.shape[0] // world
    return gathered.view(world, per, gathered.shape[1]).transpose(0, 1).reshape(per * world, gathered.shape[1])[:n_total].contiguous()


def merge_reference_shards(shards: Sequence[np.ndarray]) -> np.ndarray:
    """Interleave per-rank blocks of per-reference records (rank r holds reference points r, r+world, ...)
    back into reference-point order."""
    world = len(shards)
    total = sum(s.shape[0] for s in shards)
    out = np.zeros((total,) + tuple(shards[0].shape[1:]), dtype=shards[0].dtype)
    for r, s in enumerate(shards):
        out[r::world][: s.shape[0]] = s
    return out
