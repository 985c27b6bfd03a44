"""Device-resident matching: clouds already in HBM, explicit stream, reusable workspace.

This is the entry bench.py and the multi-GPU path use (ppf_match_device in include/ppf_hip.h).
torch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _capi
from ._capi import BatchStats, MatchStats, Pose, Vote, check, lib
from .detector import Pose3D, PPF3DDetector

POSE_WORDS = C.sizeof(Pose) // 8  # a ppf_pose record as float64 words (27)


class Workspace:
    def __init__(self, timing: bool = False):
        p = C.c_void_p()
        check(lib().ppf_workspace_create(C.byref(p)))
        self.ptr = p.value
        if timing:
            check(lib().ppf_workspace_enable_timing(self.ptr, 1))

    def __del__(self):
        try:
            if self.ptr:
                lib().ppf_workspace_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass

    def set_option(self, option: int, value: float):
        """ppf_workspace_set_option: _capi.PPF_OPT_HIT_FRACTION / PPF_OPT_GROUP_ROUND_BUCKETS."""
        check(lib().ppf_workspace_set_option(self.ptr, int(option), float(value)))

    def match_device(self, det: PPF3DDetector, d_scene_ptr: int, ns: int, stride: int, step: float, dist: float,
                     *, presampled: bool = True, d_edge_ptr: Optional[int] = None, ne: int = 0, estride: int = 6,
                     ref_offset: int = 0, ref_stride: int = 1, skip_clustering: bool = False, stream: int = 0,
                     vote_mode: int = 0, normal_offset: int = 3, edge_normal_offset: int = 3):
        """Enqueue one match on `stream` (raw hipStream_t value, 0 = default stream)."""
        det._require_trained()
        mp = det._params(step, dist, presampled, ref_offset, ref_stride, skip_clustering, vote_mode)
        check(lib().ppf_match_device(det._model.ptr, self.ptr, C.c_void_p(d_scene_ptr), ns, stride, normal_offset,
                                     C.c_void_p(d_edge_ptr) if d_edge_ptr else None, ne, estride, edge_normal_offset, C.byref(mp),
                                     C.c_void_p(stream) if stream else None))

    def results(self, cap_ref: int, want_poses: bool = True) -> dict:
        votes = (Vote * max(cap_ref, 1))()
        raw = (Pose * max(cap_ref, 1))()
        fin = (Pose * max(cap_ref, 1))()
        n_ref, n_pose = C.c_int(0), C.c_int(0)
        st = MatchStats()
        check(lib().ppf_workspace_results(self.ptr, votes, raw, cap_ref, C.byref(n_ref), fin if want_poses else None,
                                          cap_ref, C.byref(n_pose) if want_poses else None, C.byref(st)))
        nr = n_ref.value
        tri = np.frombuffer(votes, dtype=np.uint32)[: 3 * nr].reshape(nr, 3).copy()
        return {"n_ref": nr, "triples": tri, "raw_poses": [Pose3D(raw[i]) for i in range(nr)],
                "poses": [Pose3D(fin[i]) for i in range(n_pose.value)] if want_poses else [],
                "stats": _capi.stats_dict(st)}

    def top_poses(self, k: int):
        """Wait for the call and fetch only the clustered poses (as a ctypes array of ppf_pose records, at most
        `cap` of them are converted by the caller) and the counters -- the lean per-step read-back."""
        if not hasattr(self, "_fin") or len(self._fin) < self._cap_hint:
            self._fin = (Pose * self._cap_hint)()
        n_pose = C.c_int(0)
        st = MatchStats()
        check(lib().ppf_workspace_results(self.ptr, None, None, 0, None, self._fin, self._cap_hint, C.byref(n_pose),
                                          C.byref(st)))
        return self._fin, min(n_pose.value, k), n_pose.value, _capi.stats_dict(st)

    _cap_hint = 65536

    def stats(self) -> dict:
        st = MatchStats()
        check(lib().ppf_workspace_results(self.ptr, None, None, 0, None, None, 0, None, C.byref(st)))
        return _capi.stats_dict(st)

    def ref_counters(self, cap: int):
        v = np.zeros(cap, dtype=np.uint64)
        p = np.zeros(cap, dtype=np.uint64)
        check(lib().ppf_workspace_ref_counters(self.ptr, v.ctypes.data, p.ctypes.data, cap))
        return v, p

    def device_poses(self):
        ptr = C.c_void_p()
        n = C.c_int(0)
        check(lib().ppf_workspace_device_poses(self.ptr, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    # -- device-side result blocks (torch tensors of float64 words, one row per ppf_pose record) ----------------------
    def device_top_block(self, k: int, stream: int = 0):
        """(k, 27) float64 CUDA tensor: the best k clustered poses, zero rows past the count.  Filled on `stream` by a
        device-to-device copy; call stats()/results() first (they notice and repeat a call whose hit pools were too
        small)."""
        import torch
        out = torch.empty((k, POSE_WORDS), dtype=torch.float64, device="cuda")
        check(lib().ppf_workspace_copy_top_poses(self.ptr, C.c_void_p(out.data_ptr()), k, C.c_void_p(stream) if stream else None))
        return out

    def device_pose_block(self, cap: int, stream: int = 0):
        """(cap, 27) float64 CUDA tensor: the per-reference poses of the last call, zero rows past n_ref."""
        import torch
        out = torch.empty((cap, POSE_WORDS), dtype=torch.float64, device="cuda")
        check(lib().ppf_workspace_copy_raw_poses(self.ptr, C.c_void_p(out.data_ptr()), cap, C.c_void_p(stream) if stream else None))
        return out

    def cluster_device(self, det: PPF3DDetector, d_poses_ptr: int, n: int, num_poses: int, stream: int = 0, top_k: int = 5):
        """clusterPoses on a device pose list (ppf_cluster_poses_device); returns the best top_k as a device block."""
        det._require_trained()
        mp = det._params(1.0, 0.05, True)
        check(lib().ppf_cluster_poses_device(det._model.ptr, self.ptr, C.c_void_p(d_poses_ptr), n, num_poses, C.byref(mp),
                                             C.c_void_p(stream) if stream else None))
        return self.device_top_block(top_k, stream)


class BatchMatcher:
    """ppf_batch_*: crops x models over `lanes` streams with their own workspaces (BASELINE config C5)."""

    def __init__(self, lanes: int = 4, timing: bool = False):
        p = C.c_void_p()
        check(lib().ppf_batch_create(int(lanes), C.byref(p)))
        self.ptr = p.value
        self.lanes = int(lanes)
        if timing:  # HIP events around the kernels of every match: res["ms_vote_kernel"] etc. are sums over the run
            check(lib().ppf_batch_enable_timing(self.ptr, 1))

    def __del__(self):
        try:
            if self.ptr:
                lib().ppf_batch_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass

    def _run(self, dets, ptrs, ns, stride, on_device, step, dist, presampled, top_k, want_host):
        for d in dets:
            d._require_trained()
        nm, nc = len(dets), len(ptrs)
        models = (C.c_void_p * nm)(*[d._model.ptr for d in dets])
        scenes = (C.c_void_p * nc)(*ptrs)
        cnt = (C.c_int * nc)(*ns)
        mp = dets[0]._params(step, dist, presampled)
        out = (Pose * (nc * nm * top_k))() if want_host else None
        n_out = (C.c_int * (nc * nm))()
        st = BatchStats()
        check(lib().ppf_batch_run(self.ptr, models, nm, scenes, cnt, stride, 3, nc, int(on_device), C.byref(mp), out, top_k,
                                  n_out, C.byref(st)))
        res = {f: getattr(st, f) for f, _ in BatchStats._fields_}
        res["n_out"] = [n_out[i] for i in range(nc * nm)]
        if want_host:
            res["poses"] = [[[Pose3D(out[(c * nm + k) * top_k + i]) for i in range(n_out[c * nm + k])] for k in range(nm)]
                            for c in range(nc)]
        return res

    def run(self, dets, scenes, step, dist, *, presampled=False, top_k=5):
        """Host crops (N x 6 float32 arrays): staged through pinned memory inside the library."""
        clouds = [np.ascontiguousarray(s, dtype=np.float32) for s in scenes]
        self._keep = clouds
        return self._run(dets, [c.ctypes.data for c in clouds], [c.shape[0] for c in clouds], clouds[0].shape[1], False,
                         step, dist, presampled, top_k, True)

    def run_device(self, dets, d_ptrs, ns, stride, step, dist, *, presampled=True, top_k=5, want_host=False):
        """Device-resident crops; the pose block stays in HBM (res['d_top'], (crops*models*top_k, 27) float64)."""
        res = self._run(dets, list(d_ptrs), list(ns), stride, True, step, dist, presampled, top_k, want_host)
        res["d_top"] = self.device_block()
        return res

    def device_block(self):
        import torch
        ptr = C.c_void_p()
        n = C.c_int(0)
        check(lib().ppf_batch_device_block(self.ptr, C.byref(ptr), C.byref(n)))
        out = torch.empty((n.value, POSE_WORDS), dtype=torch.float64, device="cuda")
        if n.value:  # device-to-device copy into a tensor torch owns (the block itself belongs to the batch context)
            check(lib().ppf_batch_copy_block(self.ptr, C.c_void_p(out.data_ptr()), n.value, None))
        return out
