"""Device-resident matching: clouds already in HBM, explicit stream, reusable workspace.

This is the entry bench.py and the multi-GPU path use (ppf_match_device in include/ppf_hip.h).
torch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from ._capi import MatchStats, Pose, Vote, check, lib
from .detector import Pose3D, PPF3DDetector


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

    def match_device(self, det: PPF3DDetector, d_scene_ptr: int, ns: int, stride: int, step: float, dist: float,
                     *, presampled: bool = True, d_edge_ptr: Optional[int] = None, ne: int = 0, estride: int = 6,
                     ref_offset: int = 0, ref_stride: int = 1, skip_clustering: bool = False, stream: int = 0):
        """Enqueue one match on `stream` (raw hipStream_t value, 0 = default stream)."""
        det._require_trained()
        mp = det._params(step, dist, presampled, ref_offset, ref_stride, skip_clustering)
        check(lib().ppf_match_device(det._model.ptr, self.ptr, C.c_void_p(d_scene_ptr), ns, stride,
                                     C.c_void_p(d_edge_ptr) if d_edge_ptr else None, ne, estride, C.byref(mp),
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
                "stats": {k: getattr(st, k) for k, _ in MatchStats._fields_}}

    def top_poses(self, k: int):
        """Wait for the call and fetch only the clustered poses (as a ctypes array of ppf_pose records, at most
        `cap` of them are converted by the caller) and the counters -- the lean per-step read-back."""
        if not hasattr(self, "_fin") or len(self._fin) < self._cap_hint:
            self._fin = (Pose * self._cap_hint)()
        n_pose = C.c_int(0)
        st = MatchStats()
        check(lib().ppf_workspace_results(self.ptr, None, None, 0, None, self._fin, self._cap_hint, C.byref(n_pose),
                                          C.byref(st)))
        return self._fin, min(n_pose.value, k), n_pose.value, {f: getattr(st, f) for f, _ in MatchStats._fields_}

    _cap_hint = 65536

    def stats(self) -> dict:
        st = MatchStats()
        check(lib().ppf_workspace_results(self.ptr, None, None, 0, None, None, 0, None, C.byref(st)))
        return {k: getattr(st, k) for k, _ in MatchStats._fields_}

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
