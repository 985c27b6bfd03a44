"""Time the device cloud stages on the reference's depth frame (and the kNN on a larger cloud).
Run on the GPU box:  python tools/prep_timing.py        (under rocprofv3 --kernel-trace --stats for kernel times)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import prep_data as D  # noqa: E402
from yolo_ppf_pose_estimation_amd.cloud_processor import DeviceCloud  # noqa: E402


def timed(fn, repeat=5):
    best, out = None, None
    for _ in range(repeat):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return out, best * 1e3


def main():
    xyz, depth, box, intr = D.c1_frame()
    res = {"frame_points": int(xyz.shape[0])}
    scene, res["upload_ms"] = timed(lambda: DeviceCloud.upload(xyz))
    crop, res["crop_ms"] = timed(lambda: scene.crop(box, depth, intr))
    sub, res["voxel_ms"] = timed(lambda: crop.voxel_grid(0.003))
    filt, res["sor_ms"] = timed(lambda: sub.outlier_removal(50, 1.0))
    wn, res["normals_ms"] = timed(lambda: filt.normals(30))
    edges, res["edges_ms"] = timed(lambda: wn.edges(0.03))
    _, res["to_mat_ms"] = timed(lambda: wn.to_mat().rows())
    res.update(crop_points=len(crop), voxel_points=len(sub), sor_points=len(filt), edge_points=len(edges))
    big = DeviceCloud.upload(xyz[:50000])
    _, res["normals_50k_ms"] = timed(lambda: big.normals(30), repeat=3)
    _, res["sor_50k_ms"] = timed(lambda: big.outlier_removal(50, 1.0), repeat=3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
