"""Time the device ICP (and, optionally, the CPU oracle beside it) on the reference-sized case: the full bottle model
(19,753 rows) against a C2 crop (50,000 rows), top-5 poses of the match, ICP(100, 0.005, 2.5, 8).
Run on the GPU box:  python tools/icp_timing.py [--oracle]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from yolo_ppf_pose_estimation_amd import synth  # noqa: E402
from yolo_ppf_pose_estimation_amd.detector import ICP, PPF3DDetector  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle", action="store_true", help="also run the CPU oracle on the first pose and compare")
    ap.add_argument("--scene-points", type=int, default=50000)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--legacy", action="store_true", help="the earlier schedule (PPF_ICP_LEGACY): a stream per pose, exhaustive search")
    a = ap.parse_args()
    bottle = np.load(os.path.join(ROOT, "tests", "golden", "bottle_model_xyzn.npy"))
    scene, Ts = synth.make_scene(bottle, n_points=a.scene_points, seed=12345)
    det = PPF3DDetector(0.036, 0.05).trainModel(bottle)
    poses = det.match(scene, 1.0 / 20.0, 0.05)[:5]
    from yolo_ppf_pose_estimation_amd import _capi
    icp = ICP(100, 0.005, 2.5, 8, flags=_capi.PPF_ICP_LEGACY if a.legacy else 0)
    best = None
    for _ in range(a.repeat):
        work = [p.clone() for p in poses]
        t0 = time.perf_counter()
        icp.registerModelToScene(bottle, scene, work)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out = {"model_rows": int(bottle.shape[0]), "scene_rows": int(scene.shape[0]), "poses": len(work),
           "iterations": icp.last_iterations, "residuals": [p.residual for p in work], "gpu_seconds": best,
           "schedule": "legacy" if a.legacy else "batched"}
    if a.oracle:
        import oracle_lib as O
        t0 = time.perf_counter()
        P, r, it = O.icp_refine(bottle, scene, [poses[0].pose])
        out["oracle_seconds_first_pose"] = time.perf_counter() - t0
        out["oracle_threads"] = O.max_threads()
        out["bitwise_equal_first_pose"] = bool(np.array_equal(P[0], work[0].pose) and r[0] == work[0].residual
                                                and it[0] == icp.last_iterations[0])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
