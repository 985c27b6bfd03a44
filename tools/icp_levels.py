"""Why do refined poses come back with the residual 9999999999 after one to three iterations?  CPU only.

Runs the ICP oracle (oracle/ppf_icp_oracle.cpp, bit-identical to the device ICP) on the top-5 matched poses of the C2 crop --
the case profiles/r02_icp_timing.json shows -- with its per-pass trace switched on and prints, per pose and pyramid level, the
correspondences that survive the rejection threshold and picky ICP and how the level ended.

    python tools/icp_levels.py [--scene-points 50000]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from yolo_ppf_pose_estimation_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene-points", type=int, default=50000)
    a = ap.parse_args()
    bottle = np.load(os.path.join(ROOT, "tests", "golden", "bottle_model_xyzn.npy"))
    scene, Ts = synth.make_scene(bottle, n_points=a.scene_points, seed=12345)
    ora = O.OracleDetector(0.036, 0.05).train_model(bottle)
    r = ora.match(scene, relative_scene_sample_step=1.0 / 20.0, relative_scene_distance=0.05)
    poses = [np.asarray(p["pose"]) for p in r["poses"][:5]]
    from scipy.spatial import cKDTree
    tree = cKDTree(scene[:, :3].astype(np.float64))
    out = []
    for k, P in enumerate(poses):
        refined, res, its, trace = O.icp_refine_traced(bottle, scene, [P])
        moved = bottle[::10, :3].astype(np.float64) @ P[:3, :3].T + P[:3, 3]
        dd, ii = tree.query(moved)
        levels = {}
        for lv, it, ns, nd, acc, sel, code in trace.tolist():
            levels.setdefault(lv, []).append({"iteration": it, "model_rows": ns, "scene_rows": nd, "accepted": acc, "kept": sel,
                                              "exit": ["iterated", "<= 6 correspondences", "solve failed", "NaN"][code]})
        out.append({"pose": k, "votes": int(r["poses"][k]["num_votes"]), "iterations": int(its[0]), "residual": float(res[0]),
                    "median_distance_of_posed_model_to_scene_m": float(np.median(dd)),
                    "distinct_nearest_scene_points": int(len(np.unique(ii))), "model_points_queried": int(len(ii)), "levels": levels})
        print(f"pose {k}: votes {out[-1]['votes']}, {its[0]} iterations, residual {res[0]:g}, median distance of the posed model to the "
              f"scene {np.median(dd):.4f} m, {len(np.unique(ii))} distinct nearest scene points for {len(ii)} model points")
        for lv in sorted(levels, reverse=True):
            print("   level", lv, [(e["iteration"], e["accepted"], e["kept"], e["exit"]) for e in levels[lv]])
    json.dump(out, open(os.path.join(ROOT, "profiles", "r03_icp_levels.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
