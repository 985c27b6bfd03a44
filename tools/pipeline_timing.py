"""End-to-end time of everything the reference runs after YOLO's boxes, on its own depth frame, clouds resident:
upload frame -> SceneCropping -> Subsampling -> OutlierProcessing -> NormalEstimation -> EdgeExtraction ->
PointCloudXYZNormalToMat x2 -> Matching_S2B (match_S2B + ICP of the top 5) -> pose.
Run on the GPU box:  python tools/pipeline_timing.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import prep_data as D  # noqa: E402
from yolo_ppf_pose_estimation_amd.cloud_processor import CloudProcessor  # noqa: E402


def main():
    xyz, depth, box, intr = D.c1_frame()
    bottle = np.load(os.path.join(ROOT, "tests", "golden", "bottle_model_xyzn.npy"))
    K = np.array([[intr[0], 0, intr[2]], [0, intr[1], intr[3]], [0, 0, 1.0]])
    leaf, sor = 0.003, 1.0

    def run(cp):
        cp.SceneCropping(K)
        cp.Subsampling(leaf)
        cp.OutlierProcessing(50, sor)
        cp.NormalEstimation(30)
        cp.EdgeExtraction(0.03)
        obj = cp.PointCloudXYZNormalToMat(cp.objects_with_normals[0], resident=True)
        edge = cp.PointCloudXYZNormalToMat(cp.objects_edges[0], resident=True)
        return cp.Matching_S2B("bottle", obj, edge), len(obj), len(edge)

    t0 = time.perf_counter()
    trainer = CloudProcessor(None, None, [], [], [], 0.025, 0.05)
    trainer.LoadSingleModel(bottle, "bottle")
    trainer.TrainDetector(0.025, 0.05)     # the reference's train parameters (CloudProcessing.h:64-65)
    train_s = time.perf_counter() - t0
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        cp = CloudProcessor(xyz, depth, [box], [39], [0], 0.025, 0.05)
        cp.models, cp.detectors, cp.if_trained = trainer.models, trainer.detectors, trainer.if_trained
        cp.label_to_id, cp.id_to_label, cp._model_clouds = trainer.label_to_id, trainer.id_to_label, trainer._model_clouds
        pose, n_obj, n_edge = run(cp)
        times.append(time.perf_counter() - t0)
        parts = {k: round(v * 1e3, 3) for k, v in cp.timings.items()}
    print(json.dumps({"frame_points": int(xyz.shape[0]), "object_points": n_obj, "edge_points": n_edge,
                      "model_sampled_points": trainer.detectors[0].info()["n_ref"], "train_seconds": train_s,
                      "frame_to_pose_ms": [round(t * 1e3, 3) for t in times], "of_which_ms": parts, "votes": pose.numVotes, "residual": pose.residual,
                      "pose": pose.pose.round(6).tolist()}))


if __name__ == "__main__":
    main()
