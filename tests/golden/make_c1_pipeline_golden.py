"""Regenerate tests/golden/c1_pipeline_golden.npz: the ORACLES' result for the reference's whole post-YOLO chain on its
own depth frame (tests/golden/c1_depth_window.npz), with the reference's parameters:

    SceneCropping (fixed bbox, YOLO weights are not shipped) -> Subsampling(leaf 3 mm) -> OutlierProcessing(50, 1.0)
    -> NormalEstimation(30) -> EdgeExtraction(0.03) -> PointCloudXYZNormalToMat x2
    -> PPF3DDetector(0.025, 0.05).trainModel(bottle) -> match_S2B(scene, edge, 0.05, 0.05) -> top 5
    -> ICP(100, 0.005, 2.5, 8).registerModelToScene(bottle, scene, top 5)

    python tests/golden/make_c1_pipeline_golden.py        (CPU only; ~1 minute)

The file pins OUR frozen specification across rounds (the reference has no recorded outputs: parity unpinned)."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib as O  # noqa: E402
import prep_data as D  # noqa: E402

LEAF, SOR_MUL, TRAIN = 0.003, 1.0, (0.025, 0.05)


def chain():
    xyz, depth, box, intr = D.c1_frame()
    bottle = np.load(os.path.join(HERE, "bottle_model_xyzn.npy"))
    keep, _ = O.prep_crop(xyz, box, depth, intr)
    v = O.prep_voxel(xyz[keep], LEAF)
    k2, _, _ = O.prep_sor(v, 50, SOR_MUL)
    v = v[k2]
    n, c = O.prep_normals(v, 30)
    obj = O.prep_to_mat(v, n)
    edge = O.prep_to_mat(v[c > 0.03], n[c > 0.03])
    ora = O.OracleDetector(*TRAIN).train_model(bottle)
    m = ora.match(obj, edge=edge, relative_scene_sample_step=0.05, relative_scene_distance=0.05, cluster=True)
    top = m["poses"][:5]
    P, res, its = O.icp_refine(bottle, obj, [p["pose"] for p in top])
    return dict(n_crop=np.int64(keep.size), n_object=np.int64(obj.shape[0]), n_edge=np.int64(edge.shape[0]),
                object_checksum=np.float64(obj.astype(np.float64).sum()), edge_checksum=np.float64(edge.astype(np.float64).sum()),
                n_model_sampled=np.int64(ora.info()["n_ref"]), n_ref=np.int64(m["n_ref"]), total_votes=np.int64(m["votes_per_ref"].sum()),
                n_clusters=np.int64(m["n_final"]), top_votes=np.array([p["num_votes"] for p in top], dtype=np.int64),
                match_poses=np.array([p["pose"] for p in top]), icp_poses=P, icp_residuals=res, icp_iterations=its.astype(np.int64))


if __name__ == "__main__":
    t0 = time.time()
    g = chain()
    np.savez(os.path.join(HERE, "c1_pipeline_golden.npz"), **g)
    for k, v in g.items():
        print(k, v if np.size(v) < 8 else np.shape(v))
    print("seconds", time.time() - t0)
