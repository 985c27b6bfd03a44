"""Generate tests/golden/oracle_golden.npz: known-answer vectors of the CPU oracle (det mode).

    python tests/golden/make_golden.py

"parity unpinned" caveat: the reference ships no golden vectors for this path and
its library cannot be built here, so these vectors pin OUR frozen restatement (they catch
regressions of the oracle and are what the GPU path must reproduce bit-for-bit); they were not
produced by the reference.  Inputs are seeded numpy arrays and the committed bottle fixture.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from yolo_ppf_pose_estimation_amd import synth  # noqa: E402


def tiny_cloud(seed, n):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-0.05, 0.05, size=(n, 3))
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return np.concatenate([p, nrm], axis=1).astype(np.float32)


def main():
    out = {}
    bottle = np.load(os.path.join(HERE, "bottle_model_xyzn.npy"))
    # 1. tiny hand-sized case: 6-point model, 8-point scene, full accumulator of every reference point
    model = tiny_cloud(1, 6)
    scene = np.concatenate([synth.apply_pose(model, synth.rigid_pose(5, 0.02)), tiny_cloud(2, 2)], axis=0)
    det = O.OracleDetector(0.05, 0.05).train_model(model, presampled=True)
    out["tiny_model"], out["tiny_scene"] = model, scene
    out["tiny_info"] = np.array([det.info()["slots"], det.info()["num_angles"]], dtype=np.int64)
    hsh, alp = det.pairs()
    out["tiny_pair_hash"], out["tiny_pair_alpha"] = hsh, alp
    out["tiny_acc"] = np.stack([det.accumulator(scene, i) for i in range(scene.shape[0])])
    r = det.match(scene, relative_scene_sample_step=1.0, presampled=True)
    out["tiny_triples"] = r["triples"]
    # 2. bottle model (step 0.07) vs a 1500-point synthetic crop: per-reference triples + counters + top poses
    det = O.OracleDetector(0.07, 0.05).train_model(bottle)
    scene, _ = synth.make_scene(bottle, n_points=1500, seed=77)
    r = det.match(scene, relative_scene_sample_step=1.0 / 10.0, presampled=True)
    out["b07_sampled_model"] = det.sampled_model()
    out["b07_scene_seed"] = np.array([77, 1500], dtype=np.int64)
    out["b07_triples"] = r["triples"]
    out["b07_votes"] = r["votes_per_ref"]
    out["b07_pairs"] = r["pairs_per_ref"]
    out["b07_top_poses"] = np.stack([p["pose"] for p in r["poses"][:5]])
    out["b07_top_votes"] = np.array([p["num_votes"] for p in r["poses"][:5]], dtype=np.int64)
    out["b07_n_final"] = np.array([r["n_final"]], dtype=np.int64)
    bs = det.bucket_stats()
    out["b07_bucket_stats"] = np.array([bs["non_empty"], bs["max_len"]], dtype=np.int64)
    # 3. sampling regression constants measured from the reference's bottle PLY
    out["bottle_sample_steps"] = np.array([0.025, 0.035, 0.036, 0.05, 0.0714])
    out["bottle_sample_counts"] = np.array([O.sample(bottle, s).shape[0] for s in out["bottle_sample_steps"]],
                                           dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
