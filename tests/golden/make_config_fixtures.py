"""Generate tests/golden/config_c{2,3,4,5}.npz: CPU-oracle results on the BASELINE.json configurations.

    python tests/golden/make_config_fixtures.py [c2] [c3] [c4] [c5]        (no argument: all four)

The -m gpu tests (tests/test_gpu_configs.py) compare the HIP path with these files, so every configuration the
bench measures is checked against the oracle, not against the engine itself:

  c2  the bench workload at FULL size: all 2,500 vote triples, votes and pairs per reference point, the top
      clustered poses (about 4 minutes on 8 cores)
  c3  the per-rank crops of the 8-GPU run (seeds 1000..1007, 50,000 points): 20 evenly spaced reference points each
  c4  ~10k-point model vs 200,000-point scene: 40 evenly spaced reference points
  c5  4 models x 8 crops: whole matches (top-5 clustered poses per pair) on 12,000-point crops, plus 10 reference
      points per pair at the full 50,000-point size for crops 0 and 5

"parity unpinned" caveat: the reference holds no golden vectors for this path and its library
cannot be built here, so these files pin OUR frozen restatement of the upstream algorithm (oracle/ppf_oracle.cpp).
Inputs are regenerated from seeds by the tests; each fixture carries the sha256 of the cloud the oracle saw.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from yolo_ppf_pose_estimation_amd import workloads as W  # noqa: E402

STEP = 20


def _poses(r, k=W.TOP_K):
    p = r["poses"][:k]
    return (np.stack([x["pose"] for x in p]) if p else np.zeros((0, 4, 4)),
            np.array([x["num_votes"] for x in p], dtype=np.int64))


def _raw(r):
    return np.stack([x["pose"] for x in r["raw_poses"]])


def make_c2():
    ora = O.OracleDetector(W.C2["model_step"], W.REL_DISTANCE).train_model(W.bottle())
    scene = W.c2_scene()
    t0 = time.time()
    r = ora.match(scene, relative_scene_sample_step=W.SCENE_STEP, relative_scene_distance=W.REL_DISTANCE, presampled=True)
    print(f"c2: {r['n_ref']} reference points in {time.time() - t0:.1f} s, votes {int(r['votes_per_ref'].sum())}")
    top, topv = _poses(r)
    np.savez_compressed(os.path.join(HERE, "config_c2.npz"), digest=np.array(W.cloud_digest(scene)),
                        n_model=np.array([ora.info()["n_ref"]]), triples=r["triples"], votes=r["votes_per_ref"],
                        pairs=r["pairs_per_ref"], raw_pose_first=_raw(r)[:: 125], top_poses=top, top_votes=topv,
                        n_final=np.array([r["n_final"]]))


def make_c3():
    ora = O.OracleDetector(W.C2["model_step"], W.REL_DISTANCE).train_model(W.bottle())
    out = {}
    for rank in range(8):
        scene = W.c3_scene(rank)
        refs = [(3 + k * 125) * STEP for k in range(20)]
        r = ora.match(scene, relative_scene_sample_step=W.SCENE_STEP, presampled=True, ref_list=refs, cluster=False)
        out[f"digest_{rank}"] = np.array(W.cloud_digest(scene))
        out[f"triples_{rank}"] = r["triples"]
        out[f"votes_{rank}"] = r["votes_per_ref"]
        out[f"raw_{rank}"] = _raw(r)
        print(f"c3 rank {rank}: votes {int(r['votes_per_ref'].sum())}")
    out["ref_offset"], out["ref_stride"] = np.array([3]), np.array([125])
    np.savez_compressed(os.path.join(HERE, "config_c3.npz"), **out)


def make_c4():
    t0 = time.time()
    ora = O.OracleDetector(W.C4["model_step"], W.REL_DISTANCE).train_model(W.bottle())
    print(f"c4: oracle table of {ora.info()['n_ref']} model points in {time.time() - t0:.1f} s")
    scene = W.c4_scene()
    refs = [(3 + k * 250) * STEP for k in range(40)]
    t0 = time.time()
    r = ora.match(scene, relative_scene_sample_step=W.SCENE_STEP, presampled=True, ref_list=refs, cluster=False)
    print(f"c4: 40 reference points in {time.time() - t0:.1f} s, votes {int(r['votes_per_ref'].sum())}")
    np.savez_compressed(os.path.join(HERE, "config_c4.npz"), digest=np.array(W.cloud_digest(scene)),
                        n_model=np.array([ora.info()["n_ref"]]), ref_offset=np.array([3]), ref_stride=np.array([250]),
                        triples=r["triples"], votes=r["votes_per_ref"], pairs=r["pairs_per_ref"], raw=_raw(r))


def make_c5():
    models = W.c5_models()
    oras = [O.OracleDetector(W.C5_MODEL_STEP, W.REL_DISTANCE).train_model(m) for m in models]
    out = {"n_model": np.array([o.info()["n_ref"] for o in oras])}
    small = W.c5_crops(0, n_points=12000, models=models)
    for c, crop in enumerate(small):
        out[f"small_digest_{c}"] = np.array(W.cloud_digest(crop))
        for k, ora in enumerate(oras):
            r = ora.match(crop, relative_scene_sample_step=W.SCENE_STEP, presampled=True)
            top, topv = _poses(r)
            out[f"small_top_poses_{c}_{k}"], out[f"small_top_votes_{c}_{k}"] = top, topv
            out[f"small_n_final_{c}_{k}"] = np.array([r["n_final"]])
            out[f"small_votes_{c}_{k}"] = np.array([int(r["votes_per_ref"].sum())], dtype=np.uint64)
            out[f"small_triples_{c}_{k}"] = r["triples"]
        print(f"c5 small crop {c} done")
    full = W.c5_crops(0, models=models)
    refs = [(5 + k * 250) * STEP for k in range(10)]
    for c in (0, 5):
        out[f"full_digest_{c}"] = np.array(W.cloud_digest(full[c]))
        for k, ora in enumerate(oras):
            r = ora.match(full[c], relative_scene_sample_step=W.SCENE_STEP, presampled=True, ref_list=refs, cluster=False)
            out[f"full_triples_{c}_{k}"], out[f"full_votes_{c}_{k}"] = r["triples"], r["votes_per_ref"]
        print(f"c5 full crop {c} done")
    out["full_ref_offset"], out["full_ref_stride"] = np.array([5]), np.array([250])
    np.savez_compressed(os.path.join(HERE, "config_c5.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["c3", "c5", "c4", "c2"]
    for w in which:
        t0 = time.time()
        {"c2": make_c2, "c3": make_c3, "c4": make_c4, "c5": make_c5}[w]()
        print(f"{w} written in {time.time() - t0:.1f} s", flush=True)
