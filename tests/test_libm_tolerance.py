"""Expected deviation from an upstream build, stated at the level the reference consumes results.

The frozen numeric spec (include/ppf_detmath.h, "det" mode: what oracle and engine evaluate bit for bit) is not glibc; an
upstream build of the reference's library calls glibc's acos / atan2 / sin / cos (the oracle's "libm" mode).  BASELINE.json's
north_star asks for "poses within a stated rotation/translation tolerance and vote counts bit-exact"; this is that statement
for the closest stand-in for the real library that exists here (parity itself stays unpinned):

    tolerance:  vote triples differing on at most 0.1 % of the reference points; the five best clustered poses -- what
                /root/reference/include/CloudProcessing.h:455-470 hands to ICP -- within 1e-6 rad and 1e-9 model diameters
    measured:   0 of 2,500 triples (C2, 66,251,001,943 votes in both modes), 0 of 41 / 29 (C1 Matching / Matching_S2B);
                2.1e-8 rad, 5.2e-16 diameters  (tests/golden/libm_tolerance.json, generator make_libm_tolerance.py)

CPU only.  C1 is recomputed here in both modes; C2's libm run (ten minutes) is committed and a sample of it is re-run."""
import json
import os
import sys

import numpy as np

import oracle_lib as O
from conftest import GOLDEN
from yolo_ppf_pose_estimation_amd import workloads as W

sys.path.insert(0, GOLDEN)
from make_libm_tolerance import compare  # noqa: E402

MAX_TRIPLE_FRAC, MAX_ROT_RAD, MAX_TRANS_DIAMETERS = 1e-3, 1e-6, 1e-9


def _within(c):
    assert c["triples_differing_frac"] <= MAX_TRIPLE_FRAC, c
    assert c["votes_rel_diff"] <= 1e-5, c
    assert c["n_clusters_det"] == c["n_clusters_libm"] and c["top_votes_det"] == c["top_votes_libm"], c
    assert c["max_top_rot_diff_rad"] <= MAX_ROT_RAD and c["max_top_trans_diff_diameters"] <= MAX_TRANS_DIAMETERS, c


def test_committed_measurements_are_inside_the_stated_tolerance():
    d = json.load(open(os.path.join(GOLDEN, "libm_tolerance.json")))
    for key in ("c1_matching", "c1_matching_s2b", "c2"):
        _within(d[key])
    assert d["c2"]["n_ref"] == 2500 and d["c2"]["votes_det"] == 66251001943


def test_c1_reference_frame_det_against_libm(bottle):
    crop, edge = np.load(os.path.join(GOLDEN, "c1_crop_xyzn.npy")), np.load(os.path.join(GOLDEN, "c1_edge_xyzn.npy"))
    diameter = float(np.linalg.norm((bottle[:, :3].max(0) - bottle[:, :3].min(0)).astype(np.float32)))
    det = O.OracleDetector(0.025, 0.05, mode=O.MODE_DET).train_model(bottle)
    lib = O.OracleDetector(0.025, 0.05, mode=O.MODE_LIBM).train_model(bottle)
    _within(compare(det.match(crop, edge=edge, relative_scene_sample_step=0.05, relative_scene_distance=0.05),
                    lib.match(crop, edge=edge, relative_scene_sample_step=0.05, relative_scene_distance=0.05), diameter))


def test_c2_libm_fixture_is_what_the_libm_oracle_computes(bottle):
    """the committed libm-mode C2 result, re-run on every 250th reference point, and compared with the det-mode fixture"""
    fx_l = np.load(os.path.join(GOLDEN, "config_c2_libm.npz"))
    fx_d = np.load(os.path.join(GOLDEN, "config_c2.npz"))
    scene = W.c2_scene()
    assert W.cloud_digest(scene) == str(fx_l["digest"]) == str(fx_d["digest"])
    lib = O.OracleDetector(W.C2["model_step"], W.REL_DISTANCE, mode=O.MODE_LIBM).train_model(bottle)
    refs = list(range(0, scene.shape[0], 20))[::250]
    r = lib.match(scene, relative_scene_sample_step=W.SCENE_STEP, presampled=True, ref_list=refs, cluster=False)
    np.testing.assert_array_equal(r["triples"], fx_l["triples"][::250])
    assert (fx_l["triples"] != fx_d["triples"]).any(axis=1).mean() <= MAX_TRIPLE_FRAC
    assert list(fx_l["top_votes"]) == list(fx_d["top_votes"])
    assert np.abs(fx_l["top_poses"] - fx_d["top_poses"]).max() <= 1e-9
