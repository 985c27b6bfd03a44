"""Seeded random sweep of the hot path against the oracle: 24 configurations (PPF_SOAK_MATCH=N adds N more draws: the
soak runs of tools/soak.sh, results under profiles/) drawn over model shape and sampling
step, alpha resolution, distance-step rule, scene size / sampling / reference stride, presampled or not, surface or
surface-to-boundary matching, clustering thresholds, forced accumulator tiling, and for every third draw the vote kernel's
32-bit-cell instantiation as well.  Every one must give bit-exact vote
triples, vote and pair totals, bit-identical raw poses and the oracle's clustered poses (the bar of
tests/test_gpu_parity.py)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import soak_seeds
from yolo_ppf_pose_estimation_amd import synth
from yolo_ppf_pose_estimation_amd.detector import PPF3DDetector

pytestmark = pytest.mark.gpu

KINDS = ["bottle", "box", "cylinder", "torus"]


def _draw(seed):
    rng = np.random.default_rng(1000 + seed)
    cfg = dict(kind=KINDS[int(rng.integers(0, 4))],
               train_step=float(rng.choice([0.04, 0.05, 0.0625, 0.08, 0.1])),
               dist_step=float(rng.choice([0.03, 0.05, 0.08])),
               num_angles=int(rng.choice([15, 24, 30, 40])),
               n_scene=int(rng.integers(600, 4000)),
               scene_seed=int(rng.integers(0, 10 ** 6)),
               presampled=bool(rng.integers(0, 2)),
               ref_step=float(rng.choice([1.0 / 3.0, 1.0 / 7.0, 1.0 / 10.0, 1.0 / 25.0])),
               scene_dist=float(rng.choice([0.03, 0.05, 0.07])),
               s2b=bool(rng.integers(0, 3) == 0),
               weighted=bool(rng.integers(0, 4) == 0),
               pos_thr=float(rng.choice([-1.0, 0.02, 0.08])),
               rot_thr=float(rng.choice([-1.0, 0.2, 0.6])),
               max_tile_refs=int(rng.choice([0, 0, 37, 150])))
    return cfg


@pytest.mark.parametrize("seed", soak_seeds(24, "PPF_SOAK_MATCH"))
def test_random_configuration(bottle, seed):
    cfg = _draw(seed)
    model = bottle if cfg["kind"] == "bottle" else synth.make_solid(cfg["kind"], 6000, seed=seed + 1)
    det = PPF3DDetector(cfg["train_step"], cfg["dist_step"], cfg["num_angles"], max_tile_refs=cfg["max_tile_refs"])
    det.trainModel(model)
    ora = O.OracleDetector(cfg["train_step"], cfg["dist_step"], cfg["num_angles"]).train_model(model)
    det.setSearchParams(cfg["pos_thr"], cfg["rot_thr"], cfg["weighted"])
    ora.set_search_params(cfg["pos_thr"], cfg["rot_thr"], cfg["weighted"])
    info, oinfo = det.info(), ora.info()
    assert (info["n_ref"], info["num_angles"], info["slots"]) == (oinfo["n_ref"], oinfo["num_angles"], oinfo["slots"])
    if cfg["max_tile_refs"]:
        assert info["tile_refs"] <= cfg["max_tile_refs"]
    scene, _ = synth.make_scene(model, n_points=cfg["n_scene"], seed=cfg["scene_seed"])
    edge = None
    if cfg["s2b"]:
        rng = np.random.default_rng(seed)
        edge = scene[rng.random(scene.shape[0]) < 0.3]
        if not cfg["presampled"]:
            edge = edge + 0.0  # same rows as the surface cloud: exercises the "never pair a point with itself" rule
    got = det.raw_votes(scene, cfg["ref_step"], cfg["scene_dist"], presampled=cfg["presampled"], edge=edge)
    want = ora.match(scene, edge=edge, relative_scene_sample_step=cfg["ref_step"], relative_scene_distance=cfg["scene_dist"],
                     presampled=cfg["presampled"], cluster=True)
    assert got["n_ref"] == want["n_ref"], cfg
    np.testing.assert_array_equal(got["triples"], want["triples"], err_msg=str(cfg))
    assert got["stats"]["n_votes"] == int(want["votes_per_ref"].sum()), cfg
    assert got["stats"]["n_pairs"] == int(want["pairs_per_ref"].sum()), cfg
    for g, w in zip(got["raw_poses"], want["raw_poses"]):
        assert np.array_equal(g.pose, w["pose"]), cfg
    if seed % 3 == 0:  # the 32-bit instantiation of the vote kernel on the same draw (device-resident entry, forced)
        import torch
        from yolo_ppf_pose_estimation_amd import _capi
        from yolo_ppf_pose_estimation_amd.device import Workspace
        ws = Workspace()
        ws.set_option(_capi.PPF_OPT_ACC32, 1)
        d_scene = torch.from_numpy(np.ascontiguousarray(scene, dtype=np.float32)).cuda()
        d_edge = torch.from_numpy(np.ascontiguousarray(edge, dtype=np.float32)).cuda() if edge is not None else None
        ws.match_device(det, d_scene.data_ptr(), scene.shape[0], 6, cfg["ref_step"], cfg["scene_dist"], presampled=cfg["presampled"],
                        d_edge_ptr=d_edge.data_ptr() if d_edge is not None else None, ne=0 if edge is None else edge.shape[0],
                        skip_clustering=True)
        res = ws.results(max(scene.shape[0], 1))
        np.testing.assert_array_equal(res["triples"], want["triples"], err_msg="32-bit cells " + str(cfg))
        assert res["stats"]["n_votes"] == int(want["votes_per_ref"].sum()) and res["stats"]["n_acc32_items"] > 0, cfg
    poses = det.match(scene, cfg["ref_step"], cfg["scene_dist"], presampled=cfg["presampled"], edge=edge)
    assert len(poses) == want["n_final"], cfg
    for g, w in zip(poses, want["poses"]):
        assert g.numVotes == w["num_votes"], cfg
        np.testing.assert_allclose(g.pose, w["pose"], rtol=0, atol=1e-9, err_msg=str(cfg))


def _draw_policy(seed):
    rng = np.random.default_rng(5000 + seed)
    return dict(kind=KINDS[int(rng.integers(0, 4))],
                train_step=float(rng.choice([0.05, 0.0625, 0.08])), num_angles=int(rng.choice([15, 30, 31, 40])),
                n_scene=int(rng.integers(800, 5000)), scene_seed=int(rng.integers(0, 10 ** 6)),
                ref_step=float(rng.choice([1.0 / 5.0, 1.0 / 10.0, 1.0 / 25.0])),
                key_exact=bool(rng.integers(0, 2)), darboux=bool(rng.integers(0, 2)), radius=float(rng.choice([0.0, 0.3, 0.5, 0.9])),
                rot_relative=bool(rng.integers(0, 2)), alpha_2pi=bool(rng.integers(0, 2)),
                pos_thr=float(rng.choice([-1.0, 0.03])), rot_thr=float(rng.choice([-1.0, 0.35])),
                max_tile_refs=int(rng.choice([0, 0, 90, 300])), acc32=int(rng.choice([0, 0, 1])))


@pytest.mark.parametrize("seed", soak_seeds(12, "PPF_SOAK_POLICY"))
def test_random_policy_configuration(bottle, seed):
    """the same bar under drawn combinations of the PCL-semantics switches (exact keys, Darboux feature, pair radius, relative
    rotation metric, 2 pi alpha range: tests/test_gpu_policy.py has each alone), with forced tiling and 32-bit cells mixed in"""
    import torch
    from yolo_ppf_pose_estimation_amd import _capi
    from yolo_ppf_pose_estimation_amd.device import Workspace
    cfg = _draw_policy(seed)
    model = bottle if cfg["kind"] == "bottle" else synth.make_solid(cfg["kind"], 6000, seed=seed + 11)
    det = PPF3DDetector(cfg["train_step"], 0.05, cfg["num_angles"], max_tile_refs=cfg["max_tile_refs"],
                        key_equality=int(cfg["key_exact"]), feature=int(cfg["darboux"])).trainModel(model)
    radius = cfg["radius"] * det.info()["diameter"]
    det.setPolicy(pair_radius=radius, rot_metric_relative=cfg["rot_relative"], alpha_range_2pi=cfg["alpha_2pi"])
    det.setSearchParams(cfg["pos_thr"], cfg["rot_thr"])
    ora = O.OracleDetector(cfg["train_step"], 0.05, cfg["num_angles"]).train_model(model, darboux=cfg["darboux"])
    ora.set_policy(key_exact=cfg["key_exact"], pair_radius=radius, rot_relative=cfg["rot_relative"], alpha_2pi=cfg["alpha_2pi"])
    ora.set_search_params(cfg["pos_thr"], cfg["rot_thr"])
    scene, _ = synth.make_scene(model, n_points=cfg["n_scene"], seed=cfg["scene_seed"])
    want = ora.match(scene, relative_scene_sample_step=cfg["ref_step"], presampled=True)
    ws = Workspace()
    ws.set_option(_capi.PPF_OPT_ACC32, cfg["acc32"])
    d = torch.from_numpy(np.ascontiguousarray(scene, dtype=np.float32)).cuda()
    ws.match_device(det, d.data_ptr(), scene.shape[0], 6, cfg["ref_step"], 0.05, presampled=True, skip_clustering=True)
    res = ws.results(scene.shape[0])
    np.testing.assert_array_equal(res["triples"], want["triples"], err_msg=str(cfg))
    assert res["stats"]["n_votes"] == int(want["votes_per_ref"].sum()), cfg
    assert res["stats"]["n_pairs"] == int(want["pairs_per_ref"].sum()), cfg
    poses = det.match(scene, cfg["ref_step"], 0.05, presampled=True)
    assert len(poses) == want["n_final"], cfg
    for g, w in zip(poses[:10], want["poses"][:10]):
        assert g.numVotes == w["num_votes"], cfg
        np.testing.assert_allclose(g.pose, w["pose"], rtol=0, atol=1e-9, err_msg=str(cfg))


def _draw_degenerate(seed):
    rng = np.random.default_rng(3000 + seed)
    return dict(kind=["bottle", "box", "cylinder", "plane"][int(rng.integers(0, 4))],
                model_rows=int(rng.choice([12, 60, 300, 1500])), num_angles=int(rng.choice([5, 7, 16, 30, 31, 32, 33, 64])),
                dist_step=float(rng.choice([0.02, 0.05, 0.3])), n_scene=int(rng.choice([1, 2, 7, 64, 65, 700, 2500])),
                scene_seed=int(rng.integers(0, 10 ** 6)), ref_step=float(rng.choice([1.0, 1.0 / 3.0, 1.0 / 10.0])),
                duplicates=bool(rng.integers(0, 2)), zero_normals=bool(rng.integers(0, 3) == 0), long_normals=bool(rng.integers(0, 3) == 0),
                non_finite=bool(rng.integers(0, 3) == 0), s2b=bool(rng.integers(0, 3) == 0), key_exact=bool(rng.integers(0, 4) == 0),
                max_tile_refs=int(rng.choice([0, 0, 5, 64])))


def _bounded(cfg):
    """a flat model sends every pair into a handful of buckets: the oracle then casts model rows^2 votes per scene pair"""
    if cfg["kind"] in ("plane", "box") and cfg["model_rows"] > 300:
        cfg["n_scene"] = min(cfg["n_scene"], 700)
    if cfg["kind"] == "plane" and cfg["model_rows"] > 300:
        cfg["model_rows"] = 300
    return cfg


@pytest.mark.parametrize("seed", soak_seeds(10, "PPF_SOAK_DEGENERATE"))
def test_random_degenerate_inputs(bottle, seed):
    """inputs a real crop can hold and a test cloud rarely does, drawn together: models of a dozen rows, flat models (every
    pair of a face has parallel normals: acos arguments a rounding above 1, NaN angle bins), scenes of one or two rows, rows
    that exist twice, zero and over-long normals, a few non-finite coordinates or normals (such a pair's alpha is NaN and the
    reference's loop skips it), odd alpha resolutions around the count tables' limit (31 / 32), coarse and fine distance
    steps: vote triples, vote and pair totals and the raw poses equal the oracle's."""
    cfg = _bounded(_draw_degenerate(seed))
    rng = np.random.default_rng(seed)
    if cfg["kind"] == "plane":
        uv = rng.uniform(-0.1, 0.1, size=(4000, 2))
        full = np.zeros((4000, 6), np.float32)
        full[:, :2] = uv
        full[:, 2] = 0.5
        full[:, 5] = 1.0
    else:
        full = bottle if cfg["kind"] == "bottle" else synth.make_solid(cfg["kind"], 6000, seed=seed + 5)
    model = full[rng.permutation(full.shape[0])[:cfg["model_rows"]]].copy()
    det = PPF3DDetector(0.05, cfg["dist_step"], cfg["num_angles"], max_tile_refs=cfg["max_tile_refs"], key_equality=int(cfg["key_exact"]))
    det.trainModel(model, presampled=True)
    ora = O.OracleDetector(0.05, cfg["dist_step"], cfg["num_angles"]).train_model(model, presampled=True).set_policy(key_exact=cfg["key_exact"])
    T = synth.rigid_pose(cfg["scene_seed"] % 997, 0.2)
    scene = synth.apply_pose(full[rng.permutation(full.shape[0])[:cfg["n_scene"]]], T).astype(np.float32)
    n = scene.shape[0]
    if cfg["duplicates"] and n > 1:
        scene[rng.integers(0, n, max(1, n // 10))] = scene[rng.integers(0, n, max(1, n // 10))]
    if cfg["zero_normals"]:
        scene[rng.integers(0, n, max(1, n // 20)), 3:] = 0.0
    if cfg["long_normals"]:
        scene[rng.integers(0, n, max(1, n // 20)), 3:] *= 3.0
    if cfg["non_finite"]:
        for r in rng.integers(0, n, max(1, n // 50)):
            scene[r, int(rng.integers(0, 6))] = [np.nan, np.inf, -np.inf][int(rng.integers(0, 3))]
    edge = scene[rng.random(n) < 0.5] if cfg["s2b"] else None
    if edge is not None and edge.shape[0] == 0:
        edge = scene[:1]
    got = det.raw_votes(scene, cfg["ref_step"], 0.05, presampled=True, edge=edge)
    want = ora.match(scene, edge=edge, relative_scene_sample_step=cfg["ref_step"], presampled=True, cluster=False)
    assert got["n_ref"] == want["n_ref"], cfg
    np.testing.assert_array_equal(got["triples"], want["triples"], err_msg=str(cfg))
    assert got["stats"]["n_votes"] == int(want["votes_per_ref"].sum()), cfg
    assert got["stats"]["n_pairs"] == int(want["pairs_per_ref"].sum()), cfg
    for g, w in zip(got["raw_poses"], want["raw_poses"]):
        assert np.array_equal(g.pose, w["pose"], equal_nan=True), cfg


@pytest.mark.parametrize("seed", soak_seeds(4, "PPF_SOAK_BATCH"))
def test_random_batches_equal_single_matches(bottle, seed):
    """ppf_match_batch on drawn batches (1 - 4 models of different shapes, sampling steps and alpha resolutions; 1 - 7 crops of
    ragged sizes, a crop of a handful of rows among them; presampled or not; lanes that do not divide the crops): every
    (crop, model) cell equals the single match of the same detector on the same crop -- which the sweeps above tie to the oracle"""
    from yolo_ppf_pose_estimation_amd.detector import match_batch
    rng = np.random.default_rng(11000 + seed)
    nm, nc = int(rng.integers(1, 5)), int(rng.integers(1, 8))
    dets, fulls = [], []
    pos_thr, rot_thr = float(rng.choice([-1.0, 0.03])), float(rng.choice([-1.0, 0.35]))   # one ppf_match_params per batch
    for k in range(nm):
        kind = KINDS[int(rng.integers(0, 4))]
        full = bottle if kind == "bottle" else synth.make_solid(kind, 6000, seed=seed * 7 + k)
        det = PPF3DDetector(float(rng.choice([0.05, 0.08, 0.1])), 0.05, int(rng.choice([15, 30, 40])))
        det.setSearchParams(pos_thr, rot_thr)
        dets.append(det.trainModel(full))
        fulls.append(full)
    crops = []
    for c in range(nc):
        n = int(rng.choice([3, 40, 700, 1500, 4000]))
        crops.append(synth.make_scene(fulls[int(rng.integers(0, nm))], n_points=max(n, 300), seed=int(rng.integers(0, 10 ** 6)))[0][:n])
    presampled, step, top_k = bool(rng.integers(0, 2)), float(rng.choice([1.0 / 3.0, 1.0 / 10.0])), int(rng.integers(1, 7))
    got = match_batch(dets, crops, step, 0.05, presampled=presampled, top_k=top_k)
    for c, crop in enumerate(crops):
        for k, det in enumerate(dets):
            want = det.match(crop, step, 0.05, presampled=presampled)[:top_k]
            assert len(got[c][k]) == len(want), (seed, c, k)
            for g, w in zip(got[c][k], want):
                assert g.numVotes == w.numVotes and g.modelIndex == w.modelIndex
                np.testing.assert_array_equal(g.pose, w.pose)
