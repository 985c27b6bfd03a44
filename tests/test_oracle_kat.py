"""Known-answer tests that pin the CPU oracle (the reference has no golden vectors for
this path, so these are what stands between the restatement and "anything goes").  CPU only."""
import os

import numpy as np
import pytest

import oracle_lib as O
from yolo_ppf_pose_estimation_amd import synth

GOLDEN = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.npz"))


def _rand_unit(rng, n=1):
    v = rng.normal(size=(n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


# ---- (v) MurmurHash3_x64_128 public vectors ---------------------------------------------------
@pytest.mark.parametrize("data,seed,h1,h2", [
    (b"", 0, 0x0, 0x0),
    (b"hello", 0, 0xCBD8A7B341BD9B02, 0x5B1E906A48AE1D19),
    (b"The quick brown fox jumps over the lazy dog", 0, 0xE34BBC7BBC071B6C, 0x7A433CA9C49A9347),
    # mmh3's documented hash64("foo") == (-2129773440516405919, 9128664383759220103), as unsigned words:
    (b"foo", 0, 2 ** 64 - 2129773440516405919, 9128664383759220103),
])
def test_murmur3_x64_128_public_vectors(data, seed, h1, h2):
    assert O.murmur3_x64_128(data, seed) == (h1, h2)


def test_hash_is_low_word_of_h1_with_seed_42():
    f, key, h = O.pair_feature([0, 0, 0], [1, 0, 0], [0.05, 0.02, 0], [0, 1, 0], 0.2094395, 0.006)
    h1, _ = O.murmur3_x64_128(key.astype("<i4").tobytes(), 42)
    assert h == (h1 & 0xFFFFFFFF)


# ---- feature definition, invariance -----------------------------------------------------------
def test_pair_feature_definition():
    p1, n1 = np.array([0, 0, 0], np.float32), np.array([0, 0, 1], np.float32)
    p2, n2 = np.array([0.3, 0, 0.4], np.float32), np.array([1, 0, 0], np.float32)
    f, key, _ = O.pair_feature(p1, n1, p2, n2, np.deg2rad(12), 0.05, mode=O.MODE_LIBM)
    assert f[3] == pytest.approx(0.5, rel=1e-7)
    assert f[0] == pytest.approx(np.arccos(0.8), rel=1e-7)   # angle(n1, d)
    assert f[1] == pytest.approx(np.arccos(0.6), rel=1e-7)   # angle(n2, d)
    assert f[2] == pytest.approx(np.pi / 2, rel=1e-7)        # angle(n1, n2)
    assert list(key) == [int(f[0] / np.deg2rad(12)), int(f[1] / np.deg2rad(12)), int(f[2] / np.deg2rad(12)), 10]


def test_pair_feature_invariant_under_rigid_motion():
    rng = np.random.default_rng(4)
    for s in range(50):
        p = rng.uniform(-0.2, 0.2, size=(2, 3)).astype(np.float32)
        n = _rand_unit(rng, 2)
        T = synth.rigid_pose(100 + s)
        c = np.concatenate([p, n], axis=1)
        cm = synth.apply_pose(c, T)
        f0, _, _ = O.pair_feature(c[0, :3], c[0, 3:], c[1, :3], c[1, 3:], 0.2094395, 0.01)
        f1, _, _ = O.pair_feature(cm[0, :3], cm[0, 3:], cm[1, :3], cm[1, 3:], 0.2094395, 0.01)
        np.testing.assert_allclose(f0, f1, atol=2e-5)  # float32 clouds: invariance up to input rounding


# ---- PCL's pair feature (policy switch) -------------------------------------------------------------
def _pcl_pair_features(p1, n1, p2, n2):
    """pcl::computePairFeatures as published (features/src/pfh.cpp), restated in numpy fp64 for the known-answer check."""
    p1, n1, p2, n2 = (np.asarray(v, np.float64) for v in (p1, n1, p2, n2))
    dp = p2 - p1
    f4 = np.linalg.norm(dp)
    if f4 == 0:
        return None
    a1, a2 = n1 @ dp / f4, n2 @ dp / f4
    if np.arccos(abs(a1)) > np.arccos(abs(a2)):
        n1, n2, dp, f3 = n2, n1, -dp, -a2
    else:
        f3 = a1
    v = np.cross(dp, n1)
    if np.linalg.norm(v) == 0:
        return None
    v /= np.linalg.norm(v)
    w = np.cross(n1, v)
    return np.array([np.arctan2(w @ n2, n1 @ n2), v @ n2, f3, f4])


def test_darboux_feature_definition():
    step = np.deg2rad(12)
    # source normal +z, target 0.5 away in the x-z plane, target normal +x: u = n1, v = d x u / |.| = -y, w = u x v = +x
    f, key, h = O.pair_feature_darboux([0, 0, 0], [0, 0, 1], [0.3, 0, 0.4], [1, 0, 0], step, 0.05, mode=O.MODE_LIBM)
    np.testing.assert_allclose(f, [np.pi / 2, 0.0, 0.8, 0.5], atol=1e-12)
    assert list(key) == [7, 0, 3, 10]                      # floor(90/12), floor(0), floor(0.8/0.2094), floor(0.5/0.05)
    h1, _ = O.murmur3_x64_128(key.astype("<i4").tobytes(), 42)
    assert h == (h1 & 0xFFFFFFFF)
    # the point whose normal is closer to the connecting line becomes the source: swapping the arguments gives the same feature
    g, gkey, _ = O.pair_feature_darboux([0.3, 0, 0.4], [1, 0, 0], [0, 0, 0], [0, 0, 1], step, 0.05, mode=O.MODE_LIBM)
    np.testing.assert_allclose(g, f, atol=1e-12)
    # negative values floor downwards (PCL's std::floor), they do not truncate towards zero
    f, key, _ = O.pair_feature_darboux([0, 0, 0], [0, 0, 1], [0.3, 0, 0.4], [-0.6, 0.8, 0], step, 0.05, mode=O.MODE_LIBM)
    assert f[0] < 0 and f[1] < 0 and key[0] == np.floor(f[0] / step) and key[1] == np.floor(f[1] / step) == -4


def test_darboux_feature_against_the_published_formula_and_rigid_motion():
    rng = np.random.default_rng(14)
    for s in range(200):
        p = rng.uniform(-0.2, 0.2, size=(2, 3)).astype(np.float32)
        n = _rand_unit(rng, 2)
        want = _pcl_pair_features(p[0], n[0], p[1], n[1])
        for mode in (O.MODE_LIBM, O.MODE_DET):
            f, _, _ = O.pair_feature_darboux(p[0], n[0], p[1], n[1], 0.2094395, 0.01, mode=mode)
            np.testing.assert_allclose(f, want, atol=1e-12)
        c = np.concatenate([p, n], axis=1)
        cm = synth.apply_pose(c, synth.rigid_pose(300 + s))
        f1, _, _ = O.pair_feature_darboux(cm[0, :3], cm[0, 3:], cm[1, :3], cm[1, 3:], 0.2094395, 0.01)
        d = np.abs(f1 - want)
        d[0] = min(d[0], 2 * np.pi - d[0])   # atan2 branch cut
        assert d.max() < 5e-5                # float32 clouds: invariance up to input rounding


def test_darboux_degenerate_pairs_have_no_feature():
    assert O.pair_feature_darboux([0.1, 0.2, 0.3], [1, 0, 0], [0.1, 0.2, 0.3], [0, 1, 0], 0.2, 0.01) is None   # same point
    assert O.pair_feature_darboux([0, 0, 0], [0, 0, 1], [0, 0, 0.5], [0, 0, 1], 0.2, 0.01) is None             # d parallel to u
    assert _pcl_pair_features([0, 0, 0], [0, 0, 1], [0, 0, 0.5], [0, 0, 1]) is None


def test_darboux_self_match_recovers_the_pose():
    rng = np.random.default_rng(8)
    pts = rng.uniform(-0.1, 0.1, size=(220, 3)) * np.array([1.0, 0.6, 0.3])
    cloud = np.concatenate([pts, _rand_unit(rng, 220)], axis=1).astype(np.float32)
    T = synth.rigid_pose(77)
    scene = synth.apply_pose(cloud, T)
    ora = O.OracleDetector(0.05, 0.05).train_model(cloud, presampled=True, darboux=True).set_policy(key_exact=True)
    res = ora.match(scene, relative_scene_sample_step=0.2, presampled=True)
    np.testing.assert_allclose(res["poses"][0]["pose"], T, atol=0.12)
    assert res["poses"][0]["num_votes"] > 0


def test_degenerate_pair_keeps_zero_feature():
    f, key, _ = O.pair_feature([0.1, 0.2, 0.3], [1, 0, 0], [0.1, 0.2, 0.3], [0, 1, 0], 0.2094395, 0.01)
    assert list(f) == [0, 0, 0, 0] and list(key) == [0, 0, 0, 0]


# ---- frame transform + alpha composition identity -----------------------------------------------
def test_transform_rt_maps_point_to_origin_and_normal_to_x():
    rng = np.random.default_rng(5)
    for _ in range(100):
        p = rng.uniform(-1, 1, 3).astype(np.float32)
        n = _rand_unit(rng)[0]
        R, t = O.transform_rt(p, n)
        np.testing.assert_allclose(R @ p.astype(np.float64) + t, 0, atol=1e-12)
        np.testing.assert_allclose(R @ n.astype(np.float64), [1, 0, 0], atol=1e-6)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)
    R, t = O.transform_rt([0, 0, 0], [1, 0, 0])       # n.y == n.z == 0 branch: axis (0,1,0), angle 0
    np.testing.assert_allclose(R, np.eye(3), atol=0)
    R, t = O.transform_rt([0, 0, 0], [-1, 0, 0])      # angle pi about y
    np.testing.assert_allclose(R, np.diag([-1.0, 1.0, -1.0]), atol=1e-15)


def test_alpha_composition_identity():
    """T_sg^-1 . Rx(alpha_m - alpha_s) . T_mg maps a model pair onto the same pair moved rigidly."""
    rng = np.random.default_rng(6)
    for s in range(30):
        pm = rng.uniform(-0.1, 0.1, size=(2, 3)).astype(np.float32)
        nm = _rand_unit(rng, 2)
        T = synth.rigid_pose(200 + s)
        sc = synth.apply_pose(np.concatenate([pm, nm], axis=1), T)
        am = O.alpha(pm[0], nm[0], pm[1])
        a_s = O.alpha(sc[0, :3], sc[0, 3:], sc[1, :3])
        Rm, tm = O.transform_rt(pm[0], nm[0])
        Rs, ts = O.transform_rt(sc[0, :3], sc[0, 3:])
        a = am - a_s
        Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        q = Rx @ (Rm @ pm[1].astype(np.float64) + tm)
        back = Rs.T @ (q - ts)
        np.testing.assert_allclose(back, sc[1, :3].astype(np.float64), atol=2e-6)


def test_quaternion_roundtrip():
    rng = np.random.default_rng(7)
    for _ in range(200):
        R = synth.random_rotation(rng)
        q = O.dcm_to_quat(R)
        assert abs(np.linalg.norm(q) - 1) < 1e-12
        np.testing.assert_allclose(O.quat_to_dcm(q), R, atol=1e-12)


# ---- (i) regression constants measured from the reference's model file ----------------------------
def test_bottle_sampling_counts(bottle):
    assert bottle.shape == (19753, 6)
    ext = bottle[:, :3].max(0) - bottle[:, :3].min(0)
    assert float(np.linalg.norm(ext.astype(np.float32))) == pytest.approx(0.23919, abs=1e-5)
    counts = [O.sample(bottle, s).shape[0] for s in GOLDEN["bottle_sample_steps"]]
    np.testing.assert_array_equal(counts, GOLDEN["bottle_sample_counts"])
    # SURVEY.md §6 [MEASURED]: 0.035 -> 2,140, 0.036 -> 2,000, 0.0714 -> 585.  (0.025 and 0.05 give 3,870 /
    # 1,038 here rather than the survey's 3,949 / 1,130: the library's sampleStep parameter is a float, and
    # (int)(1/0.025f) = 39, (int)(1/0.05f) = 19 cells per axis; the survey's numpy used 40 / 20.)
    assert dict(zip(GOLDEN["bottle_sample_steps"].tolist(), counts))[0.035] == 2140
    assert dict(zip(GOLDEN["bottle_sample_steps"].tolist(), counts))[0.036] == 2000
    assert dict(zip(GOLDEN["bottle_sample_steps"].tolist(), counts))[0.0714] == 585


def test_sampling_rules():
    # cells in ascending index order; mean position; summed-then-normalised normal
    pc = np.array([[0.9, 0.9, 0.9, 0, 0, 1], [0.0, 0.0, 0.0, 1, 0, 0], [0.01, 0.0, 0.0, 0, 1, 0],
                   [1.0, 1.0, 1.0, 0, 0, 1]], np.float32)
    s = O.sample(pc, 0.5)  # 2 cells per axis
    assert s.shape[0] == 3
    np.testing.assert_allclose(s[0, :3], [0.005, 0, 0], atol=1e-7)
    np.testing.assert_allclose(s[0, 3:], [2 ** -0.5, 2 ** -0.5, 0], atol=1e-7)
    np.testing.assert_allclose(s[1, :3], [0.9, 0.9, 0.9], atol=1e-7)   # cell (1,1,1) -> index 7
    np.testing.assert_allclose(s[2, :3], [1.0, 1.0, 1.0], atol=1e-7)   # the max corner: cell (2,2,2) -> index 14


# ---- (iv) tiny hand-sized case with full accumulator dumps ------------------------------------------
def test_tiny_case_against_committed_vectors():
    det = O.OracleDetector(0.05, 0.05).train_model(GOLDEN["tiny_model"], presampled=True)
    info = det.info()
    assert [info["slots"], info["num_angles"]] == GOLDEN["tiny_info"].tolist() == [64, 30]
    hsh, alp = det.pairs()
    np.testing.assert_array_equal(hsh, GOLDEN["tiny_pair_hash"])
    np.testing.assert_array_equal(alp, GOLDEN["tiny_pair_alpha"])
    scene = GOLDEN["tiny_scene"]
    for i in range(scene.shape[0]):
        acc = det.accumulator(scene, i)
        np.testing.assert_array_equal(acc, GOLDEN["tiny_acc"][i])
        assert acc.sum() > 0
    r = det.match(scene, relative_scene_sample_step=1.0, presampled=True)
    np.testing.assert_array_equal(r["triples"], GOLDEN["tiny_triples"])
    # every vote of a tiny self-similar scene lands somewhere: totals match the accumulators
    np.testing.assert_array_equal(r["votes_per_ref"], GOLDEN["tiny_acc"].reshape(scene.shape[0], -1).sum(1))


def test_bottle_case_against_committed_vectors(bottle):
    det = O.OracleDetector(0.07, 0.05).train_model(bottle)
    np.testing.assert_array_equal(det.sampled_model(), GOLDEN["b07_sampled_model"])
    bs = det.bucket_stats()
    assert [bs["non_empty"], bs["max_len"]] == GOLDEN["b07_bucket_stats"].tolist()
    scene, _ = synth.make_scene(bottle, n_points=int(GOLDEN["b07_scene_seed"][1]), seed=int(GOLDEN["b07_scene_seed"][0]))
    r = det.match(scene, relative_scene_sample_step=1.0 / 10.0, presampled=True)
    np.testing.assert_array_equal(r["triples"], GOLDEN["b07_triples"])
    np.testing.assert_array_equal(r["votes_per_ref"], GOLDEN["b07_votes"])
    np.testing.assert_array_equal(r["pairs_per_ref"], GOLDEN["b07_pairs"])
    assert r["n_final"] == int(GOLDEN["b07_n_final"][0])
    for k in range(5):
        np.testing.assert_array_equal(r["poses"][k]["pose"], GOLDEN["b07_top_poses"][k])
        assert r["poses"][k]["num_votes"] == GOLDEN["b07_top_votes"][k]


# ---- (ii) self-match: known pose recovered -------------------------------------------------------------
def test_self_match_recovers_pose_of_asymmetric_object():
    """An asymmetric solid (two boxes glued into an L, plus a cylinder stub) moved rigidly: the top pose
    must bring the model onto the scene (the bottle itself is near-rotationally-symmetric, so it only pins
    the pose up to that symmetry and is not used here)."""
    a = synth.make_solid("box", 6000, seed=1)
    b = synth.make_solid("box", 3000, seed=2)
    b[:, 0] += 0.08; b[:, 2] += 0.14
    c = synth.make_solid("cylinder", 2000, seed=3)
    c[:, 1] += 0.1; c[:, 2] -= 0.02
    model = np.concatenate([a, b, c])
    det = O.OracleDetector(0.05, 0.05).train_model(model)
    T = synth.rigid_pose(31)
    scene = synth.apply_pose(model[::2], T)
    r = det.match(scene, relative_scene_sample_step=1.0 / 5.0, relative_scene_distance=0.05)
    P = r["poses"][0]["pose"]
    moved = model[:, :3].astype(np.float64) @ P[:3, :3].T + P[:3, 3]
    want = model[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    err = np.linalg.norm(moved - want, axis=1)
    diameter = np.linalg.norm(model[:, :3].max(0) - model[:, :3].min(0))
    assert np.median(err) < 0.1 * diameter  # within the voting resolution (no ICP refinement here)
    assert r["poses"][0]["num_votes"] >= r["poses"][-1]["num_votes"]


# ---- (iii) match_S2B identity ------------------------------------------------------------------------------
def test_s2b_with_edge_equal_scene_reduces_to_match(bottle):
    det = O.OracleDetector(0.07, 0.05).train_model(bottle)
    scene, _ = synth.make_scene(bottle, n_points=1200, seed=9)
    a = det.match(scene, relative_scene_sample_step=0.1, presampled=True)
    b = det.match(scene, edge=scene, relative_scene_sample_step=0.1, presampled=True)
    np.testing.assert_array_equal(a["triples"], b["triples"])
    np.testing.assert_array_equal(a["votes_per_ref"], b["votes_per_ref"])
    for pa, pb in zip(a["poses"], b["poses"]):
        np.testing.assert_array_equal(pa["pose"], pb["pose"])


# ---- reference quirks frozen in the spec ------------------------------------------------------------------------
def test_alpha_bin_spill_and_thresholds(bottle):
    det = O.OracleDetector(0.07, 0.05).train_model(bottle)
    info = det.info()
    assert info["angle_step"] == (360.0 / 30) * np.pi / 180.0
    # distance step uses the SAMPLING step (library quirk), float-rounded
    diameter = np.float32(np.linalg.norm((bottle[:, :3].max(0) - bottle[:, :3].min(0)).astype(np.float32)))
    assert info["distance_step"] == float(np.float32(float(diameter) * 0.07))
    det2 = O.OracleDetector(0.07, 0.05, dist_from_distance_step=True).train_model(bottle)
    assert det2.info()["distance_step"] == float(np.float32(float(diameter) * 0.05))
    # fewer sampled rows than the reference stride -> 1 raw pose, 0 clustered poses (rows / step == 0)
    scene, _ = synth.make_scene(bottle, n_points=15, seed=3)
    r = det.match(scene, relative_scene_sample_step=1.0 / 20.0, presampled=True)
    assert r["n_ref"] == 1 and r["n_final"] == 0


def test_det_and_libm_modes_agree_on_votes(bottle):
    """libm mode = what an upstream build would compute; det mode = the frozen spec.  Same triples here."""
    scene, _ = synth.make_scene(bottle, n_points=1000, seed=12)
    a = O.OracleDetector(0.07, 0.05, mode=O.MODE_DET).train_model(bottle).match(
        scene, relative_scene_sample_step=0.1, presampled=True, cluster=False)
    b = O.OracleDetector(0.07, 0.05, mode=O.MODE_LIBM).train_model(bottle).match(
        scene, relative_scene_sample_step=0.1, presampled=True, cluster=False)
    same = (a["triples"] == b["triples"]).all(axis=1).mean()
    assert same >= 0.98  # differences, if any, are single last-ulp bin flips
    assert abs(int(a["votes_per_ref"].sum()) - int(b["votes_per_ref"].sum())) <= 1e-4 * int(a["votes_per_ref"].sum())


def test_tiny_case_accumulators_from_an_independent_numpy_voter():
    """Second source for the 6-point model / 8-point scene of oracle_golden.npz: a voter written from SURVEY.md section 8a alone
    (feature -> 4 ints -> MurmurHash3_x64_128 low word -> `hash % slots` bucket walk without key comparison -> alpha bin over
    4 pi -> accumulator -> strict-> argmax), plain Python / numpy, nothing shared with oracle_lib or the C++ oracle.  It must
    reproduce the oracle's FULL accumulator dump and vote triples.  (Says nothing about the reference -- parity stays
    unpinned -- but a silent slip of the oracle can no longer redefine "green".)"""
    import math
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_golden.npz"))
    model, scene = g["tiny_model"].astype(np.float64), g["tiny_scene"].astype(np.float64)
    A, M64 = 30, (1 << 64) - 1
    astep = (360.0 / A) * math.pi / 180.0
    ext = g["tiny_model"][:, :3].max(0) - g["tiny_model"][:, :3].min(0)
    dstep = float(np.float32(math.sqrt(float((ext.astype(np.float64) ** 2).sum())) * 0.05))
    slots = max(16, 1 << (len(model) ** 2 - 1).bit_length())
    rotl = lambda x, r: ((x << r) | (x >> (64 - r))) & M64

    def fmix(k):
        k ^= k >> 33; k = k * 0xff51afd7ed558ccd & M64; k ^= k >> 33; k = k * 0xc4ceb9fe1a85ec53 & M64
        return k ^ (k >> 33)

    def murmur_low32(keys):  # MurmurHash3_x64_128 of four int32 (16 bytes, little endian), seed 42: low word of h1
        c1, c2, h1, h2 = 0x87c37b91114253d5, 0x4cf5ad432745937f, 42, 42
        w = [k & 0xFFFFFFFF for k in keys]
        k1, k2 = w[0] | w[1] << 32, w[2] | w[3] << 32
        k1 = rotl(k1 * c1 & M64, 31) * c2 & M64; h1 ^= k1; h1 = (rotl(h1, 27) + h2) & M64; h1 = (h1 * 5 + 0x52dce729) & M64
        k2 = rotl(k2 * c2 & M64, 33) * c1 & M64; h2 ^= k2; h2 = (rotl(h2, 31) + h1) & M64; h2 = (h2 * 5 + 0x38495ab5) & M64
        h1 ^= 16; h2 ^= 16; h1 = (h1 + h2) & M64; h2 = (h2 + h1) & M64
        h1, h2 = fmix(h1), fmix(h2)
        return ((h1 + h2) & M64) & 0xFFFFFFFF

    def slot_of(p1, n1, p2, n2):
        d = p2 - p1; f3 = math.sqrt(float(d @ d)); d = d / f3
        f = [math.acos(float(n1 @ d)), math.acos(float(n2 @ d)), math.acos(float(n1 @ n2))]
        return murmur_low32([int(v / astep) for v in f] + [int(f3 / dstep)]) % slots

    def frame(p, n):  # rotation taking n onto +x (Rodrigues about (0, n.z, -n.y)), t = -R p
        ang, ax = math.acos(n[0]), np.array([0.0, n[2], -n[1]])
        ax = ax / np.linalg.norm(ax) if (n[1] != 0 or n[2] != 0) else np.array([0.0, 1.0, 0.0])
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = math.cos(ang) * np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * np.outer(ax, ax)
        return R, -R @ p

    def alpha(R, t, p2):
        q = t + R @ p2
        a = math.atan2(-q[2], q[1])
        return -(-a if math.sin(a) * q[2] < 0 else a)

    table = {}
    for i in range(len(model)):
        R, t = frame(model[i, :3], model[i, 3:])
        for j in range(len(model)):
            if j != i:
                table.setdefault(slot_of(model[i, :3], model[i, 3:], model[j, :3], model[j, 3:]), []).append(
                    (i, float(np.float32(alpha(R, t, model[j, :3])))))
    acc = np.zeros((len(scene), len(model) * A), dtype=np.uint32)
    for i in range(len(scene)):
        R, t = frame(scene[i, :3], scene[i, 3:])
        for j in range(len(scene)):
            if j != i:
                a_s = alpha(R, t, scene[j, :3])
                for im, am in table.get(slot_of(scene[i, :3], scene[i, 3:], scene[j, :3], scene[j, 3:]), []):
                    flat = im * A + int(A * (am - a_s + 2 * math.pi) / (4 * math.pi))
                    if flat < acc.shape[1]:
                        acc[i, flat] += 1
    np.testing.assert_array_equal(acc.reshape(len(scene), len(model), A), g["tiny_acc"])
    flat = acc.argmax(1)  # numpy's argmax returns the first maximum: model row ascending, bin ascending, strict >
    np.testing.assert_array_equal(np.stack([flat // A, flat % A, acc.max(1)], 1).astype(np.uint32), g["tiny_triples"])
    assert int(g["tiny_info"][0]) == slots and int(g["tiny_info"][1]) == A


def test_tiny_case_raw_poses_from_an_independent_composition():
    """Second source for the pose assembly: pose = T_sg^-1 * Rx(alpha_idx * 4 pi / A - 2 pi) * T_mg
    composed in numpy from the golden vote triples, against the oracle's per-reference-point poses (which the GPU path must equal
    bit for bit).  numpy's libm and the oracle's deterministic math differ in the last units: 1e-12."""
    import math
    import oracle_lib as O
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_golden.npz"))
    model, scene, A = g["tiny_model"], g["tiny_scene"], 30

    def frame(p, n):
        p, n = p.astype(np.float64), n.astype(np.float64)
        ang, ax = math.acos(n[0]), np.array([0.0, n[2], -n[1]])
        ax = ax / np.linalg.norm(ax) if (n[1] != 0 or n[2] != 0) else np.array([0.0, 1.0, 0.0])
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        T = np.eye(4)
        T[:3, :3] = math.cos(ang) * np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * np.outer(ax, ax)
        T[:3, 3] = -T[:3, :3] @ p
        return T

    det = O.OracleDetector(0.05, 0.05).train_model(model, presampled=True)
    r = det.match(scene, relative_scene_sample_step=1.0, presampled=True, cluster=False)
    np.testing.assert_array_equal(r["triples"], g["tiny_triples"])
    for i, (ref, idx, votes) in enumerate(g["tiny_triples"]):
        a = float(idx) * (4 * math.pi / A) - 2 * math.pi
        Rx = np.eye(4)
        Rx[1:3, 1:3] = [[math.cos(a), -math.sin(a)], [math.sin(a), math.cos(a)]]
        want = np.linalg.inv(frame(scene[i, :3], scene[i, 3:])) @ Rx @ frame(model[ref, :3], model[ref, 3:])
        got = np.asarray(r["raw_poses"][i]["pose"], dtype=np.float64).reshape(4, 4)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
        assert r["raw_poses"][i]["num_votes"] == int(votes)


def test_pose_clustering_from_an_independent_greedy_clusterer():
    """Second source for clusterPoses on the golden bottle case (150 voted poses): poses by votes
    descending (ties: input order), each joins the FIRST cluster whose first pose lies within the position threshold (the
    relative sampling step used as metres) and whose rotation angle differs by less than the (ineffective, ~30 rad) rotation
    threshold, else opens one; a cluster's votes are its members' sum, its translation their plain mean, clusters by votes
    descending.  Membership, vote sums, cluster count and translations against the oracle's clustered result (the rotation
    average goes through upstream's quaternion conventions and is left to the oracle)."""
    import oracle_lib as O
    from yolo_ppf_pose_estimation_amd import synth
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_golden.npz"))
    bottle = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bottle_model_xyzn.npy"))
    scene, _ = synth.make_scene(bottle, n_points=int(g["b07_scene_seed"][1]), seed=int(g["b07_scene_seed"][0]))
    det = O.OracleDetector(0.07, 0.05).train_model(bottle)
    r = det.match(scene, relative_scene_sample_step=1.0 / 10.0, presampled=True)
    raw = r["raw_poses"]
    n_poses = len(r["sampled_scene"]) // 10 if len(r["sampled_scene"]) else len(raw)  # rows / step, integer division
    order = sorted(range(len(raw)), key=lambda i: -raw[i]["num_votes"])  # stable: ties keep input order
    pos_thr, rot_thr = 0.07, (360.0 / (2 * np.pi / 30)) / 180.0 * np.pi
    clusters = []
    for i in order[:min(n_poses, len(order))]:
        for c in clusters:
            head = raw[c[0]]
            if np.linalg.norm(head["t"] - raw[i]["t"]) < pos_thr and abs(raw[i]["angle"] - head["angle"]) < rot_thr:
                c.append(i)
                break
        else:
            clusters.append([i])
    votes = [sum(raw[i]["num_votes"] for i in c) for c in clusters]
    corder = sorted(range(len(clusters)), key=lambda k: -votes[k])
    assert len(clusters) == r["n_final"] == int(g["b07_n_final"][0])
    for rank, k in enumerate(corder[:5]):
        assert votes[k] == r["poses"][rank]["num_votes"] == int(g["b07_top_votes"][rank])
        t_mean = np.sum([raw[i]["t"] for i in clusters[k]], axis=0) * (1.0 / len(clusters[k]))
        np.testing.assert_allclose(r["poses"][rank]["t"], t_mean, rtol=0, atol=1e-15)
        np.testing.assert_allclose(g["b07_top_poses"][rank][:3, 3], t_mean, rtol=0, atol=1e-15)
