"""CPU checks of the ICP oracle (oracle/ppf_icp_oracle.cpp; restates cv::ppf_match_3d::ICP::registerModelToScene as
called at /root/reference/include/CloudProcessing.h:465-470, :518-523).  Parity unpinned: the reference ships no ICP
fixtures, so these are property tests of the restated algorithm (convergence to a known rigid motion, identity fixed
point, determinism, independence of the poses in one call)."""
import numpy as np
import pytest

import oracle_lib as O
from yolo_ppf_pose_estimation_amd import synth

BOTTLE = "tests/golden/bottle_model_xyzn.npy"


def _perturb(T, deg, shift):
    d = np.eye(4)
    a = np.deg2rad(deg)
    d[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
    d[:3, 3] = shift
    return d @ T


@pytest.fixture(scope="module")
def clouds():
    bottle = np.load(BOTTLE)
    T = synth.rigid_pose(5, 0.1)
    return bottle[::8].copy(), synth.apply_pose(bottle[3::5], T), T


def test_icp_recovers_known_motion(clouds):
    model, scene, T = clouds
    P, res, its = O.icp_refine(model, scene, [_perturb(T, 4.0, [0.004, -0.003, 0.002])])
    assert np.abs(P[0] - T).max() < 5e-4
    assert res[0] < 0.01 and 0 < its[0] <= 272  # sum over 8 levels of round(100/(level+1))
    R = P[0][:3, :3]
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-9)


def test_icp_fixed_point_and_determinism(clouds):
    model, scene, T = clouds
    P1, r1, i1 = O.icp_refine(model, scene, [T])
    P2, r2, i2 = O.icp_refine(model, scene, [T])
    assert np.array_equal(P1, P2) and np.array_equal(r1, r2) and np.array_equal(i1, i2)
    assert np.abs(P1[0] - T).max() < 5e-4


def test_icp_poses_are_independent(clouds):
    model, scene, T = clouds
    a, b = _perturb(T, 3.0, [0.002, 0.0, 0.0]), _perturb(T, -2.0, [0.0, 0.003, -0.001])
    Pab, rab, _ = O.icp_refine(model, scene, [a, b])
    Pa, ra, _ = O.icp_refine(model, scene, [a])
    Pb, rb, _ = O.icp_refine(model, scene, [b])
    assert np.array_equal(Pab[0], Pa[0]) and np.array_equal(Pab[1], Pb[0])
    assert rab[0] == ra[0] and rab[1] == rb[0]


def test_icp_without_rejection_and_single_level(clouds):
    model, scene, T = clouds
    P, res, its = O.icp_refine(model, scene, [_perturb(T, 2.0, [0.001, 0.001, 0.0])], iterations=30, tolerance=0.005,
                               rejection_scale=0.0, num_levels=1)
    assert its[0] <= 30
    assert np.abs(P[0] - T).max() < 2e-3
