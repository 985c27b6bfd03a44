"""The reference's whole post-YOLO chain on its own depth frame against the committed golden result
(tests/golden/c1_pipeline_golden.npz, generator make_c1_pipeline_golden.py): crop -> voxel grid -> outlier removal ->
normals -> edges -> Mats -> trainModel(0.025, 0.05) -> match_S2B(0.05, 0.05) -> top 5 -> ICP(100, 0.005, 2.5, 8).
CPU: the oracles reproduce the fixture exactly (pins the frozen specification across rounds).  GPU: the engine,
through the CloudProcessor mirror with every cloud resident on the device, gives the same numbers."""
import os
import sys

import numpy as np
import pytest

import prep_data as D

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "c1_pipeline_golden.npz"))


def test_oracle_chain_reproduces_the_golden_result(golden):
    import make_c1_pipeline_golden as G
    got = G.chain()
    for key in golden.files:
        np.testing.assert_array_equal(got[key], golden[key], err_msg=key)


@pytest.mark.gpu
def test_device_chain_matches_the_golden_result(golden, bottle):
    from yolo_ppf_pose_estimation_amd._capi import IcpParams, Pose, check, lib
    from yolo_ppf_pose_estimation_amd.cloud_processor import CloudProcessor, DeviceCloud
    import ctypes as C
    xyz, depth, box, intr = D.c1_frame()
    K = np.array([[intr[0], 0, intr[2]], [0, intr[1], intr[3]], [0, 0, 1.0]])
    cp = CloudProcessor(xyz, depth, [box], [39], [0], 0.025, 0.05)
    cp.SceneCropping(K)
    assert len(cp.objects[0]) == int(golden["n_crop"])
    cp.Subsampling(0.003)
    cp.OutlierProcessing(50, 1.0)
    cp.NormalEstimation(30)
    cp.EdgeExtraction(0.03)
    obj = cp.PointCloudXYZNormalToMat(cp.objects_with_normals[0], resident=True)
    edge = cp.PointCloudXYZNormalToMat(cp.objects_edges[0], resident=True)
    assert (len(obj), len(edge)) == (int(golden["n_object"]), int(golden["n_edge"]))
    assert obj.rows().astype(np.float64).sum() == float(golden["object_checksum"])
    assert edge.rows().astype(np.float64).sum() == float(golden["edge_checksum"])
    cp.LoadSingleModel(bottle, "bottle")
    cp.TrainDetector(0.025, 0.05)
    det = cp.detectors[0]
    assert det.info()["n_ref"] == int(golden["n_model_sampled"])
    # the match on resident clouds: all clustered poses, then the ICP of the top 5 (what Matching_S2B does)
    mp = det._params(0.05, 0.05, False)
    cap = len(obj) + 8
    out = (Pose * cap)()
    n = C.c_int(0)
    check(lib().ppf_match_clouds(det._model.ptr, obj._ptr, edge._ptr, C.byref(mp), out, cap, C.byref(n)))
    assert n.value == int(golden["n_clusters"])
    assert [out[i].num_votes for i in range(5)] == golden["top_votes"].tolist()
    for i in range(5):
        np.testing.assert_allclose(np.array(out[i].pose).reshape(4, 4), golden["match_poses"][i], rtol=0, atol=1e-12)
    prm = IcpParams()
    lib().ppf_default_icp_params(C.byref(prm))
    model_dev = DeviceCloud.upload(bottle)
    # start the ICP from the golden match poses so that its bit-exactness is tested on its own
    for i in range(5):
        out[i].pose[:] = golden["match_poses"][i].reshape(16).tolist()
    iters = (C.c_int * 5)()
    check(lib().ppf_icp_refine_clouds(model_dev._ptr, obj._ptr, C.byref(prm), out, 5, iters))
    assert list(iters) == golden["icp_iterations"].tolist()
    for i in range(5):
        np.testing.assert_array_equal(np.array(out[i].pose).reshape(4, 4), golden["icp_poses"][i])
        assert out[i].residual == golden["icp_residuals"][i]
    # and the mirror's one-call version returns the refined top pose
    pose = cp.Matching_S2B("bottle", obj, edge)
    np.testing.assert_allclose(pose.pose, golden["icp_poses"][0], rtol=0, atol=1e-9)
    assert pose.numVotes == int(golden["top_votes"][0])
