"""The pre-processing stages on the GPU (ppf_prep_* through the C-ABI) against the CPU oracle
(oracle/ppf_prep_oracle.cpp), bit for bit, on the reference's own depth frame and on synthetic clouds; then the
reference's whole sequence crop -> subsample -> outlier removal -> normals -> edges -> Matching_S2B (+ICP) through the
CloudProcessor mirror (/root/reference/src/YOLO_cropping_ppf_test.cpp:84-123)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import soak_seeds
import prep_data as D
from yolo_ppf_pose_estimation_amd import synth
from yolo_ppf_pose_estimation_amd._capi import PPFError
from yolo_ppf_pose_estimation_amd.cloud_processor import CloudProcessor, DeviceCloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frame():
    return D.c1_frame()


def test_upload_download_roundtrip(bottle):
    c = DeviceCloud.upload(bottle[:1000])
    rows, curv = c.download()
    np.testing.assert_array_equal(rows, bottle[:1000])
    assert not curv.any() and len(c) == 1000
    c3 = DeviceCloud.upload(bottle[:10, :3])
    np.testing.assert_array_equal(c3.rows()[:, :3], bottle[:10, :3])
    assert not c3.rows()[:, 3:].any()
    wide = np.zeros((10, 12), np.float32); wide[:, :6] = bottle[:10]
    np.testing.assert_array_equal(DeviceCloud.upload(wide).rows(), bottle[:10])


def test_real_frame_every_stage_bitwise(frame):
    xyz, depth, box, intr = frame
    scene = DeviceCloud.upload(xyz)
    # SceneCropping
    cropped = scene.crop(box, depth, intr)
    keep, _ = O.prep_crop(xyz, box, depth, intr)
    np.testing.assert_array_equal(cropped.xyz(), xyz[keep])
    # Subsampling
    sub = cropped.voxel_grid(0.003)
    want = O.prep_voxel(xyz[keep], 0.003)
    np.testing.assert_array_equal(sub.xyz(), want)
    # neighbour lists
    idx, d2 = sub.knn(51)
    widx, wd2 = O.prep_knn(want, 51)
    np.testing.assert_array_equal(idx, widx)
    np.testing.assert_array_equal(d2, wd2)
    # OutlierProcessing
    filt = sub.outlier_removal(50, 1.0)
    k2, _, _ = O.prep_sor(want, 50, 1.0)
    want = want[k2]
    np.testing.assert_array_equal(filt.xyz(), want)
    assert 0 < (~k2).sum() < 0.3 * k2.size
    # NormalEstimation
    wn = filt.normals(30)
    n, c = O.prep_normals(want, 30)
    rows, curv = wn.download()
    np.testing.assert_array_equal(rows[:, :3], want)
    np.testing.assert_array_equal(rows[:, 3:], n)
    np.testing.assert_array_equal(curv, c)
    # EdgeExtraction + PointCloudXYZNormalToMat
    edges = wn.edges(0.03)
    np.testing.assert_array_equal(edges.rows(), rows[c > 0.03])
    np.testing.assert_array_equal(wn.to_mat().rows(), O.prep_to_mat(want, n))
    np.testing.assert_array_equal(edges.to_mat().rows(), O.prep_to_mat(want[c > 0.03], n[c > 0.03]))


@pytest.mark.parametrize("leaf", [0.002, 0.0071, 0.05])
def test_voxel_grid_other_leaves_and_non_finite_points(frame, leaf):
    xyz = frame[0][::7].copy()
    xyz[5] = [np.nan, 0, 1]; xyz[99] = [0, np.inf, 1]
    np.testing.assert_array_equal(DeviceCloud.upload(xyz).voxel_grid(leaf).xyz(), O.prep_voxel(xyz, leaf))


def test_normals_and_outliers_on_synthetic_shapes():
    pts, nrm = D.plane_cloud(1333, seed=11, noise=0.0005)
    sp, _ = D.sphere_cloud(2100, seed=12)
    cloud = np.concatenate([pts, sp]).astype(np.float32)
    dc = DeviceCloud.upload(cloud)
    for k in (10, 30, 64):
        n, c = O.prep_normals(cloud, k)
        rows, curv = dc.normals(k).download()
        np.testing.assert_array_equal(rows[:, 3:], n)
        np.testing.assert_array_equal(curv, c)
    for mean_k, mul in ((50, 1.5), (8, 0.5), (63, 3.0)):
        keep, _, _ = O.prep_sor(cloud, mean_k, mul)
        np.testing.assert_array_equal(dc.outlier_removal(mean_k, mul).xyz(), cloud[keep])


def test_tiny_and_empty_clouds(frame):
    xyz, depth, box, intr = frame
    # a box in an empty corner of the frame: nothing survives, every later stage accepts the empty cloud
    empty = DeviceCloud.upload(xyz).crop((5, 5, 10, 10), depth, intr)
    k0, _ = O.prep_crop(xyz, (5, 5, 10, 10), depth, intr)
    assert len(empty) == k0.size == 0
    for stage in (lambda c: c.voxel_grid(0.01), lambda c: c.outlier_removal(50, 1.0), lambda c: c.normals(30),
                  lambda c: c.edges(0.03), lambda c: c.to_mat()):
        assert len(stage(empty)) == 0
    # fewer points than neighbours: SOR measures nothing and keeps all; normals of < 3 points are NaN
    few = DeviceCloud.upload(xyz[:20])
    np.testing.assert_array_equal(few.outlier_removal(50, 1.0).xyz(), xyz[:20])
    n, c = O.prep_normals(xyz[:20], 30)
    rows, curv = few.normals(30).download()
    np.testing.assert_array_equal(rows[:, 3:], n)
    np.testing.assert_array_equal(curv, c)
    rows2, curv2 = DeviceCloud.upload(xyz[:2]).normals(30).download()
    assert np.isnan(rows2[:, 3:]).all() and np.isnan(curv2).all()
    idx, _ = few.knn(30)
    widx, _ = O.prep_knn(xyz[:20], 30)
    np.testing.assert_array_equal(idx, widx)


def test_argument_errors(frame):
    xyz = frame[0][:100]
    c = DeviceCloud.upload(xyz)
    with pytest.raises(PPFError):
        c.voxel_grid(0.0)
    with pytest.raises(PPFError):
        DeviceCloud.upload(xyz * 1000).voxel_grid(1e-4)   # PCL: "leaf size is too small", index overflow
    with pytest.raises(PPFError):
        c.normals(0)
    with pytest.raises(PPFError):
        c.outlier_removal(64, 1.0)
    with pytest.raises(PPFError):
        c.crop((0, 0, 10, 10), np.zeros((0, 0), np.float32), (1, 1, 0, 0))


def test_reference_sequence_on_a_synthetic_frame(bottle):
    """render the bottle in front of a tilted wall into a depth image, then run the driver's sequence"""
    rows, cols, fx, fy, ppx, ppy = 360, 640, 460.0, 460.0, 319.5, 179.5
    T = np.eye(4); T[:3, :3] = synth.random_rotation(np.random.default_rng(8)); T[:3, 3] = [0.02, 0.0, 0.62]
    T[:3, 3] -= T[:3, :3] @ bottle[:, :3].mean(axis=0)
    T[:3, 3] += [0.0, 0.0, 0.0]
    obj = synth.apply_pose(bottle, T)[:, :3].astype(np.float64)
    vv, uu = np.mgrid[0:rows, 0:cols]
    ray = np.stack([(uu - ppx) / fx, (vv - ppy) / fy, np.ones_like(uu, dtype=np.float64)], axis=-1)
    nrm, off = np.array([0.1, -0.15, -1.0]) / np.linalg.norm([0.1, -0.15, -1.0]), -0.95
    depth = (off / (ray @ nrm)).astype(np.float32)            # wall behind the object
    pu = np.round(obj[:, 0] / obj[:, 2] * fx + ppx).astype(int)
    pv = np.round(obj[:, 1] / obj[:, 2] * fy + ppy).astype(int)
    order = np.argsort(-obj[:, 2])                              # nearest written last
    for du in (0, 1):
        for dv in (0, 1):
            depth[np.clip(pv[order] + dv, 0, rows - 1), np.clip(pu[order] + du, 0, cols - 1)] = obj[order, 2]
    box = (int(pu.min()), int(pv.min()), int(pu.max() - pu.min()), int(pv.max() - pv.min()))
    zz = depth.astype(np.float64)
    scene = np.stack([(uu - ppx) * zz / fx, (vv - ppy) * zz / fy, zz], axis=-1).reshape(-1, 3).astype(np.float32)
    K = np.array([[fx, 0, ppx], [0, fy, ppy], [0, 0, 1.0]])
    cp = CloudProcessor(scene, depth, [box], [39], [0], 0.05, 0.05)
    cp.SceneCropping(K)
    n_crop = len(cp.objects[0])
    cp.Subsampling(0.004)
    cp.OutlierProcessing(50, 1.0)
    cp.NormalEstimation(30)
    cp.EdgeExtraction(0.03)
    assert 0 < len(cp.objects[0]) < n_crop < scene.shape[0]
    obj_mat = cp.PointCloudXYZNormalToMat(cp.objects_with_normals[0])
    edge_mat = cp.PointCloudXYZNormalToMat(cp.objects_edges[0])
    assert 100 < edge_mat.shape[0] < obj_mat.shape[0]
    cp.LoadSingleModel(bottle, "bottle")
    with pytest.raises(PPFError):
        cp.Matching_S2B("bottle", obj_mat, edge_mat)             # not trained yet (:488-492)
    cp.TrainDetector(0.05, 0.05)
    # what the driver calls (src:121): edge-to-surface matching + ICP of the top 5 lands on the object ...
    pose = cp.Matching_S2B("bottle", obj_mat, edge_mat)
    assert pose is not None and pose.numVotes > 0 and 0 < pose.residual < 0.01
    from scipy.spatial import cKDTree
    model = bottle[::4]
    truth = cKDTree(synth.apply_pose(model, T)[:, :3].astype(np.float64))
    d, _ = truth.query(synth.apply_pose(model, pose.pose)[:, :3].astype(np.float64))
    assert d.mean() < 0.003
    # ... and is what the oracles give for the same inputs
    want = O.OracleDetector(0.05, 0.05).train_model(bottle).match(obj_mat, edge=edge_mat, relative_scene_sample_step=0.05,
                                                                  relative_scene_distance=0.05, cluster=True)
    P, r, _ = O.icp_refine(bottle, obj_mat, [w["pose"] for w in want["poses"][:5]])
    np.testing.assert_allclose(pose.pose, P[0], rtol=0, atol=1e-9)
    # the same with every cloud resident on the device (ppf_match_clouds + ppf_icp_refine_clouds): identical pose
    obj_dev = cp.PointCloudXYZNormalToMat(cp.objects_with_normals[0], resident=True)
    edge_dev = cp.PointCloudXYZNormalToMat(cp.objects_edges[0], resident=True)
    pose_dev = cp.Matching_S2B("bottle", obj_dev, edge_dev)
    np.testing.assert_array_equal(pose_dev.pose, pose.pose)
    assert (pose_dev.numVotes, pose_dev.residual) == (pose.numVotes, pose.residual)
    # plain surface matching votes for the wall that fills the crop (planes beat the bottle); it still returns a pose
    assert cp.Matching("bottle", obj_mat, 0.05, 0.05) is not None


@pytest.mark.parametrize("seed", soak_seeds(4, "PPF_SOAK_PREP"))
def test_stages_random_draw(frame, seed):
    """seeded draws over the stages' parameters and inputs (a random window of the reference's frame or a synthetic plane +
    sphere cloud with noise, duplicated rows and a few non-finite ones): voxel grid, neighbour lists, outlier removal, normals
    and edges equal the oracle bit for bit (PPF_SOAK_PREP=N adds N draws, tools/soak.sh)"""
    rng = np.random.default_rng(9000 + seed)
    fxyz, depth, box, intr = frame
    crng = np.random.default_rng(seed)   # SceneCropping with a drawn box around / beside the object (boxes touching the frame's border included)
    bx, by = int(box[0] + crng.integers(-150, 150)), int(box[1] + crng.integers(-150, 150))
    bw, bh = int(crng.integers(2, 400)), int(crng.integers(2, 400))
    bx, by = max(0, min(bx, depth.shape[1] - 2)), max(0, min(by, depth.shape[0] - 2))
    bw, bh = min(bw, depth.shape[1] - 1 - bx), min(bh, depth.shape[0] - 1 - by)
    keep_c, _ = O.prep_crop(fxyz, (bx, by, bw, bh), depth, intr)
    np.testing.assert_array_equal(DeviceCloud.upload(fxyz).crop((bx, by, bw, bh), depth, intr).xyz(), fxyz[keep_c])
    if rng.integers(0, 2):
        xyz = frame[0]
        lo = int(rng.integers(0, xyz.shape[0] - 30000))
        cloud = xyz[lo:lo + int(rng.integers(2000, 30000)):int(rng.integers(1, 4))].copy()
        cloud = cloud[np.isfinite(cloud).all(axis=1) & (cloud[:, 2] > 0)]
    else:
        a, _ = D.plane_cloud(int(rng.integers(200, 3000)), seed=seed, noise=float(rng.choice([0.0, 0.0005, 0.003])))
        b, _ = D.sphere_cloud(int(rng.integers(200, 4000)), seed=seed + 1)
        cloud = np.concatenate([a, b]).astype(np.float32)
        cloud = np.concatenate([cloud, cloud[rng.integers(0, cloud.shape[0], 50)]])   # rows that exist twice: distance ties
    if cloud.shape[0] < 80:
        pytest.skip("an empty window of the frame")
    leaf = float(rng.choice([0.002, 0.003, 0.005, 0.011]))
    dirty = cloud.copy()
    dirty[int(rng.integers(0, dirty.shape[0]))] = [np.nan, 0.0, 1.0]
    np.testing.assert_array_equal(DeviceCloud.upload(dirty).voxel_grid(leaf).xyz(), O.prep_voxel(dirty, leaf))
    sub = O.prep_voxel(cloud, leaf)
    dc = DeviceCloud.upload(sub)
    k = int(rng.integers(2, 65))
    if sub.shape[0] > k:
        idx, d2 = dc.knn(k)
        widx, wd2 = O.prep_knn(sub, k)
        np.testing.assert_array_equal(idx, widx)
        np.testing.assert_array_equal(d2, wd2)
    mean_k, mul = int(rng.integers(2, 64)), float(rng.choice([0.5, 1.0, 2.0]))
    keep, _, _ = O.prep_sor(sub, mean_k, mul)
    filt = dc.outlier_removal(mean_k, mul)
    np.testing.assert_array_equal(filt.xyz(), sub[keep])
    kn = int(rng.integers(3, 65))
    n, c = O.prep_normals(sub[keep], kn)
    rows, curv = filt.normals(kn).download()
    np.testing.assert_array_equal(rows[:, 3:], n)
    np.testing.assert_array_equal(curv, c)
    thr = float(rng.choice([0.01, 0.03, 0.08]))
    np.testing.assert_array_equal(filt.normals(kn).edges(thr).rows(), rows[c > thr])
