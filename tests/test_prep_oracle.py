"""CPU checks of the pre-processing oracle (oracle/ppf_prep_oracle.cpp; restates the PCL stages called from
/root/reference/include/CloudProcessing.h:263-427).  Parity unpinned (PCL is not available and the reference holds no
outputs of these stages): property tests of the restated algorithms, on synthetic clouds and on the reference's own
depth frame."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

import oracle_lib as O
import prep_data as D


def test_crop_keeps_the_box_frustum_of_the_real_frame():
    xyz, depth, box, intr = D.c1_frame()
    keep, planes = O.prep_crop(xyz, box, depth, intr)
    assert 10000 < keep.size < xyz.shape[0]
    fx, fy, ppx, ppy = intr
    p = xyz[keep].astype(np.float64)
    u, v = p[:, 0] / p[:, 2] * fx + ppx, p[:, 1] / p[:, 2] * fy + ppy
    x, y, w, h = box
    # inside the box grown by 30 px (the base quad sits 0.15 m behind the corners' rays, so allow its lateral slack)
    assert u.min() > x - 31 and u.max() < x + w + 31 and v.min() > y - 31 and v.max() < y + h + 31
    assert p[:, 2].max() <= planes[12]
    # every dropped point violates one of the five planes
    drop = np.setdiff1d(np.arange(xyz.shape[0]), keep)
    q = xyz[drop].astype(np.float64)
    side = np.stack([q @ planes[3 * f:3 * f + 3] for f in range(4)], axis=1)
    assert np.all((side.min(axis=1) < 0) | (q[:, 2] > planes[12]))


def test_voxel_grid_cells_and_centroids():
    rng = np.random.default_rng(2)
    pts = rng.uniform(-0.1, 0.1, size=(5000, 3)).astype(np.float32)
    pts[7] = [np.nan, 0, 0]
    leaf = 0.02
    v = O.prep_voxel(pts, leaf)
    cell = np.floor(v / np.float32(leaf)).astype(np.int64)
    assert np.unique(cell, axis=0).shape[0] == v.shape[0]          # one output per occupied cell
    fin = pts[np.isfinite(pts).all(axis=1)]
    want = np.unique(np.floor(fin * (np.float32(1) / np.float32(leaf))).astype(np.int64), axis=0).shape[0]
    assert v.shape[0] == want
    assert abs(v.mean(axis=0) - 0).max() < 0.01
    # ascending cell index, x fastest
    lo = np.floor(fin.min(axis=0) * (np.float32(1) / np.float32(leaf))).astype(np.int64)
    hi = np.floor(fin.max(axis=0) * (np.float32(1) / np.float32(leaf))).astype(np.int64)
    div = hi - lo + 1
    key = (cell[:, 0] - lo[0]) + (cell[:, 1] - lo[1]) * div[0] + (cell[:, 2] - lo[2]) * div[0] * div[1]
    assert np.all(np.diff(key) > 0)
    with pytest.raises(ValueError):
        O.prep_voxel(pts * 1000, 1e-4)


def test_knn_is_exact():
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(700, 3)).astype(np.float32)
    idx, d2 = O.prep_knn(pts, 12)
    _, want = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=12)
    assert (idx == want).mean() > 0.999 and np.all(idx[:, 0] == np.arange(700))
    assert np.all(np.diff(d2, axis=1) >= 0)
    idx2, _ = O.prep_knn(pts[:5], 8)  # fewer points than neighbours asked for
    assert np.all(idx2[:, 5:] == -1) and np.all(np.sort(idx2[:, :5], axis=1) == np.arange(5))


def test_outlier_removal_drops_planted_outliers():
    pts, _ = D.plane_cloud(1500, seed=4)
    rng = np.random.default_rng(5)
    far = (pts[:20] + rng.uniform(0.05, 0.1, size=(20, 3)).astype(np.float32) * [0, 0, 1])
    cloud = np.concatenate([pts, far]).astype(np.float32)
    keep, dist, thr = O.prep_sor(cloud, 50, 1.0)
    assert not keep[1500:].any() and keep[:1500].mean() > 0.9
    assert np.all(keep == ~(dist.astype(np.float64) > thr))
    keep_small, dist_small, _ = O.prep_sor(cloud[:30], 50, 1.0)  # fewer points than meanK + 1: nothing is measured
    assert keep_small.all() and not dist_small.any()


def test_normals_of_plane_and_sphere():
    pts, nrm = D.plane_cloud(1200, seed=6)
    n, c = O.prep_normals(pts, 30)
    assert np.abs(np.abs(n @ nrm) - 1).max() < 1e-6 and c.max() < 1e-6
    assert np.all((n * pts).sum(axis=1) < 0)                        # towards the camera at the origin
    sp, radial = D.sphere_cloud(3000, seed=7)
    n, c = O.prep_normals(sp, 30)
    assert np.median(np.abs((n * radial).sum(axis=1))) > 0.99
    assert np.abs(np.linalg.norm(n, axis=1) - 1).max() < 1e-6
    assert np.all((n * sp).sum(axis=1) < 0)
    n2, c2 = O.prep_normals(sp[:2], 30)
    assert np.isnan(n2).all() and np.isnan(c2).all()


def test_real_frame_pipeline_finds_edges_on_the_bottle():
    xyz, depth, box, intr = D.c1_frame()
    keep, _ = O.prep_crop(xyz, box, depth, intr)
    v = O.prep_voxel(xyz[keep], 0.003)
    k2, _, _ = O.prep_sor(v, 50, 1.0)
    v = v[k2]
    n, c = O.prep_normals(v, 30)
    rows = O.prep_to_mat(v, n)
    assert 3000 < rows.shape[0] < 40000
    assert np.abs(np.linalg.norm(rows[:, 3:], axis=1) - 1).max() < 1e-6
    edge = rows[c > 0.03]
    assert 0.01 < edge.shape[0] / rows.shape[0] < 0.5
