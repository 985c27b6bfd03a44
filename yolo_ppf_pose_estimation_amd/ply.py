"""Minimal PLY I/O for N x 6 float32 clouds (x y z nx ny nz).

Mirrors the helpers the reference's driver takes from OpenCV's ppf_match_3d namespace:
``loadPLYSimple(path, withNormals)`` (/root/reference/src/YOLO_cropping_ppf_test.cpp:114) and
``writePLY(cloud, path)`` (:127).  ASCII vertex lists only, which is what the reference's model
file (data/bottle_remesh_meter_normalized.ply, header lines 1-13) uses.
"""
from __future__ import annotations

import numpy as np


def load_ply_simple(path: str, with_normals: bool = True) -> np.ndarray:
    """Return an (N, 6) or (N, 3) float32 array from an ASCII PLY vertex list."""
    with open(path, "rb") as fh:
        n_vertex = None
        n_props = 0
        in_vertex = False
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f"{path}: no end_header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format" and tok[1] != "ascii":
                raise ValueError(f"{path}: only ascii PLY is supported")
            if tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n_vertex = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                n_props += 1
            elif tok[0] == "end_header":
                break
        if n_vertex is None:
            raise ValueError(f"{path}: no vertex element")
        data = np.loadtxt(fh, dtype=np.float64, max_rows=n_vertex, ndmin=2)
    need = 6 if with_normals else 3
    if data.shape[1] < need or n_props < need:
        raise ValueError(f"{path}: {data.shape[1]} columns, need {need}")
    out = np.ascontiguousarray(data[:, :need], dtype=np.float32)
    if with_normals:
        # loadPLYSimple re-normalises the normals it reads
        nrm = np.linalg.norm(out[:, 3:6].astype(np.float64), axis=1)
        ok = nrm > 1e-12
        out[ok, 3:6] = (out[ok, 3:6].astype(np.float64) / nrm[ok, None]).astype(np.float32)
    return out


def write_ply(cloud: np.ndarray, path: str) -> None:
    """Write an (N, 3|6) float cloud as ASCII PLY."""
    cloud = np.asarray(cloud)
    with_normals = cloud.shape[1] >= 6
    with open(path, "w") as fh:
        fh.write("ply\nformat ascii 1.0\n")
        fh.write(f"element vertex {cloud.shape[0]}\n")
        fh.write("property float x\nproperty float y\nproperty float z\n")
        if with_normals:
            fh.write("property float nx\nproperty float ny\nproperty float nz\n")
        fh.write("end_header\n")
        cols = 6 if with_normals else 3
        for row in cloud[:, :cols]:
            fh.write(" ".join(repr(float(np.float32(v))) for v in row) + "\n")


def transform_pc_pose(cloud: np.ndarray, pose: np.ndarray) -> np.ndarray:
    """ppf_match_3d::transformPCPose (/root/reference/src/YOLO_cropping_ppf_test.cpp:125):
    apply a 4x4 pose to points (with homogeneous divide) and rotate+renormalise normals."""
    cloud = np.asarray(cloud, dtype=np.float32)
    pose = np.asarray(pose, dtype=np.float64).reshape(4, 4)
    out = cloud.copy()
    p = cloud[:, :3].astype(np.float64)
    ph = p @ pose[:3, :3].T + pose[:3, 3]
    w = p @ pose[3, :3].T + pose[3, 3]
    ok = np.abs(w) > 1e-12
    ph[ok] /= w[ok, None]
    out[:, :3] = ph.astype(np.float32)
    if cloud.shape[1] >= 6:
        n = cloud[:, 3:6].astype(np.float64) @ pose[:3, :3].T
        nrm = np.linalg.norm(n, axis=1)
        ok = nrm > 1e-12
        n[ok] /= nrm[ok, None]
        out[:, 3:6] = n.astype(np.float32)
    return out
