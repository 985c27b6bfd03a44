"""Shared inputs of the pre-processing tests: the reference's depth frame (window fixture) as the organised scene
cloud its driver loads, and small synthetic clouds."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def c1_frame():
    """(scene xyz (N,3) float32 of all valid pixels, depth (720,1280) float32, box (x,y,w,h), intr (fx,fy,ppx,ppy))"""
    z = np.load(os.path.join(GOLDEN, "c1_depth_window.npz"))
    depth = np.zeros(tuple(int(v) for v in z["shape"]), dtype=np.float32)
    win = z["depth_window"]
    r0, c0 = int(z["row0"]), int(z["col0"])
    depth[r0:r0 + win.shape[0], c0:c0 + win.shape[1]] = win
    fx, fy, ppx, ppy = [float(v) for v in z["intr"]]
    vv, uu = np.nonzero(depth > 0)
    zz = depth[vv, uu].astype(np.float64)
    xyz = np.stack([(uu - ppx) * zz / fx, (vv - ppy) * zz / fy, zz], axis=1).astype(np.float32)
    return xyz, depth, tuple(int(v) for v in z["bbox"]), (fx, fy, ppx, ppy)


def plane_cloud(n=1500, seed=0, normal=(0.2, -0.3, 0.93), offset=0.7, noise=0.0):
    rng = np.random.default_rng(seed)
    nrm = np.asarray(normal, dtype=np.float64)
    nrm /= np.linalg.norm(nrm)
    a = np.cross(nrm, [1.0, 0, 0]); a /= np.linalg.norm(a)
    b = np.cross(nrm, a)
    uv = rng.uniform(-0.1, 0.1, size=(n, 2))
    p = offset * nrm + uv[:, :1] * a + uv[:, 1:] * b + noise * rng.normal(size=(n, 1)) * nrm
    return p.astype(np.float32), nrm


def sphere_cloud(n=2000, seed=1, radius=0.05, center=(0.02, -0.01, 0.6)):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return (np.asarray(center) + radius * d).astype(np.float32), d
