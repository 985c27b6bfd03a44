"""Regenerate tests/golden/c1_crop_xyzn.npy / c1_edge_xyzn.npy: BASELINE config C1's scene from the reference's
own depth frame (the scene cloud data/1_cloud.ply is missing from the reference, .MISSING_LARGE_BLOBS:1).

    python tests/golden/make_c1_fixture.py      (build container only: needs /root/reference/data/1_depth.exr)

Steps, all plain numpy/scipy -- this is fixture preparation, not the engine:
  1. decode 1_depth.exr: OpenEXR scanline file, one FLOAT channel, ZIP blocks of 16 lines
     (zlib -> predictor undo -> even/odd byte de-interleave)
  2. back-project valid pixels with the reference's intrinsics fx=614.384 fy=614.365 cx=638.121 cy=364.01
     (src/YOLO_cropping_ppf_test.cpp:35-37, include/Camera.h:56-58)
  3. YOLO weights are not shipped, so the crop uses a FIXED bounding box around the bottle (centre of the image,
     z ~ 0.6 m), expanded by 30 px and padded 0.15 m in depth like CloudProcessing.h:279-299
  4. voxel-grid subsample (leaf 3 mm), statistical outlier removal (meanK 50), PCA normals over k=30 neighbours
     flipped towards the camera (src:96-101), curvature > 0.03 -> edge cloud (src:103)
The outputs are DATA (N x 6 float32 clouds).
"""
import os
import struct
import zlib

import numpy as np
from scipy.spatial import cKDTree

HERE = os.path.dirname(os.path.abspath(__file__))
EXR = "/root/reference/data/1_depth.exr"
FX, FY, CX, CY = 614.384, 614.365, 638.121, 364.01
BBOX = (560, 250, 160, 260)  # x, y, w, h of the bottle in the 1280x720 frame (fixed: YOLO is not available)


def read_exr_float(path):
    data = open(path, "rb").read()
    assert struct.unpack_from("<I", data, 0)[0] == 20000630
    pos = 8
    hdr = {}
    while True:
        end = data.index(b"\0", pos)
        name = data[pos:end].decode()
        pos = end + 1
        if not name:
            break
        end = data.index(b"\0", pos)
        typ = data[pos:end].decode()
        pos = end + 1
        size = struct.unpack_from("<i", data, pos)[0]
        pos += 4
        hdr[name] = (typ, data[pos:pos + size])
        pos += size
    xmin, ymin, xmax, ymax = struct.unpack("<4i", hdr["dataWindow"][1])
    w, h = xmax - xmin + 1, ymax - ymin + 1
    comp = hdr["compression"][1][0]
    assert comp == 3, f"expected ZIP (16-line) compression, got {comp}"
    ch = hdr["channels"][1]
    cname = ch[:ch.index(b"\0")].decode()
    ptype = struct.unpack_from("<i", ch, len(cname) + 1)[0]
    assert ptype == 2, "expected one FLOAT channel"
    nblocks = (h + 15) // 16
    offsets = struct.unpack_from(f"<{nblocks}Q", data, pos)
    img = np.zeros((h, w), dtype=np.float32)
    for off in offsets:
        y, size = struct.unpack_from("<ii", data, off)
        raw = data[off + 8: off + 8 + size]
        lines = min(16, ymax - y + 1)
        expect = lines * w * 4
        if size < expect:
            buf = np.frombuffer(zlib.decompress(raw), dtype=np.uint8).astype(np.int32)
            buf = np.cumsum(buf - np.concatenate([[0], np.full(len(buf) - 1, 128)])) & 0xFF  # predictor
            buf = buf.astype(np.uint8)
            half = (len(buf) + 1) // 2
            out = np.empty(len(buf), dtype=np.uint8)
            out[0::2] = buf[:half]
            out[1::2] = buf[half:]
            raw = out.tobytes()
        img[y - ymin: y - ymin + lines] = np.frombuffer(raw, dtype="<f4").reshape(lines, w)
    return img


def main():
    depth = read_exr_float(EXR)
    assert depth.shape == (720, 1280)
    valid = depth > 0
    print("valid px", int(valid.sum()), "median depth", float(np.median(depth[valid])))
    x0, y0, w, h = BBOX
    x0, y0, x1, y1 = x0 - 30, y0 - 30, x0 + w + 30, y0 + h + 30
    sub = depth[y0:y1, x0:x1]
    vv, uu = np.nonzero(sub > 0)
    z = sub[vv, uu].astype(np.float64)
    corners = [depth[y0 + 30, x0 + 30], depth[y0 + 30, x1 - 30], depth[y1 - 30, x0 + 30], depth[y1 - 30, x1 - 30]]
    zc = float(np.median(z))
    keep = z < zc + 0.15   # depth pad of the crop frustum
    u = (uu[keep] + x0).astype(np.float64)
    v = (vv[keep] + y0).astype(np.float64)
    z = z[keep]
    pts = np.stack([(u - CX) * z / FX, (v - CY) * z / FY, z], axis=1)
    print("crop points", pts.shape[0], "corner depths", corners)
    # voxel grid (leaf 3 mm): centroid per voxel
    leaf = 0.003
    key = np.floor((pts - pts.min(0)) / leaf).astype(np.int64)
    _, inv = np.unique(key, axis=0, return_inverse=True)
    cnt = np.bincount(inv)
    vox = np.stack([np.bincount(inv, weights=pts[:, k]) / cnt for k in range(3)], axis=1)
    # statistical outlier removal, meanK = 50, threshold 1.0 std
    tree = cKDTree(vox)
    d, _ = tree.query(vox, k=51)
    md = d[:, 1:].mean(axis=1)
    vox = vox[md < md.mean() + 1.0 * md.std()]
    # PCA normals, k = 30, flipped towards the camera (origin); curvature = l0 / (l0 + l1 + l2)
    tree = cKDTree(vox)
    _, idx = tree.query(vox, k=30)
    nb = vox[idx] - vox[idx].mean(axis=1, keepdims=True)
    cov = np.einsum("nki,nkj->nij", nb, nb) / 30.0
    wv, vecs = np.linalg.eigh(cov)
    nrm = vecs[:, :, 0]
    flip = np.sum(nrm * vox, axis=1) > 0
    nrm[flip] *= -1
    curv = wv[:, 0] / np.maximum(wv.sum(axis=1), 1e-30)
    cloud = np.concatenate([vox, nrm], axis=1).astype(np.float32)
    edge = cloud[curv > 0.03]
    np.save(os.path.join(HERE, "c1_crop_xyzn.npy"), cloud)
    np.save(os.path.join(HERE, "c1_edge_xyzn.npy"), edge)
    print("c1_crop_xyzn", cloud.shape, "c1_edge_xyzn", edge.shape, "bbox", cloud[:, :3].min(0), cloud[:, :3].max(0))


if __name__ == "__main__":
    main()
