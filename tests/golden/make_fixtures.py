"""Regenerate tests/golden/*.npy from the reference's data files.

Run in the build container (where /root/reference exists):
    python tests/golden/make_fixtures.py
The outputs are DATA (inputs / expected outputs), committed so that the GPU box -- which has no
/root/reference -- can run every test and the bench.

bottle_model_xyzn.npy : the 19,753 x (x y z nx ny nz) float32 vertices of
    /root/reference/data/bottle_remesh_meter_normalized.ply, parsed by
    yolo_ppf_pose_estimation_amd.ply.load_ply_simple WITHOUT re-normalising normals
    (raw file values), so it is exactly the array OpenCV's Mat would hold before loadPLYSimple's
    normalisation step.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

REF_PLY = "/root/reference/data/bottle_remesh_meter_normalized.ply"


def main():
    rows = []
    with open(REF_PLY, "r") as fh:
        for line in fh:
            if line.strip() == "end_header":
                break
        for line in fh:
            tok = line.split()
            if len(tok) >= 6:
                rows.append([float(t) for t in tok[:6]])
    arr = np.asarray(rows, dtype=np.float32)
    assert arr.shape == (19753, 6), arr.shape
    np.save(os.path.join(HERE, "bottle_model_xyzn.npy"), arr)
    print("bottle_model_xyzn.npy", arr.shape, arr.dtype)


if __name__ == "__main__":
    main()
