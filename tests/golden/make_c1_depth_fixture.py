"""Regenerate tests/golden/c1_depth_window.npz: a window of the reference's own depth frame (data/1_depth.exr,
decoded by make_c1_fixture.read_exr_float) around the bottle, for the pre-processing tests (row N4): the test rebuilds
the 720 x 1280 image with zeros outside the window, back-projects the valid pixels with the reference's intrinsics and
runs SceneCropping -> Subsampling -> OutlierProcessing -> NormalEstimation -> EdgeExtraction on it.

    python tests/golden/make_c1_depth_fixture.py      (build container only: needs /root/reference/data/1_depth.exr)

The output is DATA (a float32 depth sub-image in metres and its position)."""
import os

import numpy as np

from make_c1_fixture import BBOX, EXR, read_exr_float

HERE = os.path.dirname(os.path.abspath(__file__))
ROW0, ROW1, COL0, COL1 = 150, 620, 440, 840


def main():
    depth = read_exr_float(EXR)
    assert depth.shape == (720, 1280)
    win = depth[ROW0:ROW1, COL0:COL1].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "c1_depth_window.npz"), depth_window=win, row0=ROW0, col0=COL0,
                        shape=np.array(depth.shape), bbox=np.array(BBOX),
                        intr=np.array([614.384, 614.365, 638.121, 364.01]))
    print("window", win.shape, "valid", int((win > 0).sum()), "bbox", BBOX)


if __name__ == "__main__":
    main()
