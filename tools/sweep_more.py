#!/usr/bin/env python3
"""One-off differential campaign: the random configurations of tests/test_gpu_random_sweep.py for seeds beyond the 24 the
suite runs (python tools/sweep_more.py FIRST LAST, on a GPU box).  Prints one line per seed, exits non-zero on a mismatch."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import test_gpu_random_sweep as T  # noqa: E402


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    bottle = np.load(os.path.join(ROOT, "tests", "golden", "bottle_model_xyzn.npy"))
    bad = 0
    for seed in range(first, last):
        try:
            T.test_random_configuration.__wrapped__(bottle, seed) if hasattr(T.test_random_configuration, "__wrapped__") else T.test_random_configuration(bottle, seed)
            print("seed", seed, "ok", T._draw(seed), flush=True)
        except Exception:
            bad += 1
            print("seed", seed, "FAILED", T._draw(seed), flush=True)
            traceback.print_exc()
    print("done:", last - first, "seeds,", bad, "failures", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
