"""The -O3 -march=native build of the oracle (bench.py's cpu_baseline) gives the same integers and the same poses as
the portable -O2 build the parity tests use (both compile with -ffp-contract=off -fno-fast-math)."""
import subprocess
import sys
import os
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import sys, numpy as np
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    import oracle_lib as O
    from yolo_ppf_pose_estimation_amd import synth, workloads as W
    native = %s
    if native:
        assert O.use_native_build(), "native build failed"
    bottle = W.bottle()
    scene, _ = synth.make_scene(bottle, n_points=3000, seed=5)
    r = O.OracleDetector(0.07, 0.05).train_model(bottle).match(scene, relative_scene_sample_step=0.1, presampled=True)
    np.savez(sys.argv[1], triples=r["triples"], votes=r["votes_per_ref"], poses=np.stack([p["pose"] for p in r["poses"]]))
""")


def test_native_build_is_bit_identical(tmp_path):
    outs = []
    for native in (False, True):
        out = tmp_path / f"r{int(native)}.npz"
        script = tmp_path / f"s{int(native)}.py"
        script.write_text(SCRIPT % (ROOT, os.path.join(ROOT, "tests"), native))
        r = subprocess.run([sys.executable, str(script), str(out)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    for k in ("triples", "votes", "poses"):
        np.testing.assert_array_equal(outs[0][k], outs[1][k])
