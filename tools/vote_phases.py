#!/usr/bin/env python3
"""Where k_vote's wave time goes, from a diagnostic build of the library (-DPPF_PHASE_CLOCKS, tools/build_variant.sh):

    tools/build_variant.sh phases -DPPF_PHASE_CLOCKS
    PPF_HIP_LIB=build_var/phases.so python tools/vote_phases.py [c2|c4]       # on the GPU box

Every wave sums the shader clocks (s_memtime) it spends per phase; the library adds them up over the call
(ppf_match_stats.phase_clocks).  Printed as shares of the waves' total time in the kernel."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = ["staging: clear, run table, scans, barriers", "claim + look-up + prefetch of the next item", "count-table items",
         "direct items of more than 32 records", "direct items of at most 32 records", "end of a segment: waiting for the other waves",
         "scan, reductions, result", "whole workgroup"]


def main():
    import torch
    from yolo_ppf_pose_estimation_amd import workloads as W
    from yolo_ppf_pose_estimation_amd.detector import PPF3DDetector
    from yolo_ppf_pose_estimation_amd.device import Workspace
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    bottle = W.bottle()
    if cfg == "c4":
        det = PPF3DDetector(W.C4["model_step"], W.REL_DISTANCE).trainModel(bottle)
        scene = W.c4_scene()
    else:
        det = PPF3DDetector(W.C2["model_step"], W.REL_DISTANCE).trainModel(bottle)
        scene = W.c2_scene()
    d = torch.from_numpy(scene).cuda()
    ws = Workspace(timing=True)
    st = None
    for _ in range(3):
        ws.match_device(det, d.data_ptr(), scene.shape[0], 6, W.SCENE_STEP, W.REL_DISTANCE, presampled=True)
        st = ws.results(scene.shape[0])["stats"]
    ph = st["phase_clocks"]
    tot = float(sum(ph[:7])) or 1.0
    resident = ph[7] / 100e6 / (16 * 256 * st["ms_vote_kernel"] * 1e-3)  # phase 7 is in 100 MHz ticks, summed over 16 waves per workgroup
    out = {"config": cfg, "k_vote_ms": st["ms_vote_kernel"], "phase_clocks": ph, "cu_time_with_a_workgroup_resident": resident,
           "shares": {NAMES[k]: ph[k] / tot for k in range(7)}}
    print(json.dumps(out))
    for k in range(7):
        print("%5.1f %%  %s" % (100.0 * ph[k] / tot, NAMES[k]), file=sys.stderr)
    print("k_vote %.3f ms; a workgroup was resident for %.1f %% of the CUs' time" % (st["ms_vote_kernel"], 100.0 * resident), file=sys.stderr)


if __name__ == "__main__":
    main()
