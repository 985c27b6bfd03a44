#!/usr/bin/env python3
"""bench.py — PPF pair-matches/s of the MI355X voting engine on BASELINE.json's workloads.

    python bench.py --gpus N --steps K --warmup W [--config c2|c4|c5] [--shard refs]

`--gpus N` with N > 1 starts N ranks BY ITSELF when it was not started under torch.distributed.run
(no RANK/WORLD_SIZE in the environment): the parent process never imports torch or touches the GPU, it
only spawns N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, watches ALL of them (a rank that
dies takes the others down with it instead of leaving them in a collective) and relays rank 0's line.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` the ranks already exist and each
process is one of them.  WORLD_SIZE != --gpus is an error.

A step = one pass of the hot path over one synthetic YOLO crop per rank: pair features + hash + table
lookup (k_pairs) + grouping (k_group) + Hough voting (k_vote) + argmax + pose assembly (k_finalize) + pose
clustering, with the crop and the model table already resident in HBM.

  c2  (default) BASELINE.json configs[1]: bottle sampled to 2,000 points vs one 50,000-point crop (seed 12345),
      presampled so all 50k points vote, reference stride 20 -> 2,500 reference points.  At N > 1 this is
      configs[2] (C3): every rank matches its own crop (seeds 1000+rank, weak scaling) and the only collective is
      one all_gather (RCCL) of each rank's top-5 clustered poses per step, taken from device memory.
  c4  configs[3]: ~10k-point model vs one 200,000-point scene.  `--shard refs` at N > 1 strides the scene's
      reference points over the ranks (strong scaling), all_gathers the raw per-reference poses on the
      device and clusters them on rank 0.
  c5  configs[4]: four resident model tables x 8 crops per rank through the batched entry.

The ONE line printed by rank 0 carries
  value          whole-job pair-matches (accumulator increments the reference would make, exact integer) per second
  roofline       the voting kernel against HBM: measured fabric traffic per step (committed PMC pass of THIS source
                 tree: the pass records a hash of the kernel sources and its own kernel time; another tree or a kernel time
                 more than 20 % away makes traffic_stale true and frac null; the ratio is reported as time_vs_collection) / its device time in this run (HIP events on
                 the launch stream); the SURVEY section 8d figure stays beside it as algorithmic_bytes_per_launch
  issue_roofline / lds_roofline   what actually bounds k_vote: LDS wave-instructions and atomic lane-operations per second
  host_entry     (N = 1, c2) the call the reference really makes: ppf_match from HOST memory, upload and read-back
                 included (SURVEY section 8d's "poses/s per crop"), on a warm context
  other_configs  (N = 1, c2) C4 and C5 measured in the same run, 2 steps each, with the same roofline fields, and c1_pipeline:
                 the call the reference's main() makes on its own depth frame (crop -> voxel -> SOR -> normals -> edges ->
                 match_S2B, not presampled -> ICP of the top 5): frame_to_pose_ms with prep_ms / match_ms / icp_ms, the CPU
                 oracles' time for the same chain beside it
  pipelined      (N = 1, c2) throughput with two independent crops in flight (extra information, never `value`)
  cpu_baseline   the CPU oracle (kind "port", -O3 -march=native build made on this box) on a bounded sample of the
                 same workload: min and median of 5 repetitions on all usable cores plus a 1-thread figure; N = 1 only
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# LDS atomic ceilings, lane-operations per second chip-wide:
LDS_ATOMIC_PEAK_UBENCH = 8.97e12  # profiles/r01_ubench_valu_lds.txt: ds_add_u32 14.6 lanes/clk/CU x 256 CUs x 2.4 GHz
LDS_ATOMIC_PEAK_GUIDE = 9.83e12   # MI355X_MICROARCH.md LDS section: 16 lanes/clk/CU x 256 x 2.4 GHz
LDS_INSTR_PEAK = 256 * 2.4e9 / 4.38  # ds_add_u32: one 64-lane wave-instruction per 4.38 cycles per CU (r01 / r02 ubench)
STALE_KERNEL_TIME = 0.20  # boxes of the pool run the same binary up to 11 % apart (k_vote 3.81 and 4.21 ms minutes apart, round 3): the source
# hash is what ties a counter pass to the kernel; the time check only catches a pass that cannot be about this kernel at all


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", default="c2", choices=["c2", "c4", "c5"])
    ap.add_argument("--shard", default="crops", choices=["crops", "refs"],
                    help="c4 at N > 1: 'refs' strides the reference points of the one scene over the ranks")
    ap.add_argument("--cpu-seconds", type=float, default=3.0, help="CPU baseline: seconds per repetition")
    ap.add_argument("--cells", choices=["auto", "32", "16", "limit"], default="auto",
                    help="accumulator cells of the vote kernel: auto = 16-bit, 32-bit for what overflows (the library's default); "
                         "32 = 32-bit from the first call (profiling a workload that always overflows, e.g. c4)")
    ap.add_argument("--pipeline-depth", type=int, default=1,
                    help="crops in flight: 1 (default) = strictly one after another, so the HIP-event kernel times "
                         "behind `roofline` are the kernel's own; 2-3 overlap independent crops on separate streams")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="c2 at N = 1: leave out the host-entry measurement and the C4 / C5 legs (profiling runs)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="ranks rendezvous (gloo, CPU), report RANK/WORLD_SIZE and stop before any device use")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# launcher: runs in a process that has NOT touched the GPU (no torch import up to here)
# ---------------------------------------------------------------------------------------------------------------
def launch_ranks(args, argv, poll_s=0.2):
    """Spawn one child per rank and watch ALL of them: the first child that exits non-zero ends the others (they would
    otherwise sit in init_process_group or a barrier until the collective's timeout) and its code is returned.  Rank 0's
    stdout is relayed when everything has ended; the other ranks' stderr goes to this process's stderr."""
    import socket
    import tempfile

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = set(range(args.gpus))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = abs(code) or 1
                sys.stderr.write(f"bench.py: rank {r} exited with {code}; stopping the other ranks\n")
                for o in sorted(live):
                    procs[o].terminate()
        if live:
            time.sleep(poll_s)
            if rc:  # give terminated ranks a moment, then make sure
                deadline = time.time() + 10.0
                while live and time.time() < deadline:
                    live = {o for o in live if procs[o].poll() is None}
                    time.sleep(poll_s)
                for o in live:
                    procs[o].kill()
                for o in live:
                    procs[o].wait()
                live = set()
    out0.seek(0)
    for ln in out0.read().decode().splitlines():  # the contract is ONE JSON line on stdout; library chatter goes to stderr
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    return rc


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, further limited by the cgroup CPU quota (a GPU box shows
    all 256 hardware threads but grants a share of them; more threads than that only adds context switches and would
    misstate `cores`)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def algorithmic_bytes(n_ref, n_model, num_angles, n_pairs, n_votes):
    """per reference point 24 B (its xyzn) + accumulator clear and scan 2 x 4 x N_m x A + 12 B
    result; per scene pair 24 B (xyzn) + 8 B (bucket header); per vote 8 B model entry + 8 B accumulator RMW."""
    return n_ref * (24 + 2 * 4 * n_model * num_angles + 12) + n_pairs * 32 + n_votes * 16


def kernel_source_hash():
    """sha256 over the sources libppf_hip.so is built from (the kernels and their launch code): what a PMC summary must have
    been collected on to describe the kernels of this run (tools/pmc_summary.py stores the same hash)."""
    import glob
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "yolo_ppf_pose_estimation_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(ROOT, "yolo_ppf_pose_estimation_amd", "csrc", "*.hip")) +
                   [os.path.join(ROOT, "include", "ppf_hip.h"), os.path.join(ROOT, "include", "ppf_detmath.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    # ... and the flags they are compiled with, and whether another library than the in-tree build is loaded (PPF_HIP_LIB: a
    # variant or diagnostic build carries its own -D switches): a counter pass is only this run's if all of that agrees
    import __graft_entry__ as G
    h.update(" ".join(G.HIP_FLAGS).encode())
    h.update(("lib:" + os.path.basename(os.environ["PPF_HIP_LIB"])).encode() if os.environ.get("PPF_HIP_LIB") else b"lib:in-tree")
    return h.hexdigest()


def committed_counters(workload, n_votes_per_step):
    """The newest committed PMC summary (profiles/*_pmc_<workload>.json, tools/pmc_summary.py) of this workload (same
    vote count per step), or None.  bench.py cannot collect PMC counters on itself."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_{workload}.json")), reverse=True):
        try:
            t = json.load(open(f))
            if t.get("n_votes_per_step", t.get("n_votes_per_launch")) == n_votes_per_step:
                t["file"] = os.path.basename(f)
                return t
        except Exception:
            pass
    return None


def counter_rooflines(workload, n_votes_per_step, k_vote_ms, abytes, atomics_per_step, src_hash):
    """roofline / issue_roofline / lds_roofline of k_vote from the committed counter pass of this workload, refused when the
    pass does not describe the kernel that just ran (other source tree, or a kernel time more than 20 % away: boxes of the pool differ by up to 11 % on one binary)."""
    pmc = committed_counters(workload, n_votes_per_step)
    k_vote_s = k_vote_ms * 1e-3
    stale, why = False, None
    if pmc is None:
        stale, why = True, "no committed counter pass for this workload and vote count"
    else:
        at = pmc.get("k_vote_ms_per_step_at_collection")
        if pmc.get("source_hash") != src_hash:
            stale, why = True, "the counter pass was collected on another source tree"
        elif not at or abs(at - k_vote_ms) > STALE_KERNEL_TIME * k_vote_ms:
            stale, why = True, f"k_vote took {at} ms when the counters were collected, {k_vote_ms:.3f} ms now"
    traffic = pmc.get("hbm_bytes_per_step_k_vote") if pmc and not stale else None
    achieved = traffic / k_vote_s / 1e9 if traffic else None
    roofline = {
        "bound": "hbm", "kernel": "k_vote",
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS if achieved else None,
        "traffic": traffic, "traffic_source": pmc["file"] if pmc else None,
        "traffic_stale": stale, "traffic_stale_reason": why,
        "time_vs_collection": (k_vote_ms / pmc["k_vote_ms_per_step_at_collection"]) if pmc and pmc.get("k_vote_ms_per_step_at_collection") else None,
        "avg_kernel_ms": k_vote_ms,
        "algorithmic_bytes_per_launch": abytes,
        "note": "achieved = measured fabric bytes of k_vote per step (2 x FETCH_SIZE + WRITE_SIZE of the committed PMC pass, "
                "tied to this source tree by source_hash and to this run by the kernel time) / its HIP-event time of this "
                "run.  The section 8d figure (16 B per vote) is kept as algorithmic_bytes_per_launch only: the accumulator "
                "lives in LDS and a model entry is read once per run of hits.  What bounds the kernel is the LDS "
                "instruction rate, see issue_roofline / lds_roofline",
    }
    issue = pmc.get("k_vote_issue") if pmc and not stale else None
    issue_roofline = None if not issue else {
        "kernel": "k_vote", "source": pmc["file"],
        "valu_busy_frac_of_simd_time": issue["valu_busy_frac_of_simd_time"],
        "any_inst_busy_frac_of_simd_time": issue["any_inst_busy_frac_of_simd_time"],
        "lds_busy_frac_of_simd_time": issue["lds_busy_frac_of_simd_time"],
        "lds_wave_instr_per_s": issue["lds_wave_instructions_sampled"] / k_vote_s,
        "lds_wave_instr_peak_per_s": LDS_INSTR_PEAK,
        "frac_of_lds_instr_peak": issue["lds_wave_instructions_sampled"] / k_vote_s / LDS_INSTR_PEAK,
        "lds_array_cycles_per_instr": issue.get("lds_array_cycles_per_instr"),
        "lds_bank_conflict_frac_of_lds_cycles": issue["lds_bank_conflict_frac_of_lds_cycles"],
        "valu_wave_instr_per_s": issue["valu_wave_instructions_sampled"] / k_vote_s,
        "valu_wave_instr_peak_per_s": 1024 * 2.4e9 / 4.45,
        "frac_of_valu_issue_peak": issue["valu_wave_instructions_sampled"] / k_vote_s / (1024 * 2.4e9 / 4.45),
        "note": "SQ_INSTS_LDS of one step / k_vote time against one 64-lane LDS wave-instruction per 4.38 cycles per CU "
                "(the cost of ds_add_u32 / ds_write_b32 whatever the lanes do, as long as the LDS array needs no more: two "
                "32-lane halves, 32 banks, one array cycle per distinct address on the fullest bank of a half; "
                "profiles/r03_ubench_lds_counters.md).  lds_array_cycles_per_instr = SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS.  "
                "VALU: SQ_INSTS_VALU against 1024 SIMDs x 2.4 GHz / 4.45 cycles; busy fractions = SQ_ACTIVE_INST_* / "
                "(SQ_WAVE_CYCLES / 4)",
    }
    atom = atomics_per_step or None
    lds_roofline = {
        "kernel": "k_vote", "unit": "LDS atomic lane-ops/s",
        "lds_atomics_per_launch": atom,
        "achieved": atom / k_vote_s if atom else None,
        "peak_ubench": LDS_ATOMIC_PEAK_UBENCH, "peak_guide": LDS_ATOMIC_PEAK_GUIDE,
        "frac_ubench": atom / k_vote_s / LDS_ATOMIC_PEAK_UBENCH if atom else None,
        "frac_guide": atom / k_vote_s / LDS_ATOMIC_PEAK_GUIDE if atom else None,
        "votes_per_lds_atomic": n_votes_per_step / atom if atom else None,
        "votes_per_s_kernel": n_votes_per_step / k_vote_s,
        "source": "profiles/r01_ubench_valu_lds.txt; MI355X_MICROARCH.md (ds_write-class op: 16 lanes/clk/CU)",
    }
    return roofline, issue_roofline, lds_roofline


def cpu_baseline(model_step, bottle, scene, n_ref_total, seconds_per_rep, reps=5):
    """The oracle (CPU restatement, -O3 -march=native build made on this machine) on an evenly spaced sample of the
    step's reference points: `reps` runs on every usable core (min and median reported: the box's cores are shared and
    single runs swing), and one 1-thread run."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from yolo_ppf_pose_estimation_amd import workloads as W

    native = oracle_lib.use_native_build()
    threads = min(oracle_lib.max_threads(), usable_cpus())
    ora = oracle_lib.OracleDetector(model_step, W.REL_DISTANCE).train_model(bottle)
    step = int(1.0 / W.SCENE_STEP)

    def timed(refs, nthreads):
        t0 = time.perf_counter()
        r = ora.match(scene, relative_scene_sample_step=W.SCENE_STEP, presampled=True, ref_list=refs, threads=nthreads,
                      cluster=False)
        return time.perf_counter() - t0, int(r["votes_per_ref"].sum())

    def spaced(n):
        return [int(k * n_ref_total / n) * step for k in range(n)]

    # calibrate on 2 reference points per thread, then size one repetition for ~seconds_per_rep
    dt, _ = timed(spaced(2 * threads), threads)
    n_sample = int(min(n_ref_total, max(2 * threads, seconds_per_rep / max(dt, 1e-3) * 2 * threads)))
    n_sample = max(threads, (n_sample // threads) * threads)
    refs = spaced(n_sample)
    runs = [timed(refs, threads) for _ in range(reps)]
    votes = runs[0][1]
    secs = [d for d, _ in runs]
    med, best = float(np.median(secs)), float(min(secs))
    n1 = max(2, n_sample // threads)
    d1, v1 = timed(spaced(n1), 1)
    return {
        "value": votes / med,
        "value_best_repetition": votes / best,
        "unit": "pair-matches/s",
        "cores": threads,
        "host_cpus_visible": os.cpu_count(),
        "kind": "port",
        "build": "-O3 -march=native (oracle/Makefile target native, compiled on this box)" if native
                 else "-O2 (portable build; the native build failed on this box)",
        "repetitions": reps,
        "seconds_per_repetition": [round(d, 3) for d in secs],
        "seconds_median": med, "seconds_min": best,
        "sample": f"{n_sample} of {n_ref_total} reference points (evenly spaced) of the same crop, all "
                  f"{scene.shape[0]} paired points each, {votes} pair-matches; value = median of {reps} runs ({med:.2f} s, "
                  f"{n_sample / med:.2f} poses/s), value_best_repetition = the fastest ({best:.2f} s); the box's cores are shared",
        "poses_per_s": n_sample / med,
        "one_thread": {"value": v1 / d1, "unit": "pair-matches/s", "cores": 1,
                       "sample": f"{n1} reference points, {v1} pair-matches in {d1:.2f} s"},
    }


def host_entry(det, scene, n_ref, resident_ms, reps=12, warm=3):
    """The call the reference makes (detector.match, /root/reference/include/CloudProcessing.h:441-446: its own "PPF Elapsed
    Time" bracket): ppf_match on the crop in HOST memory -- staging, upload, the whole path, read-back of every clustered
    pose -- timed around the C call alone (pre-built ctypes arguments; the Python wrapper's object construction is not the
    library's).  Warm context: the second and later calls of a detector, which is every call but the first of a session."""
    import ctypes as C
    import numpy as np
    from yolo_ppf_pose_estimation_amd import workloads as W
    from yolo_ppf_pose_estimation_amd._capi import Pose, check, lib

    sc = np.ascontiguousarray(scene, dtype=np.float32)
    mp = det._params(W.SCENE_STEP, W.REL_DISTANCE, True)
    cap = n_ref + 8
    out = (Pose * cap)()
    n = C.c_int(0)
    f = lib().ppf_match
    argv = (det._model.ptr, sc.ctypes.data, sc.shape[0], sc.shape[1], 3, None, 0, 6, 3, C.byref(mp), out, cap, C.byref(n))
    t0 = time.perf_counter()
    check(f(*argv))
    first = time.perf_counter() - t0
    for _ in range(warm):
        check(f(*argv))
    secs = []
    for _ in range(reps):
        t0 = time.perf_counter()
        check(f(*argv))
        secs.append(time.perf_counter() - t0)
    med = float(np.median(secs))
    return {
        "entry": "ppf_match (what PPF3DDetector::match / match_S2B bind to): host rows in, clustered poses out",
        "ms_per_match": med * 1e3, "ms_min": float(min(secs)) * 1e3, "ms_first_call": first * 1e3,
        "poses_per_s": n_ref / med, "clustered_poses_returned": n.value,
        "bytes_in": int(sc.nbytes), "bytes_out": int(n.value) * C.sizeof(Pose),
        "resident_step_ms": resident_ms, "over_resident_step": med * 1e3 / resident_ms - 1.0 if resident_ms else None,
        "sample": f"median of {reps} calls after {1 + warm} warm-up calls on the same crop; first call of the detector "
                  f"(cold context: counting pass + scratch allocation) {first * 1e3:.2f} ms",
    }


def c1_pipeline(bottle, reps=12, warm=3, with_cpu=True):
    """The call the reference's main() really makes (src/YOLO_cropping_ppf_test.cpp:91-122), on its own depth frame
    (tests/golden/c1_depth_window.npz) with its parameters: upload -> SceneCropping -> Subsampling -> OutlierProcessing ->
    NormalEstimation -> EdgeExtraction -> PointCloudXYZNormalToMat x2 -> Matching_S2B = match_S2B(0.05, 0.05), not presampled,
    + ICP(100, 0.005, 2.5, 8) on the top 5 (CloudProcessing.h:494-499 and :518-528 are the reference's own two timing
    brackets).  Wall clock around the whole chain, clouds resident on the device between the stages; the CPU oracles' time for
    the same chain beside it (one run; the table is trained once, outside both)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import prep_data as D
    from yolo_ppf_pose_estimation_amd.cloud_processor import CloudProcessor
    xyz, depth, box, intr = D.c1_frame()
    K = np.array([[intr[0], 0, intr[2]], [0, intr[1], intr[3]], [0, 0, 1.0]])
    leaf, sor = 0.003, 1.0
    t0 = time.perf_counter()
    trainer = CloudProcessor(None, None, [], [], [], 0.025, 0.05)
    trainer.LoadSingleModel(bottle, "bottle")
    trainer.TrainDetector(0.025, 0.05)  # the reference's parameters (CloudProcessing.h:64-65)
    train_ms = (time.perf_counter() - t0) * 1e3
    frames, parts, last = [], [], None
    for it in range(warm + reps):
        t0 = time.perf_counter()
        cp = CloudProcessor(xyz, depth, [box], [39], [0], 0.025, 0.05)
        cp.models, cp.detectors, cp.if_trained = trainer.models, trainer.detectors, trainer.if_trained
        cp.label_to_id, cp.id_to_label, cp._model_clouds = trainer.label_to_id, trainer.id_to_label, trainer._model_clouds
        cp.SceneCropping(K)
        cp.Subsampling(leaf)
        cp.OutlierProcessing(50, sor)
        cp.NormalEstimation(30)
        cp.EdgeExtraction(0.03)
        obj = cp.PointCloudXYZNormalToMat(cp.objects_with_normals[0], resident=True)
        edge = cp.PointCloudXYZNormalToMat(cp.objects_edges[0], resident=True)
        t1 = time.perf_counter()
        pose = cp.Matching_S2B("bottle", obj, edge)
        t2 = time.perf_counter()
        if it >= warm:
            frames.append((t2 - t0) * 1e3)
            parts.append(((t1 - t0) * 1e3, cp.timings["match"] * 1e3, cp.timings["icp"] * 1e3))
        last = (pose, len(obj), len(edge))
    pose, n_obj, n_edge = last
    med = lambda v: float(np.median(np.asarray(v)))
    out = {"workload": "C1: the reference's depth frame (166,718 valid pixels), its bbox, model trained at 0.025 / 0.05; crop -> voxel 3 mm -> "
                       "SOR(50, 1.0) -> normals(30) -> edges(0.03) -> match_S2B(0.05, 0.05), not presampled -> ICP(100, 0.005, 2.5, 8) on the top 5",
           "frame_points": int(xyz.shape[0]), "object_points": n_obj, "edge_points": n_edge,
           "model_sampled_points": trainer.detectors[0].info()["n_ref"],
           "frame_to_pose_ms": med(frames), "frame_to_pose_ms_min": float(min(frames)),
           "prep_ms": med([p[0] for p in parts]), "match_ms": med([p[1] for p in parts]), "icp_ms": med([p[2] for p in parts]),
           "train_ms_once": train_ms, "votes": int(pose.numVotes), "residual": float(pose.residual),
           "sample": f"median of {reps} frames after {warm} warm-up frames; match_ms / icp_ms are the two brackets the reference prints"}
    golden = os.path.join(ROOT, "tests", "golden", "c1_pipeline_golden.npz")
    if os.path.exists(golden):
        g = np.load(golden)
        out["equals_oracle_golden"] = bool(int(g["top_votes"][0]) == int(pose.numVotes) and float(g["icp_residuals"][0]) == float(pose.residual)
                                           and int(g["n_object"]) == n_obj and int(g["n_edge"]) == n_edge)
    if with_cpu:
        import oracle_lib as O
        t0 = time.perf_counter()
        keep, _ = O.prep_crop(xyz, box, depth, intr)
        v = O.prep_voxel(xyz[keep], leaf)
        k2, _, _ = O.prep_sor(v, 50, sor)
        v = v[k2]
        nrm, curv = O.prep_normals(v, 30)
        o_obj = O.prep_to_mat(v, nrm)
        o_edge = O.prep_to_mat(v[curv > 0.03], nrm[curv > 0.03])
        t1 = time.perf_counter()
        ora = O.OracleDetector(0.025, 0.05).train_model(bottle)
        t2 = time.perf_counter()
        threads = min(O.max_threads(), usable_cpus())
        m = ora.match(o_obj, edge=o_edge, relative_scene_sample_step=0.05, relative_scene_distance=0.05, cluster=True, threads=threads)
        t3 = time.perf_counter()
        P, res, its = O.icp_refine(bottle, o_obj, [q["pose"] for q in m["poses"][:5]])
        t4 = time.perf_counter()
        out["cpu_oracles"] = {"frame_to_pose_ms": ((t1 - t0) + (t3 - t2) + (t4 - t3)) * 1e3, "prep_ms": (t1 - t0) * 1e3,
                              "match_ms": (t3 - t2) * 1e3, "icp_ms": (t4 - t3) * 1e3, "train_ms_once": (t2 - t1) * 1e3,
                              "threads": threads, "kind": "port", "runs": 1,
                              "note": "match on `threads` threads over reference points (where upstream puts its omp parallel for); the stage and ICP oracles "
                                      "use their own OpenMP loops (kNN, neighbour search)",
                              "same_result": bool(m["poses"][0]["num_votes"] == pose.numVotes and float(res[0]) == float(pose.residual))}
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing above imported torch or initialised HIP.
        sys.exit(launch_ranks(args, argv))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}\n")
        sys.exit(2)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

    if args.dry_launch:
        import datetime
        import torch.distributed as dist
        if os.environ.get("PPF_BENCH_DRY_FAIL_RANK") == str(rank):  # launcher test: a rank that dies before the rendezvous
            sys.exit(7)
        if world > 1:
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=120))
            dist.barrier()
        print(json.dumps({"dry_launch": True, "rank": rank, "local_rank": local_rank, "world": world,
                          "dist_world": dist.get_world_size() if world > 1 else 1}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    import datetime
    import numpy as np
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # rehearsal knobs (one-GPU box): PPF_BENCH_ONE_DEVICE=1 puts every rank on cuda:0, PPF_BENCH_BACKEND=gloo swaps
    # RCCL for gloo.  The driver's multi-GPU runs use neither: one rank per GPU over RCCL ("nccl").
    one_device = os.environ.get("PPF_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("PPF_BENCH_BACKEND", "nccl")
    device_index = 0 if one_device else local_rank
    if device_index >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{device_index}, the box shows {torch.cuda.device_count()} device(s)")
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # a short rendezvous timeout: a rank that never arrives must not hold the others for the default ten minutes
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index),
                                    timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=180))

    from yolo_ppf_pose_estimation_amd import _capi, parallel, workloads as W
    from yolo_ppf_pose_estimation_amd.detector import PPF3DDetector
    from yolo_ppf_pose_estimation_amd.device import BatchMatcher, Workspace

    assert torch.cuda.current_device() == device_index, "the rank's current device is not its own"
    bottle = W.bottle()
    if world > 1:
        # one line per rank on stderr before the first step: a wrong placement on an 8-GPU box shows in the run's tail
        def pci_bus_id(dev):
            try:
                import ctypes as C
                buf = C.create_string_buffer(64)
                hip = C.CDLL("libamdhip64.so")  # the runtime _capi.lib() made the process-wide one
                return buf.value.decode() if hip.hipDeviceGetPCIBusId(buf, 64, int(dev)) == 0 and buf.value else "?"
            except Exception:  # noqa: BLE001
                p = torch.cuda.get_device_properties(dev)
                return "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
        _capi.lib()
        probe = PPF3DDetector(0.2, 0.05).trainModel(bottle[::50])  # a throw-away table: where does THIS rank's engine put its memory?
        sys.stderr.write(f"bench.py rank {rank}/{world}: local_rank {local_rank} torch.cuda.current_device {torch.cuda.current_device()} "
                         f"ppf_model_get_device {probe.device()} pci {pci_bus_id(torch.cuda.current_device())} backend {backend} "
                         f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}\n")
        sys.stderr.flush()
        del probe
    step_stride = int(1.0 / W.SCENE_STEP)
    src_hash = kernel_source_hash()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_single(config, n_steps, n_warm, cells="auto", depth_arg=1, shard=False, scene=None):
        """C2 / C3 / C4: one crop per rank through ppf_match_device, `depth` crops in flight.  Returns the raw figures."""
        model_step = W.C2["model_step"] if config == "c2" else W.C4["model_step"]
        if scene is None:
            scene = (W.c2_scene() if world == 1 else W.c3_scene(rank)) if config == "c2" else W.c4_scene()
        n_scene = scene.shape[0]
        det = PPF3DDetector(model_step, W.REL_DISTANCE,
                            max_tile_refs=int(os.environ.get("PPF_TILE_REFS", "0"))).trainModel(bottle)
        info = det.info()
        assert det.device() == device_index, "the trained model does not live on this rank's device"
        d_scene = torch.from_numpy(scene).cuda()
        # Crops are independent, so consecutive steps CAN be software-pipelined like a serving loop would do.  The
        # default is depth 1: with overlap the per-kernel HIP-event times `roofline` is computed from are stretched.
        depth = max(1, min(depth_arg, max(n_warm, 1)))
        # HIP maps streams onto a few hardware queues round robin: two streams created late in a process can share one and
        # then do not overlap at all.  With as many streams as there are queues, consecutive steps sit on different queues
        # wherever the sequence starts.
        streams = [torch.cuda.Stream() for _ in range(1 if depth == 1 else 4)]
        wss = [Workspace(timing=True) for _ in range(depth)]
        if cells != "auto":  # 32: 32-bit cells for everything; 16: 16-bit cells first, always; limit: a learned per-item vote limit
            for w_ in wss:
                w_.set_option(_capi.PPF_OPT_ACC32, {"32": 1, "16": 2, "limit": 3}[cells])
        n_ref_total = (n_scene + step_stride - 1) // step_stride
        ref_kw = {"ref_offset": rank, "ref_stride": world, "skip_clustering": True} if shard else {}
        n_ref_max = (n_ref_total + world - 1) // world
        ws_cluster = Workspace() if shard else None  # rank 0 clusters the merged list here
        tally = {"votes": 0, "pairs": 0, "atomics": 0}
        ms = {"k_pairs": [], "k_group": [], "k_vote": [], "device_total": []}
        last, gathered = {}, {}

        def enqueue(i):
            s = streams[i % len(streams)]
            with torch.cuda.stream(s):
                wss[i % depth].match_device(det, d_scene.data_ptr(), n_scene, 6, W.SCENE_STEP, W.REL_DISTANCE,
                                            presampled=True, stream=s.cuda_stream, **ref_kw)

        def collect(i, timed):
            ws, s = wss[i % depth], streams[i % len(streams)]
            with torch.cuda.stream(s):
                st = ws.stats()  # waits for that step's stream (and re-runs a step whose scratch estimate was too small)
                if shard:
                    # the path's only collective: all_gather of each rank's raw per-reference poses (216 B each),
                    # device to device; rank 0 clusters the merged list (ppf_cluster_poses_device)
                    blk = ws.device_pose_block(n_ref_max, s.cuda_stream)
                    allp = parallel.gather_device(blk, dist)
                    if rank == 0:
                        merged = parallel.merge_reference_shards_device(allp, world, n_ref_total)
                        gathered["poses"] = ws_cluster.cluster_device(det, merged.data_ptr(), n_ref_total, n_scene // step_stride,
                                                                      stream=s.cuda_stream, top_k=W.TOP_K)
                else:
                    blk = ws.device_top_block(W.TOP_K, s.cuda_stream)  # top-5 clustered poses, still in HBM
                    gathered["poses"] = parallel.gather_device(blk, dist) if world > 1 else blk  # 5 x 216 B per rank
            if timed:
                ms["k_vote"].append(st["ms_vote_kernel"]); ms["k_pairs"].append(st["ms_pair_kernel"])
                ms["k_group"].append(st.get("ms_group_kernel", 0.0)); ms["device_total"].append(st["ms_total_device"])
                tally["votes"] += st["n_votes"]; tally["pairs"] += st["n_pairs"]
                tally["atomics"] += st.get("n_lds_atomics", 0)
                last.update(st)

        def run(n, timed):
            for i in range(n + depth - 1):
                if i < n:
                    enqueue(i)
                if i >= depth - 1:
                    collect(i - (depth - 1), timed)

        run(n_warm, False)
        sync()
        t0 = time.perf_counter()
        run(n_steps, True)
        sync()
        elapsed = time.perf_counter() - t0
        return {"elapsed": elapsed, "tally": tally, "kernel_ms": {k: float(np.mean(v)) for k, v in ms.items()}, "st": dict(last),
                "info": info, "n_scene": n_scene, "n_ref_total": n_ref_total, "depth": depth, "det": det, "scene": scene,
                "model_step": model_step}

    def run_c5(n_steps, n_warm):
        """C5: four resident tables x 8 crops per rank through ppf_batch_run (timing on: kernel times summed over the matches)."""
        models = W.c5_models()
        dets = [PPF3DDetector(W.C5_MODEL_STEP, W.REL_DISTANCE).trainModel(m) for m in models]
        crops = W.c5_crops(rank, models=models)
        d_crops = [torch.from_numpy(c).cuda() for c in crops]
        bm = BatchMatcher(lanes=int(os.environ.get("PPF_BATCH_LANES", "4")), timing=True)
        tally = {"votes": 0, "pairs": 0, "atomics": 0}
        ms = {"k_pairs": [], "k_group": [], "k_vote": []}

        def run(n, timed):
            for _ in range(n):
                res = bm.run_device(dets, [t.data_ptr() for t in d_crops], [c.shape[0] for c in crops], 6, W.SCENE_STEP,
                                    W.REL_DISTANCE, presampled=True, top_k=W.TOP_K)
                if world > 1:
                    parallel.gather_device(res["d_top"], dist)
                if timed:
                    tally["votes"] += res["n_votes"]; tally["pairs"] += res["n_pairs"]
                    tally["atomics"] += res.get("n_lds_atomics", 0)
                    ms["k_vote"].append(res["ms_vote_kernel"]); ms["k_pairs"].append(res["ms_pair_kernel"])
                    ms["k_group"].append(res["ms_group_kernel"])

        run(n_warm, False)
        sync()
        t0 = time.perf_counter()
        run(n_steps, True)
        sync()
        elapsed = time.perf_counter() - t0
        overlapped = {k: float(np.mean(v)) for k, v in ms.items()}
        # kernel times for the rooflines: the lanes of the timed steps overlap, which stretches every kernel's event time by
        # an amount that changes from run to run; one more step on a SINGLE lane gives each kernel's own time
        kernel_ms = overlapped
        if bm.lanes > 1:
            bm1 = BatchMatcher(lanes=1, timing=True)
            one = {}
            for rep in range(2):  # the first run of a batch context sizes its pools
                r1 = bm1.run_device(dets, [t.data_ptr() for t in d_crops], [c.shape[0] for c in crops], 6, W.SCENE_STEP,
                                    W.REL_DISTANCE, presampled=True, top_k=W.TOP_K)
                one = {"k_vote": r1["ms_vote_kernel"], "k_pairs": r1["ms_pair_kernel"], "k_group": r1["ms_group_kernel"]}
            kernel_ms = dict(one, lanes_overlapped=overlapped)
            del bm1
        return {"elapsed": elapsed, "tally": tally, "kernel_ms": kernel_ms,
                "info": dets[0].info(), "n_models": len(dets), "n_crops": len(crops), "lanes": bm.lanes,
                "n_scene": W.C2["n_points"]}

    def describe(config, shard=False):
        if config == "c2":
            return ("C2: bottle model (2,000 sampled pts, step 0.036) vs one 50,000-pt synthetic crop per GPU "
                    "(presampled, every point paired), reference stride 20 -> 2,500 reference points, 30 alpha bins"
                    + ("" if world == 1 else f"; C3: {world} crops, one per GPU, seeds 1000+rank"))
        if config == "c4":
            return ("C4: bottle model sampled at 0.0135 (~10k pts) vs one 200,000-pt synthetic scene (presampled), "
                    "reference stride 20 -> 10,000 reference points, 30 alpha bins"
                    + (f"; reference points strided over {world} ranks, raw poses all_gathered, clustered on rank 0"
                       if shard else ("" if world == 1 else f"; {world} replicas")))
        return (f"C5: 4 resident model tables (bottle, box, cylinder, torus at step 0.036) x "
                f"{W.C5_CROPS_PER_RANK} crops of 50,000 pts per GPU, every crop matched against every table "
                f"through ppf_batch_run (32 matches per step per GPU)")

    def leg_fields(config, r, n_steps):
        """The per-configuration figures shared by the main line and other_configs: times, rates, kernel times, rooflines."""
        votes_step = r["tally"]["votes"] // n_steps
        pairs_step = r["tally"]["pairs"] // n_steps
        atom_step = r["tally"]["atomics"] / n_steps if r["tally"]["atomics"] else None
        info = r["info"]
        n_ref = r["st"]["n_ref"] if "st" in r else r["n_models"] * r["n_crops"] * ((r["n_scene"] + step_stride - 1) // step_stride)
        abytes = algorithmic_bytes(n_ref, info["n_ref"], info["num_angles"], pairs_step, votes_step)
        roof, issue, ldsr = counter_rooflines(config, votes_step, r["kernel_ms"]["k_vote"], abytes, atom_step, src_hash)
        return {"ms_per_step": r["elapsed"] / n_steps * 1e3, "steps": n_steps,
                "pair_matches_per_s": r["tally"]["votes"] / r["elapsed"], "scene_pairs_per_s": r["tally"]["pairs"] / r["elapsed"],
                "votes_per_step_per_gpu": votes_step, "kernel_ms": r["kernel_ms"],
                "roofline": roof, "issue_roofline": issue, "lds_roofline": ldsr}

    shard_refs = args.config == "c4" and args.shard == "refs" and world > 1
    scaling = "strong" if shard_refs else "weak"
    if args.config == "c5":
        res = run_c5(args.steps, args.warmup)
    else:
        res = run_single(args.config, args.steps, args.warmup, cells=args.cells, depth_arg=args.pipeline_depth, shard=shard_refs)
    if args.config != "c2":
        args.no_cpu_baseline = True

    red_dev = "cuda" if backend == "nccl" else "cpu"
    t = torch.tensor([res["elapsed"]], dtype=torch.float64, device=red_dev)
    tot = torch.tensor([float(res["tally"]["votes"]), float(res["tally"]["pairs"])], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    all_votes, all_pairs = float(tot[0].item()), float(tot[1].item())

    if rank == 0:
        info = res["info"]
        line = {
            "metric": "PPF pair-matches/sec (accumulator votes) per cropped scene",
            "value": all_votes / elapsed,
            "unit": "pair-matches/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u32 votes / f64 pair features",
            "data": "synthetic",
            "config": {"workload": describe(args.config, shard_refs), "n_model": info["n_ref"], "n_scene": res["n_scene"],
                       "n_tiles": info["n_tiles"], "tile_refs": info["tile_refs"], "table_buckets": info["n_buckets"],
                       "table_entries": info["n_entries"],
                       "parallelism": (f"reference points strided over {world} ranks, RCCL all_gather of raw poses"
                                       if shard_refs else
                                       f"crops sharded 1/GPU x{world}, RCCL all_gather of top-{W.TOP_K} poses only")},
            "scene_pairs_per_s": all_pairs / elapsed,
            "kernel_source_hash": src_hash,
        }
        leg = leg_fields(args.config, res, args.steps)
        line["votes_per_step_per_gpu"] = leg["votes_per_step_per_gpu"]
        line["kernel_ms"] = leg["kernel_ms"]
        if world == 1:
            line["roofline"], line["issue_roofline"], line["lds_roofline"] = leg["roofline"], leg["issue_roofline"], leg["lds_roofline"]
        else:  # counter passes are single-GPU runs
            line["roofline"], line["issue_roofline"], line["lds_roofline"] = None, None, leg["lds_roofline"]
        if args.config == "c5":
            n_match = res["n_models"] * res["n_crops"]
            line["crops_per_s"] = world * res["n_crops"] * args.steps / elapsed
            line["matches_per_s"] = world * n_match * args.steps / elapsed
            line["config"]["batch_lanes"] = res["lanes"]
            line["cpu_baseline"] = None
        else:
            st = res["st"]
            line["config"]["n_ref"] = st["n_ref"]
            line["config"]["pipeline_depth"] = res["depth"]
            line["poses_per_s"] = world * st["n_ref"] * args.steps / elapsed
            line["clustered_poses"] = st.get("n_poses", 0)
            line["scratch_bytes"] = st.get("scratch_bytes", None)
            line["batches_per_step"] = st.get("n_batches", None)
            line["acc32_items_per_step"] = st.get("n_acc32_items", None)  # (reference point, tile)s voted with 32-bit cells (0: all 16-bit)
            if world == 1 and args.config == "c2" and not args.no_other_configs:
                # the entry the reference's detector.match() binds to, from host memory, on the same crop
                line["host_entry"] = host_entry(res["det"], res["scene"], st["n_ref"], line["ms_per_step"])
            if world == 1 and not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(res["model_step"], bottle, res["scene"], res["n_ref_total"], args.cpu_seconds)
            else:
                line["cpu_baseline"] = None
            if world == 1 and args.config == "c2" and not args.no_other_configs and res["depth"] == 1:
                # what a serving loop gets from keeping two independent crops in flight on two streams (k_pairs / k_group of
                # one under k_vote of the other).  Extra information only: `value`, kernel_ms and the rooflines above come
                # from the strictly serial steps, whose HIP-event kernel times are the kernels' own
                rp = run_single("c2", max(args.steps, 10), 3, depth_arg=2, scene=res["scene"])
                line["pipelined"] = {"crops_in_flight": rp["depth"], "ms_per_step": rp["elapsed"] / max(args.steps, 10) * 1e3,
                                     "pair_matches_per_s": rp["tally"]["votes"] / rp["elapsed"],
                                     "note": "same crop and model, steps enqueued two deep; per-kernel event times of such a run "
                                             "overlap and are not reported"}
                del rp
            if world == 1 and args.config == "c2" and not args.no_other_configs:
                # the other single-GPU configurations of BASELINE.json under the same clock: 2 steps each
                del res
                torch.cuda.empty_cache()
                other = {}
                r4 = run_single("c4", 2, 1)
                other["c4"] = dict(leg_fields("c4", r4, 2), workload=describe("c4"), n_model=r4["info"]["n_ref"],
                                   n_ref=r4["st"]["n_ref"], batches_per_step=r4["st"].get("n_batches"),
                                   acc32_items_per_step=r4["st"].get("n_acc32_items"),
                                   poses_per_s=r4["st"]["n_ref"] * 2 / r4["elapsed"])
                del r4
                r5 = run_c5(2, 1)
                other["c5"] = dict(leg_fields("c5", r5, 2), workload=describe("c5"), batch_lanes=r5["lanes"],
                                   crops_per_s=r5["n_crops"] * 2 / r5["elapsed"],
                                   matches_per_s=r5["n_models"] * r5["n_crops"] * 2 / r5["elapsed"],
                                   kernel_ms_note="sums over the 32 matches of one extra step run on a single lane (each kernel alone on the device); "
                                                  "lanes_overlapped = the same sums over the timed steps, whose lanes overlap and stretch every "
                                                  "kernel's event time")
                del r5
                torch.cuda.empty_cache()
                other["c1_pipeline"] = c1_pipeline(bottle, with_cpu=not args.no_cpu_baseline)
                line["other_configs"] = other
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
