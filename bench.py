#!/usr/bin/env python3
"""bench.py — PPF pair-matches/s of the MI355X voting engine on BASELINE.json's workload.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one synthetic YOLO crop per rank: pair features + hash +
table lookup + Hough voting (k_vote) + argmax + pose assembly (k_finalize) + pose clustering, with
the crop and the model table already resident in HBM.  At N=1 the workload is BASELINE.json
configs[1] (C2): the bottle model sampled to 2,000 points vs one 50,000-point synthetic crop
(seed 12345), presampled so all 50k points vote, reference stride 20 -> 2,500 reference points.
At N>1 every rank matches its own crop (config C3, seeds 1000+rank; weak scaling) and the only
collective is one all_gather (RCCL) of each rank's top poses per step.

The line printed by rank 0 carries
  value        whole-job pair-matches (accumulator increments, exact integer) per second
  roofline     the voting kernel against the HBM roofline: algorithmic bytes per launch
               (SURVEY.md §8d / DESIGN.md §5) / its average device time measured with HIP events
               on the launch stream
  cpu_baseline the CPU oracle (kind "port": our restatement of the reference's library path,
               OpenMP over reference points where upstream parallelises) timed on a bounded
               sample of the same workload, N=1 only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# ds_add_u32 ceiling measured with profiles/ubench_valu_lds.hip: 14.6 lanes/clk/CU x 256 CUs x 2.4 GHz
LDS_ATOMIC_PEAK = 8.97e12

MODEL_STEP = 0.036      # bottle -> 2,000 sampled model points
SCENE_POINTS = 50000
SCENE_STEP = 1.0 / 20.0  # reference stride 20 -> 2,500 reference points
TOP_K = 5                # poses kept per crop (CloudProcessing.h:455,508)


def algorithmic_bytes(n_ref, n_model, num_angles, n_pairs, n_votes):
    """per reference point 24 B (its xyzn) + accumulator clear and scan
    2 x 4 x N_m x A + 12 B result; per scene pair 24 B (xyzn) + 8 B (bucket header); per vote 8 B
    model entry + 8 B accumulator read-modify-write."""
    return n_ref * (24 + 2 * 4 * n_model * num_angles + 12) + n_pairs * 32 + n_votes * 16


def measured_traffic(n_votes):
    """HBM/fabric bytes per k_vote launch from the committed PMC passes (profiles/*_pmc_traffic.json), if they
    were taken on this exact workload; otherwise None.  bench.py cannot collect PMC counters on itself."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            t = json.load(open(f))
            if t.get("n_votes_per_launch") == n_votes:
                best = t["hbm_bytes_per_launch_k_vote"]["gfx950_corrected_2xFETCH"]
        except Exception:
            pass
    return best


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


def cpu_baseline(bottle, scene, n_ref_total, target_seconds=15.0):
    """Oracle (CPU restatement) on a bounded sample of the step's reference points."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    threads = min(oracle_lib.max_threads(), usable_cpus())
    ora = oracle_lib.OracleDetector(MODEL_STEP, 0.05).train_model(bottle)
    step = int(1.0 / SCENE_STEP)
    # calibrate on 2 reference points per thread, then size the sample for ~target_seconds
    probe = [(k * (n_ref_total // (2 * threads))) * step for k in range(2 * threads)]
    t0 = time.perf_counter()
    r = ora.match(scene, relative_scene_sample_step=SCENE_STEP, presampled=True, ref_list=probe, threads=threads,
                  cluster=False)
    dt = time.perf_counter() - t0
    n_sample = int(min(n_ref_total, max(2 * threads, (target_seconds / max(dt, 1e-3)) * len(probe))))
    n_sample = max(threads, (n_sample // threads) * threads)
    refs = [int(k * n_ref_total / n_sample) * step for k in range(n_sample)]
    t0 = time.perf_counter()
    r = ora.match(scene, relative_scene_sample_step=SCENE_STEP, presampled=True, ref_list=refs, threads=threads,
                  cluster=False)
    dt = time.perf_counter() - t0
    votes = int(r["votes_per_ref"].sum())
    return {
        "value": votes / dt,
        "unit": "pair-matches/s",
        "cores": threads,
        "host_cpus_visible": os.cpu_count(),
        "kind": "port",
        "sample": f"{n_sample} of {n_ref_total} reference points (evenly spaced) of the same crop, all "
                  f"{scene.shape[0]} paired points each, {votes} pair-matches in {dt:.2f} s, "
                  f"{n_sample / dt:.2f} poses/s",
        "poses_per_s": n_sample / dt,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", default="c2", choices=["c2", "c4"],
                    help="c2 (default, BASELINE configs[1]: the contract's workload); c4: 10k-pt model vs 200k-pt scene "
                         "(HBM-resident stress, informational: no cpu_baseline)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--pipeline-depth", type=int, default=1,
                    help="crops in flight: 1 (default) = strictly one after another, so the HIP-event kernel times "
                         "behind `roofline` are clean; 2-3 overlap independent crops on separate streams (+4-6 %% "
                         "throughput, but concurrent kernels stretch each other's event times)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # rehearsal knobs (one-GPU box): PPF_BENCH_ONE_DEVICE=1 puts every rank on cuda:0, PPF_BENCH_BACKEND=gloo swaps
    # RCCL for gloo.  The driver's multi-GPU runs use neither: one rank per GPU over RCCL ("nccl").
    one_device = os.environ.get("PPF_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("PPF_BENCH_BACKEND", "nccl")
    device_index = 0 if one_device else local_rank
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)

    from yolo_ppf_pose_estimation_amd import parallel, synth
    from yolo_ppf_pose_estimation_amd._capi import Pose
    from yolo_ppf_pose_estimation_amd.detector import PPF3DDetector
    from yolo_ppf_pose_estimation_amd.device import Workspace
    import ctypes as C

    global MODEL_STEP, SCENE_POINTS
    n_instances = 1
    if args.config == "c4":
        MODEL_STEP, SCENE_POINTS, n_instances = 0.0135, 200000, 2
        args.no_cpu_baseline = True
    bottle = np.load(os.path.join(ROOT, "tests", "golden", "bottle_model_xyzn.npy"))
    det = PPF3DDetector(MODEL_STEP, 0.05, max_tile_refs=int(os.environ.get("PPF_TILE_REFS", "0"))).trainModel(bottle)
    info = det.info()
    seed = (12345 if args.config == "c2" else 4) if world == 1 else 1000 + rank
    scene, _ = synth.make_scene(bottle, n_points=SCENE_POINTS, seed=seed, n_instances=n_instances)
    d_scene = torch.from_numpy(scene).cuda()
    # Crops are independent, so consecutive steps CAN be software-pipelined like a serving loop would do: step i is
    # enqueued on one of `depth` (stream, workspace) pairs before the clustered poses of step i-depth+1 are read back,
    # which lets the single-workgroup clustering tail of one crop overlap the voting of the next.  The default is
    # depth 1 (no overlap): with overlap the per-kernel HIP-event times that `roofline` is computed from are no longer
    # the kernel's own.
    depth = max(1, min(args.pipeline_depth, max(args.warmup, 1)))
    streams = [torch.cuda.Stream() for _ in range(depth)]
    wss = [Workspace(timing=True) for _ in range(depth)]
    n_ref_total = (SCENE_POINTS + int(1.0 / SCENE_STEP) - 1) // int(1.0 / SCENE_STEP)

    def enqueue(i):
        s = streams[i % depth]
        with torch.cuda.stream(s):
            wss[i % depth].match_device(det, d_scene.data_ptr(), SCENE_POINTS, 6, SCENE_STEP, 0.05, presampled=True,
                                        stream=s.cuda_stream)

    def collect(i):
        with torch.cuda.stream(streams[i % depth]):
            # waits for that step's stream; clustering already ran on the device, only the clustered poses come back
            fin, k_top, n_clusters, st = wss[i % depth].top_poses(TOP_K)
            if world > 1:
                # the path's only collective: all_gather (RCCL) of each rank's top poses, 5 x 216 B per rank
                parallel.gather_poses(parallel.poses_to_array(fin, k_top, TOP_K), device="cuda" if backend == "nccl" else None)
        return {"stats": st, "n_clusters": n_clusters}

    def run(n_steps, on_result=None):
        res = None
        for i in range(n_steps + depth - 1):
            if i < n_steps:
                enqueue(i)
            if i >= depth - 1:
                res = collect(i - (depth - 1))
                if on_result:
                    on_result(res)
        return res

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    sync()
    vote_ms, pair_ms, tally = [], [], [0, 0]

    def account(r):
        st = r["stats"]
        vote_ms.append(st["ms_vote_kernel"])
        pair_ms.append(st["ms_pair_kernel"])
        tally[0] += st["n_votes"]
        tally[1] += st["n_pairs"]

    t0 = time.perf_counter()
    res = run(args.steps, account)
    sync()
    elapsed = time.perf_counter() - t0
    votes, pairs = tally

    n_poses_clustered = res["n_clusters"]
    red_dev = "cuda" if backend == "nccl" else "cpu"
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    tot = torch.tensor([float(votes), float(pairs)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    all_votes, all_pairs = float(tot[0].item()), float(tot[1].item())

    if rank == 0:
        st = res["stats"]
        avg_vote_s = float(np.mean(vote_ms)) * 1e-3
        abytes = algorithmic_bytes(st["n_ref"], info["n_ref"], info["num_angles"], st["n_pairs"], st["n_votes"])
        hbm_only = abytes - 8 * st["n_votes"] - st["n_ref"] * 2 * 4 * info["n_ref"] * info["num_angles"]
        achieved = abytes / avg_vote_s / 1e9
        line = {
            "metric": "PPF pair-matches/sec (accumulator votes) per cropped scene",
            "value": all_votes / elapsed,
            "unit": "pair-matches/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 votes / f64 pair features",
            "data": "synthetic",
            "config": {
                "workload": ("C2: bottle model (2,000 sampled pts, step 0.036) vs one 50,000-pt synthetic crop per GPU "
                             "(presampled, every point paired), reference stride 20 -> 2,500 reference points, "
                             "30 alpha bins" if args.config == "c2" else
                             "C4: bottle model sampled at 0.0135 (~10k pts) vs one 200,000-pt synthetic scene "
                             "(presampled), reference stride 20 -> 10,000 reference points, 30 alpha bins")
                            + ("" if world == 1 else f"; C3: {world} crops, one per GPU, seeds 1000+rank"),
                "n_model": info["n_ref"], "n_scene": SCENE_POINTS, "n_ref": st["n_ref"],
                "n_tiles": info["n_tiles"], "tile_refs": info["tile_refs"],
                "table_buckets": info["n_buckets"], "table_entries": info["n_entries"],
                "parallelism": f"crops sharded 1/GPU x{world}, RCCL all_gather of top-{TOP_K} poses only",
                "pipeline_depth": depth,
            },
            "poses_per_s": world * st["n_ref"] * args.steps / elapsed,
            "scene_pairs_per_s": all_pairs / elapsed,
            "votes_per_step_per_gpu": st["n_votes"],
            "clustered_poses": n_poses_clustered,
            "kernel_ms": {"k_pairs": float(np.mean(pair_ms)), "k_vote": float(np.mean(vote_ms)),
                          "device_total": st["ms_total_device"]},
            "roofline": {
                "bound": "hbm",
                "kernel": "k_vote",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(st["n_votes"]) if world == 1 else None,
                "algorithmic_bytes_per_launch": abytes,
                "hbm_only_algorithmic_bytes_per_launch": hbm_only,
                "avg_kernel_ms": avg_vote_s * 1e3,
                "note": "16 B/vote counts the 8 B accumulator RMW although the accumulator is LDS-resident; "
                        "hbm_only_* removes the LDS part (8 B/vote + clear/scan)",
            },
        }
        # what actually bounds k_vote: one LDS atomic per vote (plus ~5 VALU); reported beside the contract's
        # HBM roofline because the accumulator never leaves LDS
        line["lds_atomic_roofline"] = {"kernel": "k_vote", "achieved": st["n_votes"] / avg_vote_s, "peak": LDS_ATOMIC_PEAK,
                                       "unit": "votes/s", "frac": st["n_votes"] / avg_vote_s / LDS_ATOMIC_PEAK,
                                       "source": "profiles/r01_ubench_valu_lds.txt"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(bottle, scene, n_ref_total, args.cpu_seconds)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
