#!/usr/bin/env python3
"""Where k_vote's work is, from the table and the crop themselves (numpy, CPU only; statistics, not parity: the pair keys are
binned with numpy's arccos, which may differ from the exact chain on a handful of pairs).

For a sample of C2's reference points: every run (one bucket, m hits of this reference point, c model entries in the bucket)
and the class k_vote files it under -- count tables (m >= 24 hits and >= 32 pair records) or direct votes -- with the votes,
entry visits and LDS wave-instructions each class costs under the kernel's scheme (17 counted atomics + 4 table-row reads
per entry pair block and table of <= S hits; own-cell votes m/Q per entry; one atomic per vote on the direct path), and the
same for alternative table sizes S and cell counts Q.

    python tools/vote_cost_model.py [--refs 40] [--json out.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from yolo_ppf_pose_estimation_amd import workloads as W  # noqa: E402

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def rotl(x, r):
    return ((x << np.uint64(r)) | (x >> np.uint64(64 - r))) & M64


def fmix(k):
    k ^= k >> np.uint64(33); k = (k * np.uint64(0xff51afd7ed558ccd)) & M64
    k ^= k >> np.uint64(33); k = (k * np.uint64(0xc4ceb9fe1a85ec53)) & M64
    k ^= k >> np.uint64(33)
    return k


def murmur_low32(k0, k1, k2, k3):
    """low 32 bits of h1 of MurmurHash3_x64_128 over four int32 keys, seed 42 (the reference's hashPPF)"""
    c1, c2 = np.uint64(0x87c37b91114253d5), np.uint64(0x4cf5ad432745937f)
    h1 = np.full(k0.shape, 42, np.uint64); h2 = h1.copy()
    u = lambda v: (v.astype(np.int64) & 0xFFFFFFFF).astype(np.uint64)
    a = u(k0) | (u(k1) << np.uint64(32)); b = u(k2) | (u(k3) << np.uint64(32))
    k = (a * c1) & M64; k = rotl(k, 31); k = (k * c2) & M64; h1 ^= k
    h1 = rotl(h1, 27); h1 = (h1 + h2) & M64; h1 = (h1 * np.uint64(5) + np.uint64(0x52dce729)) & M64
    k = (b * c2) & M64; k = rotl(k, 33); k = (k * c1) & M64; h2 ^= k
    h2 = rotl(h2, 31); h2 = (h2 + h1) & M64; h2 = (h2 * np.uint64(5) + np.uint64(0x38495ab5)) & M64
    h1 ^= np.uint64(16); h2 ^= np.uint64(16); h1 = (h1 + h2) & M64; h2 = (h2 + h1) & M64
    h1 = fmix(h1); h2 = fmix(h2); h1 = (h1 + h2) & M64
    return (h1 & np.uint64(0xFFFFFFFF)).astype(np.int64)


def runs_of_reference(scene, r, astep, dstep, slots, bucket_size):
    P = scene[:, :3].astype(np.float64); Nn = scene[:, 3:].astype(np.float64)
    d = P - P[r]; f3 = np.linalg.norm(d, axis=1); ok = f3 > 0
    dn = d[ok] / f3[ok, None]
    f0 = np.arccos(np.clip(dn @ Nn[r], -1, 1)); f1 = np.arccos(np.clip((Nn[ok] * dn).sum(1), -1, 1))
    f2 = np.arccos(np.clip(Nn[ok] @ Nn[r], -1, 1))
    k = murmur_low32((f0 / astep).astype(np.int64), (f1 / astep).astype(np.int64), (f2 / astep).astype(np.int64),
                     (f3[ok] / dstep).astype(np.int64)) % slots
    s, m = np.unique(k, return_counts=True)
    c = bucket_size[s]
    keep = c > 0
    return m[keep], c[keep]


def cost(m, c, S=191, Q=32, agg_min=24, rec_min=32):
    """LDS wave-instructions (64 lanes) of one reference point's runs under the kernel's scheme"""
    rec = (c + 1) // 2
    agg = (m >= agg_min) & (rec >= rec_min)
    blocks = (rec + 63) // 64                      # 64 pair records = 128 entries per wave pass
    tables = np.where(agg, (m + S - 1) // S, 0)
    counted = tables * blocks * (34 + 8)           # 2 x 17 counted atomics + the table-row / cell-range reads
    # own-cell loop: two hits per step, 2 reads + 4 atomics; the step count follows the fullest cell among the block's entries
    per_table_hits = np.where(agg, np.minimum(m, S), 0)
    own_steps = np.ceil(np.maximum(per_table_hits / Q * 2.0, 1.0) / 2.0)   # ~ max cell = 2 x mean
    own = tables * blocks * own_steps * 6 * agg
    direct = np.where(agg, 0, m * np.where(rec <= 32, 1, 2 * blocks))
    votes = m.astype(np.float64) * c
    return agg, counted.astype(np.float64), own.astype(np.float64), direct.astype(np.float64), votes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--refs", type=int, default=40)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    bottle = W.bottle()
    ora = O.OracleDetector(W.C2["model_step"], W.REL_DISTANCE).train_model(bottle)
    info = ora.info(); N, slots = info["n_ref"], info["slots"]
    hsh, _ = ora.pairs()
    off = ~np.eye(N, dtype=bool)
    slot_m = (hsh[off].astype(np.uint64) % np.uint64(slots)).astype(np.int64)
    bucket_size = np.bincount(slot_m, minlength=slots)
    scene = W.c2_scene()
    n_ref_total = scene.shape[0] // 20
    refs = [int(k * n_ref_total / a.refs) * 20 for k in range(a.refs)]
    M, Cc = [], []
    for r in refs:
        m, c = runs_of_reference(scene, r, info["angle_step"], info["distance_step"], slots, bucket_size)
        M.append(m); Cc.append(c)
    m = np.concatenate(M); c = np.concatenate(Cc)
    scale = n_ref_total / len(refs)
    agg, counted, own, direct, votes = cost(m, c)
    out = {"refs_sampled": len(refs), "runs_per_ref": len(m) / len(refs), "votes_per_step_est": float(votes.sum() * scale),
           "hits_per_ref": float(m.sum() / len(refs))}
    out["lds_wave_instr_per_step_est"] = {"counted_and_row_reads": float(counted.sum() * scale), "own_cell": float(own.sum() * scale),
                                          "direct": float(direct.sum() * scale)}
    # where the count-table runs sit by hits per run
    bins = [(24, 48), (48, 96), (96, 192), (192, 384), (384, 768), (768, 1536), (1536, 1 << 30)]
    tab = []
    for lo, hi in bins:
        sel = agg & (m >= lo) & (m < hi)
        tab.append({"hits_per_run": f"{lo}..{hi - 1 if hi < (1 << 30) else 'inf'}", "runs_per_ref": float(sel.sum() / len(refs)),
                    "share_of_votes": float(votes[sel].sum() / votes.sum()),
                    "share_of_counted_instr": float(counted[sel].sum() / max(counted.sum(), 1)),
                    "share_of_own_cell_instr": float(own[sel].sum() / max(own.sum(), 1))})
    out["count_table_runs_by_hits"] = tab
    dsel = ~agg
    out["direct_runs"] = {"share_of_votes": float(votes[dsel].sum() / votes.sum()), "runs_per_ref": float(dsel.sum() / len(refs)),
                          "share_with_more_than_32_records": float(votes[dsel & (c > 64)].sum() / max(votes[dsel].sum(), 1)),
                          "median_hits": float(np.median(m[dsel])), "mean_entries": float(c[dsel].mean())}
    # alternatives: bigger tables (16-bit counts) and more cells
    alts = []
    for S, Q in [(191, 32), (191, 64), (383, 32), (383, 64), (767, 64), (1023, 64), (1023, 128), (4095, 128)]:
        _, cn, ow, di, _ = cost(m, c, S=S, Q=Q)
        alts.append({"hits_per_table": S, "cells": Q, "counted": float(cn.sum() * scale), "own_cell": float(ow.sum() * scale),
                     "direct": float(di.sum() * scale), "total": float((cn.sum() + ow.sum() + di.sum()) * scale)})
    out["alternatives_lds_wave_instr_per_step"] = alts
    print(json.dumps(out, indent=1))
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
