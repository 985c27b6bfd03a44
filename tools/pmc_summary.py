#!/usr/bin/env python3
"""Summarise the counter passes of tools/pmc_vote.sh into the JSON bench.py reads (profiles/rNN_pmc_<workload>.json).

    python tools/pmc_summary.py gpurun_out/pmc_dir WORKLOAD N_VOTES_PER_STEP KVOTE_MS_PER_STEP [K_VOTE_LAUNCHES_PER_STEP] > profiles/r02_pmc_c2.json

Units as rocprofv3 reports them: FETCH_SIZE / WRITE_SIZE in KiB per launch; SQ_* summed over the waves the tool sampled
(SQ_WAVES tells how many: on this pool half of the launched waves), cycle-like SQ counters in quad-cycles
(MI355X_MICROARCH.md).  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 read-counter halving, same guide)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import kernel_source_hash  # noqa: E402  (the same hash bench.py checks a summary against)


STEPS_PROFILED = 3  # tools/pmc_vote.sh runs bench.py --steps 2 --warmup 1: three identical steps


def load(path):
    """Counters per kernel and STEP (sum over the launches of the run / steps).  The two k_vote instantiations (16-bit cells;
    32-bit cells: the launch that repeats what overflowed, or everything with --cells 32) are added up as "k_vote"."""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()  # "void k_vote<false, true>(MatchArgs)" -> k_vote
        agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
    return {k: {c: v / STEPS_PROFILED for c, v in d.items()} for k, d in agg.items()}


def main():
    d, workload, n_votes, kvote_ms = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
    launches = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    per = collections.defaultdict(dict)
    for f in sorted(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)):
        for k, c in load(f).items():
            per[k].update(c)
    keep = {k: per[k] for k in ("k_vote", "k_pairs", "k_group") if k in per}
    kv = keep.get("k_vote", {})
    out = {
        "source": "rocprofv3 --pmc, one pass per counter set (tools/pmc_vote.sh), python3 bench.py --steps 2 --warmup 1 "
                  "--no-cpu-baseline, MI355X; counters are sums over the launches of one step (k_vote: both instantiations)",
        "workload": workload,
        "n_votes_per_launch": n_votes if launches == 1 else None,
        "n_votes_per_step": n_votes,
        "k_vote_launches_per_step": launches,
        "k_vote_ms_per_step_at_collection": kvote_ms,
        "source_hash": kernel_source_hash(),  # of the tree the passes ran on: bench.py refuses the summary on any other
        "counters": keep,
    }
    if "FETCH_SIZE" in kv and "WRITE_SIZE" in kv:
        rep = (kv["FETCH_SIZE"] + kv["WRITE_SIZE"]) * 1024.0
        cor = (2 * kv["FETCH_SIZE"] + kv["WRITE_SIZE"]) * 1024.0
        out["hbm_bytes_per_step_k_vote_as_reported"] = rep
        out["hbm_bytes_note"] = "gfx950 read-counter halving: bytes = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md)"
        out["hbm_bytes_per_step_k_vote"] = cor
        out["hbm_gbs_k_vote"] = cor / (kvote_ms * 1e-3) / 1e9
    if "SQ_WAVE_CYCLES" in kv:
        simd_quad = kv["SQ_WAVE_CYCLES"] / 4.0  # 4 waves per SIMD resident for the whole launch (16-wave workgroups, 1 per CU)
        out["k_vote_issue"] = {
            "valu_busy_frac_of_simd_time": kv.get("SQ_ACTIVE_INST_VALU", 0) / simd_quad,
            "lds_busy_frac_of_simd_time": kv.get("SQ_ACTIVE_INST_LDS", 0) / simd_quad,
            "scalar_busy_frac_of_simd_time": kv.get("SQ_ACTIVE_INST_SCA", 0) / simd_quad,
            "any_inst_busy_frac_of_simd_time": kv.get("SQ_ACTIVE_INST_ANY", 0) / simd_quad,
            "wave_time_waiting_frac": kv.get("SQ_WAIT_ANY", 0) / kv["SQ_WAVE_CYCLES"],
            "wave_time_issue_stalled_frac": kv.get("SQ_WAIT_INST_ANY", 0) / kv["SQ_WAVE_CYCLES"],
            "lds_bank_conflict_frac_of_lds_cycles": kv.get("SQ_LDS_BANK_CONFLICT", 0) / max(kv.get("SQ_LDS_IDX_ACTIVE", 1), 1),
            "lds_array_cycles_per_instr": kv.get("SQ_LDS_IDX_ACTIVE", 0) / max(kv.get("SQ_INSTS_LDS", 1), 1),
            "waves_sampled": kv.get("SQ_WAVES"),
            "valu_wave_instructions_sampled": kv.get("SQ_INSTS_VALU"),
            "lds_wave_instructions_sampled": kv.get("SQ_INSTS_LDS"),
            "salu_wave_instructions_sampled": kv.get("SQ_INSTS_SALU"),
        }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
