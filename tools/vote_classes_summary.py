#!/usr/bin/env python3
"""Per-class attribution of k_vote from the attribution builds (tools/vote_variants.sh on build_var/a_*.so).

    python tools/vote_classes_summary.py gpurun_out/r03_final/classes product r03 > profiles/r03_vote_classes.md

Every class is priced as (product build) - (build without that class): kernel time (HIP events, 10 steps), LDS
wave-instructions, LDS-array cycles, bank- and address-conflict cycles, VALU wave-instructions, and the share of wave time
parked (SQ_WAIT_ANY) of the build WITHOUT the class."""
import collections
import csv
import glob
import json
import os
import sys

STEPS = 3.0


def counters(d):
    agg = collections.defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_vote<false, false>" in row["Kernel_Name"]:
                agg[row["Counter_Name"]] += float(row["Counter_Value"]) / STEPS
    return agg


def main():
    root, prod = sys.argv[1], sys.argv[2]
    tag = sys.argv[3] if len(sys.argv) > 3 else "r03"

    def load(name):
        b = json.load(open(os.path.join(root, name + ".json")))
        c = counters(os.path.join(root, "pmc_" + name))
        return {"ms": b["kernel_ms"]["k_vote"], "lds": c["SQ_INSTS_LDS"], "idx": c["SQ_LDS_IDX_ACTIVE"], "bank": c["SQ_LDS_BANK_CONFLICT"],
                "addr": c["SQ_LDS_ADDR_CONFLICT"], "valu": c["SQ_INSTS_VALU"], "wait": c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1),
                "wait_lds": c["SQ_WAIT_INST_LDS"] / max(c["SQ_WAVE_CYCLES"], 1)}

    p = load(prod)
    rows = [("counted atomics + the table-row reads behind them (count-table items)", "a_counted0", -1),
            ("own-cell loop: one-by-one votes of an entry's own 1/64 cell (count-table items)", "a_owncell0", -1),
            ("direct votes on buckets of more than 32 records (loads + votes)", "a_dbig0", -1),
            ("   of which: the record loads alone (build that loads but does not vote, minus the one that does neither)", ("a_dbig2", "a_dbig0"), +1),
            ("direct votes on buckets of at most 32 records", "a_dsmall0", -1)]
    print("# k_vote by class of work (C2, MI355X)\n")
    print(f"Product build `{prod}`: k_vote {p['ms']:.3f} ms, {p['lds'] / 1e6:.1f} M LDS wave-instructions, {p['idx'] / p['lds']:.2f} LDS-array cycles per "
          f"instruction of which {p['bank'] / p['lds']:.2f} bank-conflict and {p['addr'] / p['lds']:.2f} address-conflict cycles, {p['valu'] / 1e6:.0f} M VALU "
          f"wave-instructions, {p['wait']:.1%} of wave time parked (SQ_WAIT_ANY), {p['wait_lds']:.1%} stalled on LDS issue (SQ_WAIT_INST_LDS).\n")
    print("Each class = product minus a build that leaves the class out (`PPF_ABL_*`, ppf_match_kernels.h; wrong votes, the 16-bit overflow check off): "
          "time includes the latency the class exposes.\n")
    print("| class | k_vote ms | LDS wave-instr (M) | array cycles / instr of the class | bank-conflict cycles (M) | share of the product's conflict cycles | "
          "address-conflict cycles (M) | VALU wave-instr (M) | SQ_WAIT_ANY without the class |")
    print("|---|---|---|---|---|---|---|---|---|")
    out = {"product": p, "classes": {}}
    tot = collections.defaultdict(float)
    for label, name, sign in rows:
        if isinstance(name, tuple):
            a, b = load(name[0]), load(name[1])
            d = {k: a[k] - b[k] for k in ("ms", "lds", "idx", "bank", "addr", "valu")}
            wait = a["wait"]
        else:
            v = load(name)
            d = {k: (p[k] - v[k]) * (1 if sign < 0 else -1) for k in ("ms", "lds", "idx", "bank", "addr", "valu")}
            wait = v["wait"]
        out["classes"][label] = dict(d, wait_any_without=wait)
        if not isinstance(name, tuple):
            for k in d:
                tot[k] += d[k]
        per = f"{d['idx'] / d['lds']:.2f}" if d["lds"] > 1e5 else "-"
        print(f"| {label} | {d['ms']:.3f} | {d['lds'] / 1e6:.1f} | {per} | {d['bank'] / 1e6:.0f} | {d['bank'] / p['bank']:.1%} | "
              f"{d['addr'] / 1e6:.0f} | {d['valu'] / 1e6:.0f} | {wait:.1%} |")
    n = load("a_none")
    print(f"| everything else: staging, claims, item look-up, record loads and table copies of count-table items, clear, scan (build with all four vote "
          f"classes out) | {n['ms']:.3f} | {n['lds'] / 1e6:.1f} | {n['idx'] / n['lds']:.2f} | {n['bank'] / 1e6:.0f} | {n['bank'] / p['bank']:.1%} | {n['addr'] / 1e6:.0f} | "
          f"{n['valu'] / 1e6:.0f} | {n['wait']:.1%} |")
    print(f"| sum of the rows (without the sub-row) | {tot['ms'] + n['ms']:.3f} | {(tot['lds'] + n['lds']) / 1e6:.1f} | | {(tot['bank'] + n['bank']) / 1e6:.0f} | "
          f"{(tot['bank'] + n['bank']) / p['bank']:.1%} | {(tot['addr'] + n['addr']) / 1e6:.0f} | {(tot['valu'] + n['valu']) / 1e6:.0f} | |")
    for extra, lab in (("a_aggonly", "count-table items only (no direct votes at all)"), ("a_directonly", "direct items only (no counted atomics, no own-cell loop)")):
        v = load(extra)
        print(f"\n`{extra}` — {lab}: {v['ms']:.3f} ms, {v['lds'] / 1e6:.1f} M LDS wave-instructions at {v['idx'] / v['lds']:.2f} array cycles, "
              f"{v['valu'] / 1e6:.0f} M VALU, SQ_WAIT_ANY {v['wait']:.1%}, SQ_WAIT_INST_LDS {v['wait_lds']:.1%}.")
        out[extra] = v
    print("\nHow to read it (arithmetic in DESIGN.md section 4): the LDS pipe charges an atomic or a store 4 cycles and a read 2.3 whatever its lanes "
          "do, plus what its array cycles exceed that by (profiles/r03_ubench_lds_counters.md), so a class whose instructions average 4 array cycles or "
          "fewer pays nothing for its conflicts: the counted atomics own the largest share of the conflict cycles and none of their cost; the direct "
          "votes' conflicts (above 4.5 array cycles per instruction) are the ones that cost.  The times add up to the product's: the classes do not hide "
          "each other, a wave's work items are a serial chain (claim, look-up, loads, votes) and neither the LDS pipe nor the VALU is saturated.  "
          "The count tables themselves are built by k_tables (its own line in the kernel statistics), once per run.")
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", f"{tag}_vote_classes.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
