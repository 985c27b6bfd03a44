#!/usr/bin/env python3
"""Per-kernel counter table of one or more `rocprofv3 --pmc` output directories (tools/vote_variants.sh).

    python tools/pmc_table.py [--kernel k_vote] [--steps 3] DIR ...

Counters are summed over the launches of the run and divided by the profiled steps; k_vote's two instantiations (16-bit
cells, 32-bit cells) are listed separately (the 32-bit one only repeats what overflowed)."""
import collections
import csv
import glob
import os
import sys


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].replace("void ", "").split("(")[0].strip()
            agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
    return agg


def main():
    args = sys.argv[1:]
    kernel, steps = "k_vote", 3.0
    while args and args[0].startswith("--"):
        if args[0] == "--kernel":
            kernel = args[1]
        elif args[0] == "--steps":
            steps = float(args[1])
        args = args[2:]
    for d in args:
        for name, c in sorted(load(d).items()):
            if not name.startswith(kernel):
                continue
            n = c.get("SQ_INSTS_LDS", 0) / steps
            if n < 1e3:
                continue
            idx, bank, addr = c.get("SQ_LDS_IDX_ACTIVE", 0) / steps, c.get("SQ_LDS_BANK_CONFLICT", 0) / steps, c.get("SQ_LDS_ADDR_CONFLICT", 0) / steps
            wc = c.get("SQ_WAVE_CYCLES", 0) / steps
            print(f"{os.path.basename(d.rstrip('/')):>22s} {name:24s} LDS insts {n:.4e}  array cyc/inst {idx / n:5.2f}  bank-conflict cyc/inst {bank / n:5.2f} "
                  f"({bank / max(idx, 1):.1%} of array cycles)  addr-conflict {addr / n:5.2f}  VALU insts {c.get('SQ_INSTS_VALU', 0) / steps:.4e}  "
                  f"WAIT_ANY {c.get('SQ_WAIT_ANY', 0) / max(wc * steps, 1):.1%}  WAIT_INST_LDS {c.get('SQ_WAIT_INST_LDS', 0) / max(wc * steps, 1):.1%} of wave cycles")


if __name__ == "__main__":
    main()
