#!/usr/bin/env python3
"""Average per-launch PMC counters of one kernel from a tools/pmc_vote.sh output directory, as JSON.
    python tools/pmc_parse.py gpurun_out/pmc_xyz [kernel-name-prefix]"""
import collections
import csv
import glob
import json
import sys


def load(path, kern):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0].strip() == kern:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    d, kern = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_vote")
    out = {}
    for f in sorted(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)):
        out.update(load(f, kern))
    print(json.dumps({"kernel": kern, "counters_per_launch": out}, indent=1))


if __name__ == "__main__":
    main()
