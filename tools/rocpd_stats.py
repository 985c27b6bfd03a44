#!/usr/bin/env python3
"""Per-kernel statistics of a rocprofv3 run kept as a rocpd SQLite file (rocprofv3 --kernel-trace --stats -d DIR -o NAME
writes DIR/NAME_results.db on this image): the table `--stats` prints, as CSV.

    python tools/rocpd_stats.py gpurun_out/prof/NAME_results.db [> profiles/rNN_kernel_stats.csv]
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [])
        a.append(e - s)
    total = sum(sum(v) for v in agg.values()) or 1
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f'"{name}",{len(v)},{sum(v)},{sum(v) / len(v):.1f},{100.0 * sum(v) / total:.2f},{min(v)},{max(v)}')


if __name__ == "__main__":
    main(sys.argv[1])
