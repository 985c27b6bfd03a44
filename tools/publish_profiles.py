#!/usr/bin/env python3
"""Turn a tools/collect_profiles.sh output directory into the tracked files under profiles/ (round tag as argument).

    python tools/publish_profiles.py gpurun_out/r02_final r02            # bench lines, kernel stats, PMC summaries
    python tools/publish_profiles.py gpurun_out/r02_final r02 --lines    # only the c2 / c4 bench lines (after re-running them
                                                                         # so that their roofline objects read the new summaries)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_line(path):
    return open(path).read().strip().splitlines()[-1]


def main():
    src, tag = sys.argv[1], sys.argv[2]
    only_lines = "--lines" in sys.argv
    names = ["bench_c2", "bench_c4"] if only_lines else [
        "bench_c2", "bench_c4", "bench_c5", "bench_c3_2ranks_one_device_gloo", "bench_c4_refs_2ranks_one_device_gloo",
        "bench_c5_2ranks_one_device_gloo"]
    lines = {}
    for n in names:
        line = last_line(os.path.join(src, n + ".json"))
        lines[n] = json.loads(line)
        open(os.path.join(ROOT, "profiles", f"{tag}_{n}.json"), "w").write(line + "\n")
        d = lines[n]
        print(n, "ms/step %.3f" % d["ms_per_step"], "value %.3e" % d["value"], d.get("kernel_ms"), d.get("crops_per_s"))
    if only_lines:
        return
    for w in ("c2", "c4"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_stats.py")] +
                             [os.path.join(src, f"trace_{w}", f) for f in os.listdir(os.path.join(src, f"trace_{w}")) if f.endswith(".db")],
                             check=True, capture_output=True, text=True).stdout
        open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_{w}.csv"), "w").write(out)
        d = lines[f"bench_{w}"]
        summary = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(src, f"pmc_{w}"), w.upper(),
                                  str(d["votes_per_step_per_gpu"]), str(d["kernel_ms"]["k_vote"]), str(d.get("batches_per_step") or 1)],
                                 check=True, capture_output=True, text=True).stdout
        open(os.path.join(ROOT, "profiles", f"{tag}_pmc_{w}.json"), "w").write(summary)
        s = json.loads(summary)
        print(w, "k_vote fabric GB/s %.0f" % s["hbm_gbs_k_vote"], {k: round(v, 3) for k, v in s["k_vote_issue"].items() if isinstance(v, float) and v < 10})


if __name__ == "__main__":
    main()
