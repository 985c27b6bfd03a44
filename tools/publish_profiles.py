#!/usr/bin/env python3
"""Turn a tools/collect_profiles.sh output directory into the tracked files under profiles/ (round tag as argument).

    python tools/publish_profiles.py gpurun_out/r03_final r03            # bench lines, kernel stats, PMC summaries, vote classes
    python tools/publish_profiles.py gpurun_out/r03_final r03 --lines    # only the bench lines (after re-running them so that
                                                                         # their roofline objects read the new summaries)

The PMC summaries record the hash of the kernel sources of THIS tree (tools/pmc_summary.py): publish from the tree the passes
ran on, and do not touch yolo_ppf_pose_estimation_amd/csrc or include/ppf_hip.h / ppf_detmath.h afterwards, or bench.py will
(rightly) call the counters stale.
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
    names = ["bench_c2", "bench_c4", "bench_c5", "bench_c3_2ranks_one_device_gloo", "bench_c4_refs_2ranks_one_device_gloo",
             "bench_c5_2ranks_one_device_gloo"]
    lines = {}
    for n in names:
        p = os.path.join(src, n + ".json")
        if not os.path.exists(p):
            continue
        line = last_line(p)
        lines[n] = json.loads(line)
        open(os.path.join(ROOT, "profiles", f"{tag}_{n}.json"), "w").write(line + "\n")
        d = lines[n]
        print(n, "ms/step %.3f" % d["ms_per_step"], "value %.3e" % d["value"], d.get("kernel_ms"), d.get("crops_per_s"))
    if "bench_c2" in lines and lines["bench_c2"].get("host_entry"):
        h = lines["bench_c2"]["host_entry"]
        print("host entry: %.3f ms per match (resident step %.3f ms), first call %.2f ms" % (h["ms_per_match"], h["resident_step_ms"], h["ms_first_call"]))
    if only_lines:
        return
    for w in ("c2", "c4", "c5"):
        tdir = os.path.join(src, f"trace_{w}")
        if os.path.isdir(tdir):
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(tdir) for f in fs if f.endswith(".db")]
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_stats.py")] + dbs, check=True, capture_output=True, text=True).stdout
            open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_{w}.csv"), "w").write(out)
        pdir = os.path.join(src, f"pmc_{w}")
        if os.path.isdir(pdir) and f"bench_{w}" in lines:
            d = lines[f"bench_{w}"]
            launches = d.get("batches_per_step") or (d["config"].get("batch_lanes") and 32) or 1
            summary = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), pdir, w.upper(),
                                      str(d["votes_per_step_per_gpu"]), str(d["kernel_ms"]["k_vote"]), str(launches)],
                                     check=True, capture_output=True, text=True).stdout
            open(os.path.join(ROOT, "profiles", f"{tag}_pmc_{w}.json"), "w").write(summary)
            s = json.loads(summary)
            print(w, "k_vote fabric GB/s %.0f" % s["hbm_gbs_k_vote"], {k: round(v, 3) for k, v in s["k_vote_issue"].items() if isinstance(v, float) and v < 10})
    # the reference's real call (frame -> pose) and its ICP step
    for n in ("pipeline_timing", "icp_timing", "icp_timing_legacy"):
        q = os.path.join(src, n + ".json")
        if os.path.exists(q):
            open(os.path.join(ROOT, "profiles", f"{tag}_{n}.json"), "w").write(last_line(q) + "\n")
            d = json.loads(last_line(q))
            print(n, d.get("frame_to_pose_ms") or d.get("gpu_seconds"), d.get("of_which_ms", ""))
    icp_csv = [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(src, "trace_icp")) for f in fs if f.endswith("kernel_stats.csv")]
    if icp_csv:
        rows = [ln for ln in open(icp_csv[0]).read().splitlines() if ln.startswith('"Name"') or "icp" in ln]
        open(os.path.join(ROOT, "profiles", f"{tag}_icp_kernel_stats.csv"), "w").write("\n".join(rows) + "\n")
    cdir = os.path.join(src, "classes")
    if os.path.isdir(cdir):
        md = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "vote_classes_summary.py"), cdir, "product", tag], check=True,
                            capture_output=True, text=True).stdout
        open(os.path.join(ROOT, "profiles", f"{tag}_vote_classes.md"), "w").write(md)
        print(md.splitlines()[2])


if __name__ == "__main__":
    main()
