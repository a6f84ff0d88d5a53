#!/usr/bin/env python3
"""Copy the judged summaries of a tools/gpu_profile.sh run from gpurun_out/ into profiles/:
  profiles/<tag>_<workload>_bench_line.json    the bench.py JSON line
  profiles/<tag>_<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command
  profiles/<tag>_<workload>_pmc.json           HBM bytes per launch of the dominant kernel from the PMC passes
                                               (+ workload, tiles per launch and the hash of the device code, which
                                               bench.py checks before it quotes the traffic)
usage: collect_profiles.py <tag> <workload> [--dry]   (--dry prints, writes nothing)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
sys.path.insert(0, ROOT)


def newest(pattern):
    paths = glob.glob(pattern, recursive=True)
    return sorted(paths, key=os.path.getmtime)[-1] if paths else None


def dominant_kernel(stats_csv):
    """(name, average ns, calls) of the kernel with the largest total duration."""
    best = None
    with open(stats_csv) as f:
        for r in csv.DictReader(f):
            total = float(r.get("TotalDurationNs") or r.get("TotalDuration(ns)") or 0)
            if best is None or total > best[3]:
                best = (r["Name"], float(r.get("AverageNs") or r.get("Average(ns)") or 0), int(float(r["Calls"])), total)
    return best[:3] if best else (None, None, 0)


def counter_mean(folder, counter, kernel):
    path = newest(os.path.join(folder, "**", "*counter_collection.csv"))
    if path is None:
        return None, 0
    vals = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and r["Kernel_Name"].split("(")[0].strip() == kernel.split("(")[0].strip():
                vals.append(float(r["Counter_Value"]))
    # the first dispatches of a run are warm-up steps of the same kernel: identical traffic, keep them
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    tag, workload = sys.argv[1], sys.argv[2]
    dry = "--dry" in sys.argv
    base = os.path.join(OUT, "prof_%s_%s" % (tag, workload))
    line = None
    with open(os.path.join(base, "bench.log")) as f:
        for row in f:
            if row.startswith("{"):
                line = json.loads(row)
    stats = newest(os.path.join(base, "stats", "**", "*kernel_stats.csv"))
    kernel, avg_ns, calls = dominant_kernel(stats) if stats else (None, None, 0)
    pmc = None
    if kernel:
        fetch_kb, n_f = counter_mean(os.path.join(base, "pmc_FETCH_SIZE"), "FETCH_SIZE", kernel)
        write_kb, n_w = counter_mean(os.path.join(base, "pmc_WRITE_SIZE"), "WRITE_SIZE", kernel)
        if fetch_kb is not None and write_kb is not None:
            import bench
            # MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes
            # of a wide coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-byte streaming stores
            fetch_b, write_b = 2.0 * fetch_kb * 1024.0, write_kb * 1024.0
            cfg = line["config"] if line else {}
            roof = (line or {}).get("roofline", {})
            pmc = {"workload": workload, "kernel": kernel, "tiles_per_launch": cfg.get("tiles_per_launch"),
                   "kernel_source_hash": bench.kernel_source_hash(),
                   "rocprof_avg_launch_us": round(avg_ns / 1e3, 2), "rocprof_calls": calls,
                   "bench_avg_launch_us": roof.get("avg_launch_us"),
                   "FETCH_SIZE_KiB_mean": fetch_kb, "WRITE_SIZE_KiB_mean": write_kb, "dispatches": [n_f, n_w],
                   "fetch_bytes_per_launch_corrected": fetch_b, "write_bytes_per_launch": write_b,
                   "hbm_bytes_per_launch": fetch_b + write_b,
                   "algorithmic_bytes_per_launch": roof.get("algorithmic_bytes_per_launch"),
                   "correction": "FETCH_SIZE x2 (gfx950 counts 128-B read requests as 64 B), WRITE_SIZE x1, KiB -> bytes"}
    print("bench line:", json.dumps(line)[:300] if line else None)
    print("dominant kernel:", kernel, avg_ns, calls)
    print("pmc:", json.dumps(pmc))
    if dry:
        return
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    stem = os.path.join(ROOT, "profiles", "%s_%s" % (tag, workload))
    if line:
        with open(stem + "_bench_line.json", "w") as f:
            json.dump(line, f, indent=1)
    if pmc:
        with open(stem + "_pmc.json", "w") as f:
            json.dump(pmc, f, indent=1)
    if stats:
        shutil.copy(stats, stem + "_kernel_stats.csv")


if __name__ == "__main__":
    main()
