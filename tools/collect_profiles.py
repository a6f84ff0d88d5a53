#!/usr/bin/env python3
"""Copy the judged summaries of a tools/gpu_round.sh run from gpurun_out/ into profiles/:
  profiles/<tag>_bench_line.json      the bench.py JSON line
  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats of the same command
  profiles/<tag>_pmc.json             HBM bytes per launch of the dominant kernel from the PMC passes
usage: collect_profiles.py <tag> [--dry]   (--dry prints, writes nothing)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
KERNEL = "vm_kernel"


def counter_mean(counter):
    vals = collections.defaultdict(list)
    paths = glob.glob(os.path.join(OUT, "pmc_" + counter, "**", "*counter_collection.csv"), recursive=True)
    # gpurun merges new files into gpurun_out/ without deleting older runs: only the newest counts
    for path in sorted(paths, key=os.path.getmtime)[-1:]:
        with open(path) as f:
            for r in csv.DictReader(f):
                if KERNEL in r["Kernel_Name"] and "tail" not in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    vals[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    if not vals:
        return None, None, 0
    name = max(vals, key=lambda k: len(vals[k]))
    return name, sum(vals[name]) / len(vals[name]), len(vals[name])


def main():
    tag = sys.argv[1]
    dry = "--dry" in sys.argv
    line = None
    with open(os.path.join(OUT, "bench.log")) as f:
        for row in f:
            if row.startswith("{"):
                line = json.loads(row)
    kname, fetch_kb, n_f = counter_mean("FETCH_SIZE")
    _, write_kb, n_w = counter_mean("WRITE_SIZE")
    pmc = None
    if fetch_kb is not None and write_kb is not None:
        # MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes
        # of a wide coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-byte streaming stores
        fetch_b, write_b = 2.0 * fetch_kb * 1024.0, write_kb * 1024.0
        cfg = line["config"] if line else {}
        pmc = {"kernel": kname, "side": (cfg.get("tile") or [None])[0], "tiles_per_launch": cfg.get("tiles_per_launch"),
               "FETCH_SIZE_KiB_mean": fetch_kb, "WRITE_SIZE_KiB_mean": write_kb, "dispatches": [n_f, n_w],
               "fetch_bytes_per_launch_corrected": fetch_b, "write_bytes_per_launch": write_b,
               "hbm_bytes_per_launch": fetch_b + write_b,
               "algorithmic_bytes_per_launch": (line or {}).get("roofline", {}).get("algorithmic_bytes_per_launch"),
               "correction": "FETCH_SIZE x2 (gfx950 counts 128-B read requests as 64 B), WRITE_SIZE x1, KiB -> bytes"}
    stats = sorted(glob.glob(os.path.join(OUT, "prof_" + tag, "**", "*kernel_stats.csv"), recursive=True),
                   key=os.path.getmtime)[-1:]
    print("bench line:", json.dumps(line)[:300] if line else None)
    print("pmc:", json.dumps(pmc))
    if stats:
        with open(stats[0]) as f:
            for i, row in enumerate(f):
                if i < 6:
                    print(row.rstrip()[:200])
    if dry:
        return
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    if line:
        with open(os.path.join(ROOT, "profiles", tag + "_bench_line.json"), "w") as f:
            json.dump(line, f, indent=1)
    if pmc:
        with open(os.path.join(ROOT, "profiles", tag + "_pmc.json"), "w") as f:
            json.dump(pmc, f, indent=1)
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))


if __name__ == "__main__":
    main()
