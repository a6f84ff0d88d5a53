"""Prints the measurement table of MEASUREMENTS.md section 8 for one round from profiles/<tag>_*_bench_line.json and
profiles/<tag>_*_pmc.json (run after tools/collect_profiles.py):  python tools/regen_design_table.py r04"""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
ALGO = {"chain5": 3221225472, "mosaic": 4563402752, "reclass": 1342177280, "bilinear": 5368709120, "tiler": 3221225472}
LABEL = {"chain5": "**chain5** (configs[1], headline): 16 tiles of 4096² f32 per launch, 12 B/px",
         "mosaic": "mosaic (configs[4]): 10-Block view, 256 tiles, 17 B/px",
         "reclass": "reclass (configs[2]): Reclassify + Mask, 4 × 8192² int32, 5 B/px",
         "bilinear": "bilinear (configs[3]): 16384² → 32768², 5 B/out px",
         "tiler": "tiler: chain under `RasterTiler(4096)` over resident 16384² sources"}


def grp(n):
    return f"{int(round(n)):,}".replace(",", " ")


print("| workload (BASELINE configs) | Mpix/s | dominant kernel | µs / launch (events / rocprof / kernel-only) | algorithmic B / launch "
      "| frac of 8 TB/s (rocprof) | PMC traffic ÷ algorithmic |")
print("|---|---|---|---|---|---|---|")
for wl in ("chain5", "mosaic", "reclass", "bilinear", "tiler"):
    b = json.load(open("profiles/%s_%s_bench_line.json" % (tag, wl)))
    pm = json.load(open("profiles/%s_%s_pmc.json" % (tag, wl)))
    r = b["roofline"]
    only = "%.1f" % r["kernel_only_us"] if r.get("kernel_only_us") else "–"
    kernel = pm["kernel"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
    print("| %s | %s | `%s` | %.1f / %.1f / %s | %s | %.3f | %.4f |" % (
        LABEL[wl], grp(b["value"]), kernel, r["avg_launch_us"], pm["rocprof_avg_launch_us"], only, grp(ALGO[wl]),
        ALGO[wl] / pm["rocprof_avg_launch_us"] / 1e6 / 8, pm["hbm_bytes_per_launch"] / pm["algorithmic_bytes_per_launch"]))
print("device code:", json.load(open("profiles/%s_chain5_pmc.json" % tag))["kernel_source_hash"])
