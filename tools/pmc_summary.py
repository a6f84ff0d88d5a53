#!/usr/bin/env python3
"""Average PMC counter values per kernel from rocprofv3 counter_collection CSVs.
usage: pmc_summary.py <dir> [kernel-substring]"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "vm_kernel"
agg = collections.defaultdict(list)
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if needle in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-48:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print("%-50s %-24s n=%-4d mean=%.1f" % (k, c, len(v), sum(v) / len(v)))
