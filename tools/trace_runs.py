import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    # per-call durations in order, grouped by kernel name prefix
    seq = [(r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows]
    # script order: per dtype, per sigma: 13 launches each of y then x kernels -> print mean of each run of identical names
    out, i = [], 0
    while i < len(seq):
        j = i
        tot = 0
        names = {}
        while j < len(seq) and j - i < 26 and "gauss" in seq[j][0]:
            names.setdefault(seq[j][0], []).append(seq[j][1]); j += 1
        if j == i:
            i += 1; continue
        print("  ".join("%s %.1f us" % (k[17:], sum(v) / len(v) / 1e3) for k, v in names.items()))
        i = j
