#!/usr/bin/env python3
"""Top kernels of the newest rocprofv3 --kernel-trace --stats output under a directory.
usage: python tools/kstats.py gpurun_out/prof_xyz [rows]"""
import csv, glob, os, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "")
    name = name[: name.find("(")] if "(" in name else name
    print("%-72s calls %6s avg_us %9.2f total_ms %9.2f" % (name.replace("void ", "")[:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
print("total kernel ms %.2f, launches %d" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6, sum(int(r["Calls"]) for r in rows)))
