#!/usr/bin/env python3
"""print the top kernels of a rocprofv3 --stats run: python tools/kstat.py DIR [n_proofs]"""
import csv, glob, sys
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:22]:
        print("%-58s calls %5s avg %8.3f ms  per-proof %7.2f ms  %5s%%" % (r["Name"].split("(")[0][-58:], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6 / n, r["Percentage"]))
