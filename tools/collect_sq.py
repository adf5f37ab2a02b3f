#!/usr/bin/env python3
"""Per-kernel sums of SQ counters from rocprofv3 --pmc passes (profiles/r2_pmc_sq_by_kernel.csv).

  python tools/collect_sq.py OUT.csv DIR [DIR ...]     # each DIR = one rocprofv3 -d output (one --pmc pass)"""
import collections
import csv
import glob
import sys

out, dirs = sys.argv[1], sys.argv[2:]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
counters = []
for k, d in enumerate(dirs):
    seen = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] not in counters:
                counters.append(r["Counter_Name"])
            if k == 0:
                seen.add((name, r["Dispatch_Id"]))
    for name, _ in seen:
        calls[name] += 1
with open(out, "w") as f:
    f.write("kernel,dispatches," + ",".join(counters) + "\n")
    key = counters[0]
    for name in sorted(tot, key=lambda n: -tot[n].get(key, 0)):
        f.write('"%s",%d,' % (name, calls[name]) + ",".join("%.0f" % tot[name].get(c, 0) for c in counters) + "\n")
print("wrote", out)
