#!/usr/bin/env python3
"""GPU idle analysis of a rocprofv3 --kernel-trace run: python tools/gaps.py DIR
Prints busy time, idle time and the largest idle gaps (with the kernels around them) between the first and the
last kernel of the LAST complete proof-sized window (delimited by k0_cpu_rows launches)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-50:]))
rows.sort()
k0 = [i for i, r in enumerate(rows) if "k0_cpu_rows" in r[2]]
a, b = k0[4], k0[5]            # one timed proof of bench.py (launch 0 = verified proof, 1-2 warm-up, 3-7 timed)
win = rows[a:b]
busy = sum(e - s for s, e, _ in win)
wall = win[-1][1] - win[0][0]
print("kernels %d  wall %.2f ms  busy %.2f ms  idle %.2f ms" % (len(win), wall / 1e6, busy / 1e6, (wall - busy) / 1e6))
gaps = sorted(((win[i + 1][0] - win[i][1], win[i][2], win[i + 1][2]) for i in range(len(win) - 1)), reverse=True)
hist = {}
for g, x, y in gaps:
    hist[(x, y)] = hist.get((x, y), 0) + max(g, 0)
for (x, y), g in sorted(hist.items(), key=lambda kv: -kv[1])[:14]:
    print("  %8.1f us idle in total between %s -> %s" % (g / 1e3, x, y))
print("largest single gaps (offset from the first kernel of the proof):")
single = sorted(((win[i + 1][0] - win[i][1], win[i][1] - win[0][0], win[i][2], win[i + 1][2]) for i in range(len(win) - 1)), reverse=True)
for g, at, x, y in single[:25]:
    print("  %7.1f us at %7.2f ms  %s -> %s" % (g / 1e3, at / 1e6, x, y))
