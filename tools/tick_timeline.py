#!/usr/bin/env python3
"""Per-tick timeline from a rocprofv3 kernel trace of the churn stream: for the last ticks, every kernel's start offset
(from the tick's first kernel), duration and the idle gap before it.  A tick starts at each k_scan / k_inc_scan launch."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ticks, cur = [], None
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").replace("riogp::", "").split("(")[0]
    if name.startswith("k_scan") or name.startswith("k_inc_scan"):
        cur = []
        ticks.append(cur)
    if cur is not None:
        cur.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
show = ticks[-3:]
for t in show:
    t0 = t[0][1]
    prev_end = t0
    print("tick: %d kernels, first start -> last end %.1f us, sum of kernel time %.1f us" % (
        len(t), (t[-1][2] - t0) / 1e3, sum(e - s for _, s, e in t) / 1e3))
    for name, s, e in t:
        print("  +%7.1f us  %-28s %6.1f us   gap %5.1f" % ((s - t0) / 1e3, name[:28], (e - s) / 1e3, (s - prev_end) / 1e3))
        prev_end = e
if len(ticks) > 2:
    starts = [t[0][1] for t in ticks]
    d = [(b - a) / 1e3 for a, b in zip(starts[1:-1], starts[2:])]
    print("tick-to-tick period (us): median %.1f min %.1f max %.1f over %d" % (sorted(d)[len(d) // 2], min(d), max(d), len(d)))
