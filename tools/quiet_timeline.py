#!/usr/bin/env python3
"""Out of a rocprofv3 kernel trace of tools/quiet_overlap_ab.py: k_scan durations, scan-to-scan periods and where k_resolve runs.
Usage: quiet_timeline.py <kernel_trace.csv>"""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
scans = [(s, e) for s, e, k in ev if "k_scan" in k]
res = [(s, e) for s, e, k in ev if "k_resolve" in k]
# split into runs: a gap of more than 1 ms between scans starts a new run
runs, cur = [], [scans[0]]
for a, b in zip(scans, scans[1:]):
    if b[0] - a[1] > 1_000_000: runs.append(cur); cur = []
    cur.append(b)
runs.append(cur)
for r in runs:
    if len(r) < 50: continue
    d = np.array([e - s for s, e in r]) / 1e3
    per = np.diff(np.array([s for s, e in r])) / 1e3
    gap = np.array([b[0] - a[1] for a, b in zip(r, r[1:])]) / 1e3
    lo, hi = r[0][0], r[-1][1]
    rr = [(s, e) for s, e in res if lo <= s <= hi]
    inside = sum(1 for s, e in rr if any(a <= s < b for a, b in r))
    print("scans %4d  duration median %.1f us  period median %.1f us  gap median %.1f us (p90 %.1f)  k_resolve median %.1f us, %d of %d started inside a scan"
          % (len(r), np.median(d), np.median(per), np.median(gap), np.percentile(gap, 90), np.median([(e - s) / 1e3 for s, e in rr]), inside, len(rr)))
