#!/usr/bin/env python3
"""Per-call kernel timeline of place_pending_dev out of a rocprofv3 kernel trace of tools/pp_probe.py:
prints, for the LAST call of each batch size, every kernel with its start offset and duration (us)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].split("(")[0].replace("void ", "").replace("riogp::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# a call starts at k_part_bin; list until the output kernel
calls, cur = [], None
for k in ks:
    if k[0].startswith("k_part_bin"):
        cur = [k]; calls.append(cur)
    elif cur is not None:
        cur.append(k)
        if k[0].startswith("k_pp_win_output"): cur = None
for c in calls:
    c[:] = [k for k in c]
# pp_probe.py: calls 0-3 = 1 M requests (first touch), 4-7 = 10 M (first touch), 8.. = 10 M (8: first touch, then sticky)
for name, i in (("1 M first touch", 3), ("10 M first touch", 7), ("10 M sticky", len(calls) - 1)):
    c = calls[i]
    t0 = c[0][1]
    print("%s: %.1f us first start -> last end" % (name, (c[-1][2] - t0) / 1e3))
    for k in c: print("  %-46s +%8.1f  %8.1f" % (k[0][:46], (k[1] - t0) / 1e3, (k[2] - k[1]) / 1e3))
