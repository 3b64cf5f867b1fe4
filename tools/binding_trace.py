#!/usr/bin/env python3
"""Phase traces (lab build, wall_clock64 stamps of every workgroup) of the whole-table fix-up on the two tables on which capacity
binds.  table 6 k_cut_apply: start | node tables | rows streamed, packed | cut waves located | undecided rows settled | - | - | end;
table 1 / 2 k_fill rounds (see fill_trace.py), table 3 k_scan.   Usage: binding_trace.py [contended|skew] [solves]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
which = sys.argv[1] if len(sys.argv) > 1 else "contended"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
g = rio_gp.LabPlacement(n, m)
cap, aff = (synth.contended_cap(cfg), cfg["aff"]) if which == "contended" else (cfg["cap"], synth.skew_affinity(n, m))
g.set_nodes(cap, cfg["alive"])
g.set_objects(n, cfg["load"], aff)
if len(sys.argv) > 3:
    g.set_compact("auto", cut_pack=sys.argv[3])
for _ in range(4):
    st = g.solve()
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 1   # 1 traces | +2 no pack stores | +4 no `next` stores (timing experiments)
rio_gp.lab_lib().rio_gp_debug_ktrace(g.handle, flags, 0, None)
for _ in range(reps):
    st = g.solve()
out = {"which": which, "last": st}
names = {3: "k_scan", 6: "k_cut_apply", 0: "k_cut_settle", 1: "k_fill round 0 (apply)", 2: "k_fill rounds (fill only)", 4: "node order inside the last k_fill"}
tabs = {t: g.ktrace(True, t).astype(np.int64) for t in names}
g.ktrace(False)
g.close()
base = min(int(t[t[:, 0] > 0][:, 0].min()) for t in tabs.values() if (t[:, 0] > 0).any())
for t, tr in tabs.items():
    rows = tr[tr[:, 0] > 0]
    if not len(rows):
        continue
    t0 = rows[:, 0].min()
    rec = {"workgroups": int(len(rows)), "first_start_us": float((t0 - base) / 100.0),
           "first_start_to_last_end_us": float((rows[:, 7].max() - t0) / 100.0),
           "start_skew_us": float((rows[:, 0].max() - t0) / 100.0)}
    ph = {}
    prev = rows[:, 0]
    for c in range(1, 8):
        col = rows[:, c]
        ok = col > 0
        if not ok.any():
            continue
        d = (col[ok] - prev[ok]) / 100.0
        ph["->%d" % c] = {"median": float(np.median(d)), "max": float(d.max()), "argmax_wg": int(np.flatnonzero(ok)[d.argmax()])}
        prev = np.where(ok, col, prev)
    rec["phases_us"] = ph
    out[names[t]] = rec
print(json.dumps(out, indent=1))
