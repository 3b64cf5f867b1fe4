#!/usr/bin/env python3
"""Where k_cut_fused spends its time on the config-5 churn stream: per-workgroup phase trace of the last tick's launch
(rio_gp_debug_cut_trace).  Usage: cut_trace.py [ticks]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
which = sys.argv[2] if len(sys.argv) > 2 else "churn"
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
g = rio_gp.GpuPlacement(n, m)
if which == "churn":
    g.set_nodes(cfg["cap"], cfg["alive"])
    g.set_objects(n, cfg["load"], cfg["aff"])
    g.set_assign(synth.warm_assign(n, m))
    g.tick()
else:  # contended cold solve (capacity 0.9 x load): every node cut, all in a few workgroups
    g.set_nodes((cfg["cap"].astype(np.float64) * 0.72).astype(np.uint64), cfg["alive"])
    g.set_objects(n, cfg["load"], cfg["aff"])
g.cut_trace(True)
for k in range(reps):
    if which == "churn":
        g.set_alive_all(synth.churn_mask(m, 2 + k))
        st = g.tick()
    else:
        st = g.solve()
    tr = g.cut_trace(True, read=True).astype(np.int64)
    t0 = tr[:, 0].min()
    us = lambda x: x / 100.0
    dur = us(tr[:, 1] - tr[:, 0]) , us(tr[:, 2]), us(tr[:, 3]), us(tr[:, 4])
    worst = np.argsort(-(dur[0] + dur[1] + dur[2] + dur[3]))[:4]
    print("tick %d cut_nodes %d: max nloc %d, start skew %.1f us" % (k, st["cut_nodes"], tr[:, 5].max(), us(tr[:, 0].max() - t0)))
    for b in worst:
        print("   wg %3d nloc %4d S %3d | P0 %.1f P1 %.1f P2 %.1f (P2a %.1f) P3 %.1f us | started +%.1f" % (
            b, tr[b, 5], tr[b, 6], dur[0][b], dur[1][b], dur[2][b], us(tr[b, 7]), dur[3][b], us(tr[b, 0] - t0)))
    print("   median wg: P0 %.1f P1 %.1f P2 %.1f P3 %.1f" % tuple(float(np.median(d)) for d in dur))
    tot = dur[0] + dur[1] + dur[2] + dur[3] + (us(tr[:, 7]) if os.environ.get("RIO_TRACE_ADD_WALKS") else 0)
    print("   SUMMARY tick %d: worst workgroup %.1f us (wg %d, nloc %d, S %d)" % (k, tot.max(), tot.argmax(), tr[tot.argmax(), 5], tr[tot.argmax(), 6]))
g.close()
