#!/usr/bin/env python3
"""Where the time of a place_pending call of up to 4 096 requests from host buffers goes: per-call wall time (sticky hits /
first touches) and the one-workgroup kernel's own phase stamps (lab build, wall_clock64): start | requests + rows + node
operands here | first request per object elected (LDS table) | claim totals checked | decisions published (LDS) | results and
table stores issued | completion word stored.  Usage: pp_one_trace.py [requests=4096] [calls=200]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
out = {"requests": k, "calls": calls}
rng = np.random.default_rng(3)
for what in ("sticky", "first_touch"):
    g = rio_gp.LabPlacement(n, m)
    g.set_nodes(cfg["cap"], cfg["alive"])
    g.set_objects(n, cfg["load"], cfg["aff"])
    if what == "sticky":
        g.set_assign(synth.warm_assign(n, m))
        g.tick()
    batches = [(rng.choice(n, k, replace=False).astype(np.uint32), rng.integers(0, m, k).astype(np.uint32)) for _ in range(calls + 8)]
    for b in batches[:4]:
        g.place_pending(*b)
    t0 = time.perf_counter()
    for b in batches[4:4 + calls]:
        g.place_pending(*b)
    dt = (time.perf_counter() - t0) / calls
    rec = {"us_per_call": dt * 1e6, "requests_per_s": k / dt}
    g.ktrace(True)
    spans = []
    for b in batches[4 + calls:]:
        node, flag = g.place_pending(*b)
        tr = g.ktrace(True, 6).astype(np.int64)[0]
        spans.append([(int(x) - int(tr[0])) / 100.0 if x else None for x in tr])
    g.ktrace(False)
    rec["kernel_phase_us"] = spans[-1]
    rec["flags_last_call"] = {int(f): int(c) for f, c in zip(*np.unique(flag, return_counts=True))}
    out[what] = rec
    g.close()
print(json.dumps(out))
