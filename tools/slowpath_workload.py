#!/usr/bin/env python3
"""Workload for profiling the fix-up path (cut + water-fill): config-5 churn ticks, the contended cold solve
and the skewed cold solve.  Usage: slowpath_workload.py [churn|contended|skew] [reps] [auto|always|never (packed)] [default|nospec|spec]
  default = speculative enqueue after a solve that needed the fix-up (what the library does); nospec / spec = never / always"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
which = sys.argv[1] if len(sys.argv) > 1 else "churn"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
g = rio_gp.LabPlacement(n, m)
if len(sys.argv) > 3:
    g.set_compact(sys.argv[3])   # auto | always | never
fx = sys.argv[4] if len(sys.argv) > 4 else "default"
if fx != "default":
    g.set_speculate({"nospec": "never", "spec": "always"}[fx])
res = {"which": which, "reps": reps, "n": n, "m": m, "compact": sys.argv[3] if len(sys.argv) > 3 else "auto", "fixup": fx}
if which == "churn":
    g.set_nodes(cfg["cap"], cfg["alive"])
    g.set_objects(n, cfg["load"], cfg["aff"])
    g.set_assign(synth.warm_assign(n, m))
    g.tick()
    masks = [synth.churn_mask(m, 2 + k) for k in range(reps)]
    g.sync(); moved = 0; t0 = time.perf_counter()
    for k in range(reps):
        g.set_alive_all(masks[k])
        st = g.tick()
        moved += st["claimed"] + st["spilled"]
    dt = time.perf_counter() - t0
    res.update(ms_per_tick=dt / reps * 1e3, rows_per_s=n * reps / dt, moved_per_s=moved / dt, last=st)
else:
    if which == "contended":
        cap = (cfg["cap"].astype(np.float64) * 0.72).astype(np.uint64)
        aff = cfg["aff"]
    else:
        cap = cfg["cap"]
        aff = np.minimum((np.random.default_rng(1).pareto(1.1, n)).astype(np.int64), m - 1).astype(np.uint32)
    g.set_nodes(cap, cfg["alive"])
    g.set_objects(n, cfg["load"], aff)
    st = g.solve(); g.sync(); t0 = time.perf_counter()
    for _ in range(reps):
        st = g.solve()
    dt = time.perf_counter() - t0
    res.update(ms_per_solve=dt / reps * 1e3, rows_per_s=n * reps / dt, last=st)
print(json.dumps(res))
g.close()
