#!/usr/bin/env python3
"""Config 4 on one GPU (100 M objects x 4 096 nodes): committed churn ticks at full size, for a rocprofv3 kernel trace of the
fix-up kernels at that size.  Usage: c4_tick_probe.py [ticks=4]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = synth.config("c4")
n, m = cfg["n"], cfg["m"]
g = rio_gp.GpuPlacement(n, m)
g.set_nodes(cfg["cap"], cfg["alive"])
g.set_objects(n, cfg["load"], cfg["aff"])
g.set_assign(synth.warm_assign(n, m))
g.tick()
g.sync(); t0 = time.perf_counter()
for k in range(ticks):
    g.set_alive_all(synth.churn_mask(m, 2 + k))
    st = g.tick()
dt = time.perf_counter() - t0
print(json.dumps({"n": n, "m": m, "ticks": ticks, "ms_per_tick": dt / ticks * 1e3, "last": st}))
g.close()
