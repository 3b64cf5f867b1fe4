#!/usr/bin/env python3
"""clean_server(s) on the warm 10 M x 1 024 table: wall clock of the synchronous ABI call (one node / 10 % of the nodes)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
# RIO_GP_CLEAN_FULL_FROM (lab build): lanes of a wave that must evict before the wave writes its whole kilobyte back (16; 64 = never)
g = rio_gp.LabPlacement(n, m) if os.environ.get("RIO_GP_CLEAN_FULL_FROM") else rio_gp.GpuPlacement(n, m)
g.set_nodes(cfg["cap"], cfg["alive"])
g.set_objects(n, cfg["load"], cfg["aff"])
warm = synth.warm_assign(n, m)
dead = list(np.flatnonzero(synth.churn_mask(m, 1) == 0))
out = {}
for name, fn in (("one_node", lambda: g.clean_server(3)), ("ten_percent", lambda: g.clean_servers(dead))):
    ts = []
    for k in range(12):
        g.set_assign(warm); g.get_nodes()
        t0 = time.perf_counter(); ev = fn(); t = time.perf_counter() - t0
        if k > 1: ts.append(t)
    out[name] = {"call_us": float(np.mean(ts)) * 1e6, "call_us_min": float(np.min(ts)) * 1e6, "evicted": int(ev),
                 "bytes": 4 * n + 4 * int(ev), "frac_of_8TBps": (4 * n + 4 * int(ev)) / float(np.mean(ts)) / 8e12}
print(json.dumps(out))
g.close()
