#!/usr/bin/env python3
"""Where do the 25 us of k_pp_small at 256 requests go?  Three request mixes under rocprofv3 --kernel-trace: sticky hits only
(no claim), first touches spread over all requesters, first touches on ONE requester."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
g = rio_gp.GpuPlacement(n, m)
g.set_nodes(np.full(m, 1 << 50, np.uint64), cfg["alive"]); g.set_objects(n, cfg["load"], cfg["aff"])
g.set_assign(synth.warm_assign(n, m)[: n])   # every object placed: sticky hits
rng = np.random.default_rng(1)
which = sys.argv[1]
if which != "sticky":
    g.set_assign(np.full(n, 0xFFFFFFFF, np.uint32))
for _ in range(50):
    idx = rng.integers(0, n, 256).astype(np.uint32)
    req = rng.integers(0, m, 256).astype(np.uint32) if which != "one" else np.full(256, 7, np.uint32)
    g.place_pending(idx, req)
g.close()
