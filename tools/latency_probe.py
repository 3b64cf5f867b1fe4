#!/usr/bin/env python3
"""Per-call latency of the trait-shaped entry points (micro-batches from host buffers): what one
get_or_create_placement / lookup costs when the host does NOT batch.  Prints JSON."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
g = rio_gp.GpuPlacement(n, m)
g.set_nodes(cfg["cap"], cfg["alive"]); g.set_objects(n, cfg["load"], cfg["aff"])
rng = np.random.default_rng(5)
out = {}
for k in (1, 16, 256, 1000, 4096):
    reps = 300
    batches = [(rng.integers(0, n, k).astype(np.uint32), rng.integers(0, m, k).astype(np.uint32)) for _ in range(reps)]
    g.place_pending(*batches[0])
    t0 = time.perf_counter()
    for ii, rq in batches:
        g.place_pending(ii, rq)
    dt = (time.perf_counter() - t0) / reps
    out["place_pending_%d" % k] = {"us_per_call": dt * 1e6, "requests_per_s": k / dt}
    t0 = time.perf_counter()
    for ii, rq in batches:
        g.lookup_batch(ii)
    dt = (time.perf_counter() - t0) / reps
    out["lookup_%d" % k] = {"us_per_call": dt * 1e6, "lookups_per_s": k / dt}
    if k >= 256:   # (single-entry update / remove: tools/latency_small_ops.py)
        t0 = time.perf_counter()
        for ii, rq in batches:
            g.update_batch(ii, rq)
        dt = (time.perf_counter() - t0) / reps
        out["update_%d" % k] = {"us_per_call": dt * 1e6, "updates_per_s": k / dt}
        t0 = time.perf_counter()
        for ii, rq in batches:
            g.remove_batch(ii)
        dt = (time.perf_counter() - t0) / reps
        out["remove_%d" % k] = {"us_per_call": dt * 1e6, "removes_per_s": k / dt}
print(json.dumps(out))
g.close()
