#!/usr/bin/env python3
"""Pipelined churn ticks (config 5): host enqueue time per tick against the whole time per tick, with and without the
liveness pushes in the loop (is the stream host-bound?)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
g = rio_gp.GpuPlacement(n, m)
g.set_nodes(cfg["cap"], cfg["alive"])
g.set_objects(n, cfg["load"], cfg["aff"])
g.set_assign(synth.warm_assign(n, m))
g.tick()
masks = [synth.churn_mask(m, 2 + k) for k in range(60)]
for k in range(10):
    g.set_alive_all(masks[k]); g.tick_async()
g.tick_wait(); g.sync()
t0 = time.perf_counter()
for k in range(10, 60):
    g.set_alive_all(masks[k]); g.tick_async()
t1 = time.perf_counter()
g.tick_wait()
t2 = time.perf_counter()
print(json.dumps({"host_enqueue_us_per_tick": (t1 - t0) / 50 * 1e6, "total_us_per_tick": (t2 - t0) / 50 * 1e6}))
# the same without liveness pushes in the loop (ticks over a fixed mask: fast path after the first)
for k in range(5):
    g.tick_async()
g.tick_wait(); g.sync()
t0 = time.perf_counter()
for k in range(50):
    g.tick_async()
t1 = time.perf_counter()
g.tick_wait()
t2 = time.perf_counter()
print(json.dumps({"fixed_mask_host_enqueue_us_per_tick": (t1 - t0) / 50 * 1e6, "fixed_mask_total_us_per_tick": (t2 - t0) / 50 * 1e6}))
g.close()
