#!/usr/bin/env python3
"""Per-call latency of the single-object write paths (update / remove / clean_server) at the dense and string layers."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
cfg = synth.config("c3")
g = rio_gp.GpuPlacement(cfg["n"], cfg["m"]); g.set_nodes(cfg["cap"], cfg["alive"]); g.set_objects(cfg["n"], cfg["load"], cfg["aff"])
g.set_assign(synth.warm_assign(cfg["n"], cfg["m"]))
res = {}
def t(name, fn, reps=300):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    res[name] = (time.perf_counter() - t0) / reps * 1e6
i1 = np.array([12345], np.uint32); n1 = np.array([7], np.uint32)
i16 = np.arange(100, 116, dtype=np.uint32); n16 = (i16 % 9).astype(np.uint32)
t("dense update_batch(1)", lambda: g.update_batch(i1, n1))
t("dense update_batch(16)", lambda: g.update_batch(i16, n16))
t("dense remove_batch(1)", lambda: g.remove_batch(i1))
t("dense lookup_batch(1)", lambda: g.lookup_batch(i1))
p = rio_gp.GpuObjectPlacement(max_objects=1 << 16, max_nodes=64)
p.set_member("10.0.0.1:5000", True)
p.update("Obj", "1", "10.0.0.1:5000")
t("trait update", lambda: p.update("Obj", "1", "10.0.0.1:5000"))
t("trait lookup", lambda: p.lookup("Obj", "1"))
t("trait remove+update", lambda: (p.remove("Obj", "1"), p.update("Obj", "1", "10.0.0.1:5000")))
t("trait clean_server (empty server)", lambda: p.clean_server("10.0.0.9:5000"))
print(json.dumps({k: round(v, 1) for k, v in res.items()}))
