#!/usr/bin/env python3
"""Quiet committed ticks of config 3 (nothing changes between ticks: k_scan + k_resolve per tick, no fix-up) with k_resolve on a
stream of its own beside the next tick's scan, the scans themselves CHAINED over two streams (workgroup b of tick k + 1 waits
for workgroup b of tick k, not for its launch: the product's way), the same without the chain (lab knob: chain=False) and
everything on the main stream (overlap=False), alternating in ONE run; final table and `used` compared.  Usage: quiet_overlap_ab.py [ticks=200] [config=c3|c4|c2] [rows]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = synth.config(sys.argv[2] if len(sys.argv) > 2 else "c3", **({"n_override": int(sys.argv[3])} if len(sys.argv) > 3 else {}))
n, m = cfg["n"], cfg["m"]
out = {"n": n, "m": m, "ticks": ticks, "runs": []}
final = {}
for name, ov, chn in (("chained", True, True), ("overlap", True, False), ("main stream", False, False),
                      ("chained#2", True, True), ("overlap#2", True, False), ("main stream#2", False, False)) + (
                      tuple(("chained, hand-over per workgroup#%d" % k if k % 2 else "chained#%d" % (3 + k), True, True) for k in range(8)) if os.environ.get("CHAIN_PER_WAVE_AB") else ()) + (
                      tuple(("chained, k_resolve in line#%d" % k if k % 2 else "chained#%d" % (13 + k), True, True) for k in range(6)) if os.environ.get("CHAIN_INLINE_AB") else ()) + (
                      tuple(("diag%d: %s" % (d, {1: "chained kernel, one stream, no waits", 2: "two streams, no waits"}[d]), True, True) for d in (1, 2))
                      if os.environ.get("CHAIN_DIAGS") else ()):
    os.environ["RIO_GP_CHAIN_PER_WAVE"] = "0" if "per workgroup" in name else "1"
    os.environ["RIO_GP_CHAIN_INLINE_BELOW"] = str(1 << 40) if "in line" in name else "0"
    if name.startswith("diag"):
        os.environ["RIO_GP_CHAIN_DIAG"] = name[4]
    g = rio_gp.LabPlacement(n, m)
    g.set_compact("auto", overlap=ov, chain=chn)
    g.set_nodes(cfg["cap"], cfg["alive"])
    g.set_objects(n, cfg["load"], cfg["aff"])
    g.set_assign(cfg["cur"])
    for _ in range(20):
        g.tick_async()
    g.tick_wait(); g.sync()
    t0 = time.perf_counter()
    for _ in range(ticks):
        g.tick_async()
    t1 = time.perf_counter()
    sts = g.tick_wait()
    t2 = time.perf_counter()
    us = (t2 - t0) / ticks * 1e6
    out["runs"].append({"k_resolve": name, "us_per_tick": us, "host_enqueue_us_per_tick": (t1 - t0) / ticks * 1e6,
                        "frac_of_8TBps": 16 * n / (us * 1e-6) / 8e12, "slow_path_ticks": sum(x["slow_path"] for x in sts)})
    final[name] = (g.get_assign(), g.get_nodes()[2], sts[-1])
    g.close()
a0 = final["main stream"]
out["equal"] = all(np.array_equal(a0[0], v[0]) and np.array_equal(a0[1], v[1]) and a0[2] == v[2] for v in final.values())
print(json.dumps(out, indent=1))
