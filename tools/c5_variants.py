#!/usr/bin/env python3
"""Config-5 churn tick (10 M x 1 024, 10 % of the nodes flip per tick) under the scan variants of the lab build:
  never   k_scan<COMPACT> streams cur/load/aff and rewrites the column (round 3's tick)
  auto    k_inc_scan (the assignment column alone, in place) + k_rebal (the pending rows dealt out evenly to the fix-up's
          workgroups): the product's choice
Per variant: us per tick pipelined (rio_gp_tick_async) and synchronous (rio_gp_tick), in the stream's steady state, the
kernels' own spans (wall_clock64 phase traces), and the final table compared across the variants.
Usage: c5_variants.py [ticks=100] [warm=60] [config=c3|c4]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import rio_gp  # noqa: E402
import synth  # noqa: E402

if os.environ.get("RIO_NT"):   # non-temporal column streams of the scans: 0 by table size | 1 always | 2 never (A/B runs)
    rio_gp.lab_lib().rio_gp_debug_set_scan_nt(int(os.environ["RIO_NT"]))
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 100
warm_ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cfg = synth.config(sys.argv[3] if len(sys.argv) > 3 else "c3")
n, m = cfg["n"], cfg["m"]
warm = synth.warm_assign(n, m)
masks = [synth.churn_mask(m, 2 + k) for k in range(warm_ticks + 2 * ticks + 8)]
out = {"n": n, "m": m, "ticks": ticks, "warm_ticks": warm_ticks}
final = {}
for name in (sys.argv[4].split(",") if len(sys.argv) > 4 else ("never", "auto")):
    inc = name.partition("#")[0]   # ("<inc>#2": the same variant again, to see run-order effects)
    g = rio_gp.LabPlacement(n, m)
    g.set_compact("auto", inc=inc)
    g.set_nodes(cfg["cap"], cfg["alive"])
    g.set_objects(n, cfg["load"], cfg["aff"])
    g.set_assign(warm)
    g.tick()
    k = 0
    for _ in range(warm_ticks):
        g.set_alive_all(masks[k]); k += 1
        g.tick_async()
    g.tick_wait()
    g.sync()
    if os.environ.get("RIO_INC_DEBUG"):   # timing experiments (results wrong): 2 no packing | 4 no in-place stores | 6 both
        import ctypes as C
        rio_gp.lab_lib().rio_gp_debug_ktrace(g.handle, int(os.environ["RIO_INC_DEBUG"]), 0, None)
    t0 = time.perf_counter()
    for _ in range(ticks):
        g.set_alive_all(masks[k]); k += 1
        g.tick_async()
    try:
        sts = g.tick_wait()
    except Exception as e:
        sts = [{"error": repr(e)}]
    dtp = time.perf_counter() - t0
    if os.environ.get("RIO_INC_DEBUG"):
        tr = g.ktrace(True, 3).astype(np.int64)
        rows = tr[tr[:, 0] > 0]
        out[name] = {"pipelined_us": dtp / ticks * 1e6, "debug": os.environ["RIO_INC_DEBUG"],
                    "scan_wg_median_us": float(np.median((rows[:, 7] - rows[:, 0]) / 100.0)),
                    "scan_span_us": float((rows[:, 7].max() - rows[:, 0].min()) / 100.0),
                    "scan_loop_end_median_us": float(np.median((rows[:, 2] - rows[:, 0]) / 100.0))}
        print(json.dumps(out), flush=True); os._exit(0)
    t0 = time.perf_counter()
    for _ in range(ticks):
        g.set_alive_all(masks[k]); k += 1
        st = g.tick()
    dts = time.perf_counter() - t0
    rec = {"pipelined_us": dtp / ticks * 1e6, "synchronous_us": dts / ticks * 1e6, "last": st}
    try:
        g.ktrace(True)
        for _ in range(6):
            g.set_alive_all(masks[k]); k += 1
            g.tick()
        names = {3: "scan", 5: "rebal", 0: "resolve", 1: "fill0", 2: "fill1"}
        tabs = {t: g.ktrace(True, t).astype(np.int64) for t in names}
        g.ktrace(False)
        base = min(int(t[t[:, 0] > 0][:, 0].min()) for t in tabs.values() if (t[:, 0] > 0).any())
        spans = {}
        for t, tr in tabs.items():
            rows = tr[tr[:, 0] > 0]
            if not len(rows):
                continue
            dur = (rows[:, 7] - rows[:, 0]) / 100.0
            rel = (rows - rows[:, :1]) / 100.0
            rel[rows == 0] = np.nan
            spans[names[t]] = {"start": round((int(rows[:, 0].min()) - base) / 100.0, 1), "end": round((int(rows[:, 7].max()) - base) / 100.0, 1),
                               "wg_median_us": round(float(np.median(dur)), 1), "wg_max_us": round(float(dur.max()), 1),
                               "phase_median_us": [None if np.isnan(x) else round(float(x), 1) for x in np.nanmedian(rel, axis=0)],
                               "phase_max_us": [None if np.isnan(x) else round(float(x), 1) for x in np.nanmax(rel, axis=0)]}
        rec["spans_us"] = spans
    except Exception as e:  # measurement aid
        rec["spans_us"] = {"error": repr(e)}
    final[name] = (g.get_assign(), g.get_nodes()[2])
    g.close()
    out[name] = rec
keys = list(final)
out["tables_equal_across_variants"] = all(np.array_equal(final[keys[0]][0], final[q][0]) and np.array_equal(final[keys[0]][1], final[q][1])
                                          for q in keys[1:])
print(json.dumps(out))
