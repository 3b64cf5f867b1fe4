#!/usr/bin/env python3
"""A/B sweep of k_scan variants on one GPU, one process, interleaved rounds (methodology rule 24)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth

workload = sys.argv[1] if len(sys.argv) > 1 else "c3"
variants = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4").split(",")]
cfg = synth.config(workload)
n, m = cfg["n"], cfg["m"]
hs = {}
for tpi in variants:
    os.environ["RIO_GP_SCAN_TPI"] = str(tpi)
    g = rio_gp.GpuPlacement(n, m)
    g.set_nodes(cfg["cap"], cfg["alive"])
    g.set_objects(n, cfg["load"], cfg["aff"])
    if workload == "c3w":
        g.set_assign(cfg["cur"])
    hs[tpi] = g
res = {t: {"scan": [], "res": [], "step": []} for t in variants}
set_tpi = rio_gp.lib().rio_gp_debug_set_scan_tpi
for rnd in range(6):
    for tpi in variants:
        g = hs[tpi]
        set_tpi(tpi)
        for _ in range(20):
            a, b = g.solve_profiled()
            res[tpi]["scan"].append(a); res[tpi]["res"].append(b)
        g.sync()
        t0 = time.perf_counter()
        for _ in range(100):
            g.solve_async()
        g.solve_wait()
        res[tpi]["step"].append((time.perf_counter() - t0) / 100 * 1e3)
for tpi in variants:
    r = res[tpi]
    sc = np.array(r["scan"][20:]); rs = np.array(r["res"][20:]); st = np.array(r["step"][1:])
    print("TPI=%d scan_ms median %.4f min %.4f -> %.0f GB/s (%.1f%% of 8TB/s) | resolve_ms median %.4f | step_ms median %.4f min %.4f -> %.3e dec/s" % (
        tpi, np.median(sc), sc.min(), 16 * n / np.median(sc) / 1e6, 16 * n / np.median(sc) / 1e6 / 80.0,
        np.median(rs), np.median(st), st.min(), n / (np.median(st) * 1e-3)))
g = hs[variants[0]]
for mode, name, nbytes in ((0, "grid-stride 2048x256 3r+1w", 16), (5, "grid-stride 8192x256 3r+1w", 16),
                           (6, "grid-stride 1024x256 3r+1w", 16), (1, "block-tiled 256x1024 3r+1w", 16),
                           (2, "wave-contiguous 256x1024 3r+1w", 16), (3, "read-only 3 cols", 12), (4, "copy 1r+1w", 8)):
    ms = g.stream_probe(mode, 30)
    print("probe %-34s %.4f ms -> %.0f GB/s" % (name, ms, nbytes * n / ms / 1e6))
