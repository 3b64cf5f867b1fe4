#!/usr/bin/env python3
"""k_scan beyond the Infinity Cache: the headline table (10 M rows, 160 MB of columns) fits the 256 MiB MALL, so this
sweeps the SAME kernel over tables of 10 M .. 100 M rows (up to 1.6 GB of columns: config 4's whole table on one GPU)
and prints per-launch time and algorithmic GB/s (16 B/row) next to the plain streaming probe at the same size.
Usage: scale_probe.py [rows,rows,...] [m,m,...]"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth

rows = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10000000,30000000,100000000").split(",")]
ms_ = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1024,4096").split(",")]
nmax = max(rows)
t0 = time.time()
load = synth.zipf_loads(nmax)
out = []
for m in ms_:
    aff = synth.affinity(nmax, m)
    for n in rows:
        g = rio_gp.LabPlacement(n, m)
        l, a = np.ascontiguousarray(load[:n]), np.ascontiguousarray(aff[:n])
        g.set_nodes(synth.uniform_cap(l, m), np.ones(m, np.uint8))
        g.set_objects(n, l, a)
        for _ in range(5):
            g.solve_profiled()
        sc, rs = [], []
        for _ in range(30):
            s, r = g.solve_profiled()
            sc.append(s); rs.append(r)
        g.sync(); t1 = time.perf_counter()
        for _ in range(50):
            g.solve_async()
        st, n_slow = g.solve_wait()
        step = (time.perf_counter() - t1) / 50 * 1e3
        probe = g.stream_probe(0, 10)
        rec = {"rows": n, "nodes": m, "column_bytes": 16 * n, "scan_ms": float(np.median(sc)), "resolve_ms": float(np.median(rs)),
               "scan_GBps": 16 * n / np.median(sc) / 1e6, "frac_of_8TBps": 16 * n / np.median(sc) / 1e6 / 8000.0,
               "step_ms": step, "decisions_per_s": n / (step * 1e-3), "probe_ms": probe, "probe_GBps": 16 * n / probe / 1e6,
               "slow_steps": n_slow}
        out.append(rec)
        print(json.dumps(rec), flush=True)
        g.close()
print(json.dumps({"scale_probe": out, "wall_s": time.time() - t0}))
