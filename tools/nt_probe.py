#!/usr/bin/env python3
"""Streaming-probe sweep inside and beyond the 256 MiB Infinity Cache: what the chip gives the solve's 3-read/1-write
mix, a read-only pass and a 1:1 copy, with and without non-temporal hints, next to k_scan with plain and with non-temporal
column streams (launch_scan switches to the latter at 20 Mi rows; rio_gp_debug_set_scan_nt forces either).
Usage: nt_probe.py [rows,rows,...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth

rows = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10000000,100000000").split(",")]
m = 1024
base = synth.config("c3")
out = []
for n in rows:
    reps = -(-n // base["n"])
    load = np.tile(base["load"], reps)[:n]
    aff = np.tile(base["aff"], reps)[:n]
    g = rio_gp.LabPlacement(n, m)
    g.set_nodes(synth.uniform_cap(load, m), np.ones(m, np.uint8))
    g.set_objects(n, load, aff)
    rec = {"rows": n}
    for code, name in ((2, "plain"), (1, "nt")):
        rio_gp.lab_lib().rio_gp_debug_set_scan_nt(code)
        for _ in range(5):
            g.solve_profiled()
        sc = [g.solve_profiled()[0] for _ in range(30)]
        rec["k_scan_%s_GBps" % name] = 16 * n / float(np.median(sc)) / 1e6
    rio_gp.lab_lib().rio_gp_debug_set_scan_nt(0)
    for mode, name, nb in ((0, "3r1w", 16), (5, "3r1w_8192wg", 16), (7, "3r1w_ntload", 16), (8, "3r1w_ntstore", 16),
                           (9, "3r1w_ntboth", 16), (3, "read3", 12), (4, "copy", 8), (2, "3r1w_wavecontig", 16)):
        ms = g.stream_probe(mode, 10)
        rec["probe_%s_GBps" % name] = nb * n / ms / 1e6
    out.append(rec)
    print(json.dumps(rec), flush=True)
    g.close()
