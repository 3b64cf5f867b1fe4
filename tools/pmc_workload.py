#!/usr/bin/env python3
"""Workload for the PMC passes: 30 fast-path solves of config 3 + the stream probes (known byte counts,
used to calibrate FETCH_SIZE / WRITE_SIZE as MI355X_MICROARCH.md §HBM prescribes).  No torch: the process is a few
hundred milliseconds of set-up around the kernels being counted.  Usage: pmc_workload.py [config] [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import rio_gp, synth
cfg = synth.config(sys.argv[1] if len(sys.argv) > 1 else "c3", n_override=int(sys.argv[2]) if len(sys.argv) > 2 else None)
g = rio_gp.LabPlacement(cfg["n"], cfg["m"])
g.set_nodes(cfg["cap"], cfg["alive"])
g.set_objects(cfg["n"], cfg["load"], cfg["aff"])
for _ in range(30):
    g.solve_async()
g.solve_wait()
for mode in (4, 0, 3):
    g.stream_probe(mode, 10)
g.close()
