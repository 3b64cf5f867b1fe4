#!/usr/bin/env python3
"""A/B of the big random CRUD batches on config 3's table (10 M rows x 1 024 nodes): window-partitioned kernels at
windows of 4 096 / 8 192 / 16 384 rows (chunks of 8 192 entries, round 5's form, and of 16 384, round 6's) against the plain
per-entry kernels; 10 M random entries with duplicates.
HIP events on the library stream around the whole call (kernels + the stats read-back).  Usage: crud_ab.py [reps] [variant,variant...]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
from hipbuf import DevBuf
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
L, vp = rio_gp.lab_lib(), C.c_void_p
idx = DevBuf((synth.r(np.arange(n, dtype=np.uint64), 7) % np.uint64(n)).astype(np.uint32))
node = DevBuf(synth.warm_assign(n, m, stream=8))
warm = synth.warm_assign(n, m)
out = {}
# (bit 7 of the shift word: 16 384-entry chunks wherever they fit; else the product's rule — 8 192-entry chunks for CRUD batches)
for name, shift, part in (("plain", 14, False), ("part_w4096", 12, True), ("part_w8192", 13, True),
                          ("part_w16384", 14, True), ("part_w8192_bigchunks", 13 | 0x80, True), ("part_w16384_bigchunks", 14 | 0x80, True),
                          ("part_w16384#2", 14, True), ("part_w16384_bigchunks#2", 14 | 0x80, True)):
    if len(sys.argv) > 2 and name not in sys.argv[2].split(","):   # (one variant: per-kernel profiles)
        continue
    L.rio_gp_debug_set_part_shift(shift)
    g = rio_gp.LabPlacement(n, m)
    g.set_compact("auto", partitioned_crud=part)
    g.set_nodes(cfg["cap"], cfg["alive"])
    g.set_objects(n, cfg["load"], cfg["aff"])
    g.set_assign(warm)
    h = g.handle
    tu, tr = [], []
    for r in range(reps):
        g.sync(); g.timer_begin()
        L.rio_gp_update_batch_dev(h, n, vp(idx.ptr), vp(node.ptr))
        tu.append(g.timer_end())
        g.get_nodes()
        g.sync(); g.timer_begin()
        L.rio_gp_remove_batch_dev(h, n, vp(idx.ptr))
        tr.append(g.timer_end())
        g.set_assign(warm)
    out[name] = {"update_ms": float(np.median(tu[1:])), "update_GBps": 8 * n / float(np.median(tu[1:])) / 1e6,
                 "remove_ms": float(np.median(tr[1:])), "remove_GBps": 8 * n / float(np.median(tr[1:])) / 1e6}
    print(name, json.dumps(out[name]), flush=True)
    g.close()
L.rio_gp_debug_set_part_shift(14)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "crud_ab.json"), "w"), indent=1)
