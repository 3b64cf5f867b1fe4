#!/usr/bin/env python3
"""place_pending_dev, 1 M / 10 M requests, rows per window 2^12 | 2^13 | 2^14 (lab build knob), same run."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
from hipbuf import DevBuf
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
perm = (synth.r(np.arange(n, dtype=np.uint64), 9) % np.uint64(n)).astype(np.uint32)
reqp = cfg["aff"][perm]
none = np.full(n, 0xFFFFFFFF, np.uint32)
out = {}
for shift in (14, 13, 12, 14, 13):
    rio_gp.lab_lib().rio_gp_debug_set_part_shift(shift)
    g = rio_gp.LabPlacement(n, m)
    g.set_nodes(cfg["cap"], cfg["alive"])
    g.set_objects(n, cfg["load"], cfg["aff"])
    d_idx, d_req, d_node, d_flag = DevBuf(perm), DevBuf(reqp), DevBuf(nbytes=4 * n), DevBuf(nbytes=4 * n)
    rec = {}
    for k in (1_000_000, 10_000_000):
        ts = []
        for rep in range(5):
            g.set_assign(none); g.get_nodes(); g.sync()
            g.timer_begin()
            g.place_pending_dev(k, d_idx.ptr, d_req.ptr, d_node.ptr, d_flag.ptr)
            ts.append(g.timer_end() * 1e-3)
        rec[str(k)] = round(float(np.mean(ts[1:])) * 1e6, 1)
    out.setdefault("shift_%d" % shift, []).append(rec)
    g.close()
rio_gp.lab_lib().rio_gp_debug_set_part_shift(14)
print(json.dumps(out))
