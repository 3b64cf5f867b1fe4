#!/usr/bin/env python3
"""place_pending rates: device-resident batches of 1 M / 10 M requests over the cold 10 M x 1 024 table (HIP events on the
library's stream).  Smaller batches: tools/pp_sizes.py."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
from hipbuf import DevBuf
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
if os.environ.get("RIO_PART_SHIFT"):   # lab build: rio_gp_debug_set_part_shift word (78 = 14 | 0x40: 8 192-entry chunks only; 142 = 14 | 0x80: 16 384-entry chunks at every size)
    rio_gp.lab_lib().rio_gp_debug_set_part_shift(int(os.environ["RIO_PART_SHIFT"]))
    g = rio_gp.LabPlacement(n, m)
else:
    g = rio_gp.GpuPlacement(n, m)
g.set_nodes(cfg["cap"], cfg["alive"])
g.set_objects(n, cfg["load"], cfg["aff"])
perm = (synth.r(np.arange(n, dtype=np.uint64), 9) % np.uint64(n)).astype(np.uint32)
reqp = cfg["aff"][perm]
out = {}
d_idx, d_req, d_node, d_flag = DevBuf(perm), DevBuf(reqp), DevBuf(nbytes=4 * n), DevBuf(nbytes=4 * n)
none = np.full(n, 0xFFFFFFFF, np.uint32)
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1_000_000, 10_000_000]   # (one size: per-kernel profiles)
for k in sizes:
    ts = []
    for rep in range(4):
        g.set_assign(none); g.get_nodes(); g.sync()
        g.timer_begin()
        g.place_pending_dev(k, d_idx.ptr, d_req.ptr, d_node.ptr, d_flag.ptr)
        ts.append(g.timer_end() * 1e-3)
    t = float(np.mean(ts[1:]))
    out["dev_%d" % k] = {"us": t * 1e6, "req_per_s": k / t, "GBps_28B": 28 * k / t / 1e9, "frac": 28 * k / t / 1e9 / 8000}
if len(sys.argv) > 1:
    print(json.dumps(out)); g.close(); sys.exit(0)
# warm (sticky) requests: every object already placed
g.set_assign(none); g.get_nodes()
g.place_pending_dev(n, d_idx.ptr, d_req.ptr, d_node.ptr, d_flag.ptr)
ts = []
for rep in range(3):
    g.sync(); g.timer_begin()
    g.place_pending_dev(n, d_idx.ptr, d_req.ptr, d_node.ptr, d_flag.ptr)
    ts.append(g.timer_end() * 1e-3)
t = float(np.mean(ts[1:]))
out["dev_10M_sticky"] = {"us": t * 1e6, "req_per_s": n / t, "frac": 28 * n / t / 1e9 / 8000}
# (batches of 1 .. 262 143 requests, host buffers and device-resident, first touch and sticky over >= 50 warmed calls each:
#  tools/pp_sizes.py — one method for every size)
print(json.dumps(out))
g.close()
