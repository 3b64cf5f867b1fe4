#!/usr/bin/env python3
"""Where k_pp_win_gather's time goes: wall_clock64 phase stamps of its first 256 workgroups (lab build, trace table 7) on a
device-resident batch of 10 M (default) requests over the cold 10 M x 1 024 table.  Usage: pp_gather_trace.py [requests] [shift word]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
from hipbuf import DevBuf
k = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
if len(sys.argv) > 2:
    rio_gp.lab_lib().rio_gp_debug_set_part_shift(int(sys.argv[2]))
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
g = rio_gp.LabPlacement(n, m)
g.set_nodes(cfg["cap"], cfg["alive"])
g.set_objects(n, cfg["load"], cfg["aff"])
perm = (synth.r(np.arange(n, dtype=np.uint64), 9) % np.uint64(n)).astype(np.uint32)
d_idx, d_req, d_node, d_flag = DevBuf(perm), DevBuf(cfg["aff"][perm]), DevBuf(nbytes=4 * n), DevBuf(nbytes=4 * n)
none = np.full(n, 0xFFFFFFFF, np.uint32)
names = ["init", "walk 1 (first LP records of every piece)", "walk 1 tail", "rows pass", "walk 2 (answers)", "walk 2 tail + epilogue"]
out = {"requests": k}
for rep in range(3):
    g.set_assign(none); g.get_nodes(); g.sync()
    g.ktrace(True)
    g.place_pending_dev(k, d_idx.ptr, d_req.ptr, d_node.ptr, d_flag.ptr)
    t = g.ktrace(False, 7).astype(np.int64)
t = t[t[:, 0] > 0]
slots = [0, 1, 2, 3, 4, 5, 7]
out["workgroups_traced"] = int(len(t))
out["phase_us_median"] = {names[i]: float(np.median(t[:, slots[i + 1]] - t[:, slots[i]])) / 100.0 for i in range(6)}
out["phase_us_p90"] = {names[i]: float(np.percentile(t[:, slots[i + 1]] - t[:, slots[i]], 90)) / 100.0 for i in range(6)}
out["workgroup_us_median"] = float(np.median(t[:, 7] - t[:, 0])) / 100.0
out["first_start_to_last_end_us"] = float(t[:, 7].max() - t[:, 0].min()) / 100.0
print(json.dumps(out, indent=1))
g.close()
