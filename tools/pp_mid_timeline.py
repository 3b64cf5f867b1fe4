#!/usr/bin/env python3
"""Kernel timeline of one place_pending_dev call of 16 384 first-touch requests (the plain request kernels: batches between
4 096 and the window-sorted form).  Run under rocprofv3 --kernel-trace; pass the trace csv to print the last call's launches.
Usage: pp_mid_timeline.py            (the workload)
       pp_mid_timeline.py trace.csv  (the analysis)"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
if len(sys.argv) > 1:
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
    # calls are separated by long gaps (set_assign + get_nodes + sync): take the last burst that contains a k_pp kernel
    bursts, cur = [], []
    for r in rows:
        if cur and r[0] - cur[-1][1] > 200000:
            bursts.append(cur); cur = []
        cur.append(r)
    bursts.append(cur)
    b = [x for x in bursts if any("k_pp" in r[2] for r in x)][-1]
    t0 = b[0][0]
    print("last call: %.1f us first start -> last end, %d launches" % ((b[-1][1] - t0) / 1e3, len(b)))
    for s, e, nme in b:
        nme = nme.split("(")[0].replace("void riogp::", "").replace("riogp::", "")
        print("  %-46s + %7.1f  %6.1f" % (nme[:46], (s - t0) / 1e3, (e - s) / 1e3))
    sys.exit(0)
import numpy as np
import rio_gp, synth
from hipbuf import DevBuf
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
g = rio_gp.GpuPlacement(n, m)
g.set_nodes(cfg["cap"], cfg["alive"])
g.set_objects(n, cfg["load"], cfg["aff"])
perm = (synth.r(np.arange(n, dtype=np.uint64), 9) % np.uint64(n)).astype(np.uint32)
reqp = cfg["aff"][perm]
d_idx, d_req, d_node, d_flag = DevBuf(perm), DevBuf(reqp), DevBuf(nbytes=4 * n), DevBuf(nbytes=4 * n)
none = np.full(n, 0xFFFFFFFF, np.uint32)
import time
for rep in range(4):
    g.set_assign(none); g.get_nodes(); g.sync(); time.sleep(0.01)
    g.place_pending_dev(16384, d_idx.ptr, d_req.ptr, d_node.ptr, d_flag.ptr)
g.close()
