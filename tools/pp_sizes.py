#!/usr/bin/env python3
"""place_pending per call over the request-count range, ONE method for every size: `calls` distinct batches of k requests over the
config-3 table, warmed, wall clock around the C call alone (ctypes), first touch (the batch's rows are pending: each call has
rows of its own, nothing is reset inside the clock) and sticky (the same batches again: every object is placed), from device-
resident arrays (rio_gp_place_pending_dev) and from host buffers (rio_gp_place_pending).
Usage: pp_sizes.py [calls=60] [sizes=4096,4097,...]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
from hipbuf import DevBuf

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 256, 1000, 4096, 4097, 6000, 8192, 16384, 65536, 65537, 131072, 262143]
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
g = rio_gp.GpuPlacement(n, m)
g.set_nodes(cfg["cap"], cfg["alive"])
g.set_objects(n, cfg["load"], cfg["aff"])
L, h = g._L, g._h
vp = C.c_void_p
perm = np.random.default_rng(5).permutation(n).astype(np.uint32)  # distinct rows: a batch's rows are its own
reqp = cfg["aff"][perm]
none = np.full(n, 0xFFFFFFFF, np.uint32)
out = {"calls": calls, "table": "config 3 (10 M x 1 024)", "per_size": {}}
for k in sizes:
    nb = min(calls + 4, n // k)
    rec = {}
    # ---- device-resident
    g.set_assign(none); g.get_nodes()
    d_idx, d_req = DevBuf(perm[: nb * k]), DevBuf(reqp[: nb * k])
    d_node, d_flag = DevBuf(nbytes=4 * k + 64), DevBuf(nbytes=4 * k + 64)
    for phase in ("first_touch", "sticky"):
        ts = []
        for b in range(nb):
            off = 4 * b * k
            off -= off % 16  # (the one-workgroup kernel wants 16-byte aligned arrays)
            t0 = time.perf_counter()
            rc = L.rio_gp_place_pending_dev(h, k, vp(d_idx.ptr + off), vp(d_req.ptr + off), vp(d_node.ptr), vp(d_flag.ptr))
            ts.append(time.perf_counter() - t0)
            assert rc == 0, g.last_error() if hasattr(g, "last_error") else rc
        t = float(np.median(ts[4:])) if nb > 6 else float(np.median(ts[1:]))
        rec["dev_" + phase] = {"us": round(t * 1e6, 2), "req_per_s": k / t, "p10_us": round(float(np.percentile(ts[2:], 10)) * 1e6, 2),
                               "p90_us": round(float(np.percentile(ts[2:], 90)) * 1e6, 2), "calls": nb}
    flags = d_flag.to_host()[:k]
    rec["dev_sticky"]["flags_last_call"] = {int(f): int(c) for f, c in zip(*np.unique(flags, return_counts=True))}
    for b in (d_idx, d_req, d_node, d_flag):
        b.free()
    # ---- host buffers
    g.set_assign(none); g.get_nodes()
    node, flag = np.empty(k, np.uint32), np.empty(k, np.uint32)
    batches = [(perm[b * k:(b + 1) * k].copy(), reqp[b * k:(b + 1) * k].copy()) for b in range(nb)]
    for phase in ("first_touch", "sticky"):
        ts = []
        for ii, rq in batches:
            t0 = time.perf_counter()
            rc = L.rio_gp_place_pending(h, k, ii.ctypes.data_as(vp), rq.ctypes.data_as(vp), node.ctypes.data_as(vp), flag.ctypes.data_as(vp))
            ts.append(time.perf_counter() - t0)
            assert rc == 0
        t = float(np.median(ts[4:])) if nb > 6 else float(np.median(ts[1:]))
        rec["host_" + phase] = {"us": round(t * 1e6, 2), "req_per_s": k / t, "p10_us": round(float(np.percentile(ts[2:], 10)) * 1e6, 2),
                                "p90_us": round(float(np.percentile(ts[2:], 90)) * 1e6, 2), "calls": nb}
    out["per_size"][str(k)] = rec
    print("%7d  dev ft %8.1f us  sticky %8.1f us | host ft %8.1f us  sticky %8.1f us" % (
        k, rec["dev_first_touch"]["us"], rec["dev_sticky"]["us"], rec["host_first_touch"]["us"], rec["host_sticky"]["us"]), file=sys.stderr)
worst = min(v["dev_first_touch"]["req_per_s"] for kk, v in out["per_size"].items() if int(kk) >= 4096)
out["min_req_per_s_from_4096_dev_first_touch"] = worst
print(json.dumps(out))
g.close()
