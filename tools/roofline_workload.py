#!/usr/bin/env python3
"""One PHASE of the hot path per process, for the rocprofv3 passes of tools/kernel_roofline.sh (kernel trace + the two PMC
passes): every kernel DESIGN.md section 4 names runs in exactly one phase with known algorithmic bytes, so that its
duration, its algorithmic bytes and its counter bytes can be put side by side (profiles/round5_kernel_roofline.json).
Usage: roofline_workload.py <fast|churn|churn_noinc|churn_unpacked|contended|contended_packed|crud|crud_plain|lookup_seq|clean1|pp|pp10|pp_small|pp_256|pp_mid|probes> [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth

phase = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
g = rio_gp.LabPlacement(n, m)
g.set_nodes(cfg["cap"], cfg["alive"])
g.set_objects(n, cfg["load"], cfg["aff"])
if phase == "fast":                      # k_scan<.., TPI 2>, k_resolve
    for _ in range(reps * 3):
        g.solve_async()
    g.solve_wait()
elif phase in ("churn", "churn_noinc", "churn_unpacked"):   # k_inc_scan + k_rebal (or k_scan<COMPACT>), k_resolve<SEARCH>, k_fill rounds
    if phase == "churn_unpacked":
        g.set_compact("never", cut_pack="never")
    if phase == "churn_noinc":
        g.set_compact("auto", inc="never")
    g.set_assign(synth.warm_assign(n, m))
    g.tick()
    for k in range(reps + 2):
        g.set_alive_all(synth.churn_mask(m, 2 + k))
        st = g.tick()
    if os.environ.get("RIO_KROOF_DIR"):
        import json
        json.dump({"pending_rows": st["claimed"] + st["spilled"] + st["unplaced"], "last_tick": st},
                  open(os.path.join(os.environ["RIO_KROOF_DIR"], "churn_stats.json"), "w"))
elif phase in ("contended", "contended_packed"):   # the same fix-up kernels over ALL rows (cold table, capacity 0.9 x load);
    # _packed: the cut pass packs the rows that go on to the water-fill (what the library does from the second such solve on)
    g.set_nodes((cfg["cap"].astype(np.float64) * 0.72).astype(np.uint64), cfg["alive"])
    g.set_compact("auto", cut_pack="always" if phase == "contended_packed" else "never")
    for _ in range(reps):
        g.solve()
elif phase in ("crud", "crud_plain", "pp", "pp10", "pp_small", "pp_256", "pp_mid", "clean1", "lookup_seq"):
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from hipbuf import DevBuf
    L, h, vp = rio_gp.lab_lib(), g.handle, C.c_void_p
    idx = DevBuf((synth.r(np.arange(n, dtype=np.uint64), 7) % np.uint64(n)).astype(np.uint32))
    node = DevBuf(synth.warm_assign(n, m, stream=8))
    outb, flg = DevBuf(nbytes=4 * n), DevBuf(nbytes=4 * n)
    seq = DevBuf(np.arange(n, dtype=np.uint32))
    if phase in ("crud", "crud_plain"):  # k_lookup4 (random), update / remove (window-partitioned, or the plain per-entry kernels), k_clean
        if phase == "crud_plain":
            g.set_compact("auto", partitioned_crud=False)
        g.set_assign(synth.warm_assign(n, m))
        for _ in range(3):
            L.rio_gp_lookup_batch_dev(h, n, vp(idx.ptr), vp(outb.ptr))
        for _ in range(3):
            L.rio_gp_update_batch_dev(h, n, vp(idx.ptr), vp(node.ptr))
        for _ in range(3):
            L.rio_gp_remove_batch_dev(h, n, vp(idx.ptr))
            g.set_assign(synth.warm_assign(n, m))
        for k in range(3):
            g.clean_servers(list(np.flatnonzero(synth.churn_mask(m, 1 + k) == 0)))
            g.set_assign(synth.warm_assign(n, m))
    elif phase == "lookup_seq":          # k_lookup4 over sequential indices: the coalesced ceiling of the same kernel
        g.set_assign(synth.warm_assign(n, m))
        for _ in range(5):
            L.rio_gp_lookup_batch_dev(h, n, vp(seq.ptr), vp(outb.ptr))
    elif phase == "clean1":              # k_clean, one node (clean_server)
        for k in range(4):
            g.set_assign(synth.warm_assign(n, m))
            g.get_nodes()
            g.clean_server(3 + k)
    elif phase in ("pp_small", "pp_256"):   # 4 096 requests (three launches) / 256 (k_pp_one, one launch) from host buffers, first touch then sticky
        kk = 4096 if phase == "pp_small" else 256
        ii = (synth.r(np.arange(kk, dtype=np.uint64), 7) % np.uint64(n)).astype(np.uint32)
        rq = cfg["aff"][ii]
        for _ in range(reps):
            g.place_pending(ii, rq)
    elif phase == "pp_mid":              # 16 384 device-resident requests: the general path (k_ppm_first / gather / solve / output)
        for r in range(reps):
            off = 4 * 16384 * r
            L.rio_gp_place_pending_dev(h, 16384, vp(idx.ptr + off), vp(node.ptr + off), vp(outb.ptr), vp(flg.ptr))
    else:                                # the window-sorted request path (answers from the window kernel), 1 M / 10 M requests on a cold table
        k = 10_000_000 if phase == "pp10" else 1_000_000
        for _ in range(3):
            g.set_assign(np.full(n, 0xFFFFFFFF, np.uint32))
            g.get_nodes()
            L.rio_gp_place_pending_dev(h, k, vp(idx.ptr), vp(node.ptr), vp(outb.ptr), vp(flg.ptr))
elif phase == "probes":                  # known traffic: calibration of FETCH_SIZE / WRITE_SIZE
    for mode in (4, 0, 3):
        g.stream_probe(mode, 10)
g.close()
