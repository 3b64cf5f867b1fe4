#!/usr/bin/env python3
"""Per-operation rates of the hot path on one MI355X: every §8a row, kernel-side
(inputs resident in HBM, HIP events on the library stream) and, where the ABI takes host buffers,
PCIe-inclusive wall clock.  Writes gpurun_out/<tag>_ops.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import ctypes as C
import numpy as np
import torch
import rio_gp, synth

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
L = rio_gp.lib()
out = {"n": n, "m": m, "ops": {}}


def rec(name, units, seconds, bytes_per_unit, note=""):
    out["ops"][name] = {"units": units, "seconds": seconds, "per_s": units / seconds,
                        "algorithmic_GBps": units * bytes_per_unit / seconds / 1e9, "bytes_per_unit": bytes_per_unit,
                        "note": note}
    print("%-34s %12.4e /s  %8.1f GB/s (alg)  %s" % (name, units / seconds, units * bytes_per_unit / seconds / 1e9, note))


def timed(g, fn, reps):
    fn()
    g.sync()
    g.timer_begin()
    for _ in range(reps):
        fn()
    return g.timer_end() * 1e-3 / reps


def mk(cur=None, cap=None, alive=None):
    g = rio_gp.GpuPlacement(n, m)
    g.set_nodes(cfg["cap"] if cap is None else cap, cfg["alive"] if alive is None else alive)
    g.set_objects(n, cfg["load"], cfg["aff"])
    if cur is not None:
        g.set_assign(cur)
    return g


warm = synth.warm_assign(n, m)
# --- A2 lookup / A3 update / A5 remove, device-resident batches of 10 M ---
g = mk(cur=warm)
h = g.handle
idx = torch.from_numpy((synth.r(np.arange(n, dtype=np.uint64), 7) % np.uint64(n)).astype(np.int64)).to(torch.int32).cuda()
node = torch.from_numpy(synth.warm_assign(n, m, stream=8).astype(np.int64)).to(torch.int32).cuda()
outb = torch.empty(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
vp = C.c_void_p
t = timed(g, lambda: L.rio_gp_lookup_batch_dev(h, n, vp(idx.data_ptr()), vp(outb.data_ptr())), 5)
rec("lookup_batch_dev (random idx)", n, t, 12, "includes the per-call stats read-back + sync")
seq = torch.arange(n, dtype=torch.int32, device="cuda")
t = timed(g, lambda: L.rio_gp_lookup_batch_dev(h, n, vp(seq.data_ptr()), vp(outb.data_ptr())), 5)
rec("lookup_batch_dev (sequential idx)", n, t, 12)
# object popularity is not uniform in an actor system: Zipf(1.1) over a million hot objects scattered over the table
zr = (synth.zipf_loads(n, s=1.1, kmax=1 << 20, stream=11).astype(np.uint64) - np.uint64(1)) * np.uint64(2654435761) % np.uint64(n)
zidx = torch.from_numpy(zr.astype(np.int64)).to(torch.int32).cuda()
t = timed(g, lambda: L.rio_gp_lookup_batch_dev(h, n, vp(zidx.data_ptr()), vp(outb.data_ptr())), 5)
rec("lookup_batch_dev (Zipf(1.1) popularity)", n, t, 12, "indices drawn Zipf(1.1) over 2^20 hot objects scattered over the 10 M rows")
t = timed(g, lambda: L.rio_gp_update_batch_dev(h, n, vp(idx.data_ptr()), vp(node.data_ptr())), 5)
rec("update_batch_dev (random idx, dups)", n, t, 8, "2 kernels: elect, apply (the winner resets its scratch slot)")
t = timed(g, lambda: L.rio_gp_remove_batch_dev(h, n // 10, vp(idx.data_ptr())), 5)
rec("remove_batch_dev", n // 10, t, 8)
# host-pointer (PCIe-inclusive) forms
hidx = idx.cpu().numpy().astype(np.uint32)
t0 = time.perf_counter(); g.lookup_batch(hidx[:1_000_000]); t = time.perf_counter() - t0
rec("lookup_batch host buffers 1M", 1_000_000, t, 12, "PCIe-inclusive wall clock (H2D idx + kernel + D2H out)")
# --- A4 clean_server(s): 4 B/row scan --- (steady state: 1 untimed call, then the mean of 5)
def timed_clean(fn):
    ts = []
    for k in range(6):
        g.set_assign(warm)
        g.get_nodes()
        t0 = time.perf_counter(); ev = fn(); t = time.perf_counter() - t0
        if k:
            ts.append(t)
    return float(np.mean(ts)), ev
t, ev = timed_clean(lambda: g.clean_server(3))
rec("clean_server(1 node) sync call", n, t, 4, "evicted %d; wall clock of the synchronous ABI call" % ev)
dead = list(np.flatnonzero(synth.churn_mask(m, 1) == 0))
t, ev = timed_clean(lambda: g.clean_servers(dead))
rec("clean_servers(10% nodes) sync call", n, t, 4, "evicted %d" % ev)
g.close()

# --- A7 whole-table solve variants (synchronous rio_gp_solve: includes host verdict read + fix-up launches) ---
def solve_variant(name, g, reps, note=""):
    st = g.solve()
    g.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        st = g.solve()
    t = (time.perf_counter() - t0) / reps
    rec(name, n, t, 16, note + " | slow_path=%d cut_nodes=%d spilled=%d unplaced=%d" % (
        st["slow_path"], st["cut_nodes"], st["spilled"], st["unplaced"]))
    return st

g = mk()
solve_variant("solve cold c3 (sync call)", g, 50, "fast path")
g.close()
g = mk(cur=warm)
solve_variant("solve warm c3 (sync call)", g, 50, "all kept")
g.close()
g = mk(cur=warm, alive=synth.churn_mask(m, 1))
solve_variant("solve churn 10% nodes dead", g, 20, "config 5 tick: evict + re-place, spill for dead affinity")
# churn stream: a different 10 % dies every tick, committed ticks
ticks, moved = 20, 0
g.sync(); t0 = time.perf_counter()
for k in range(ticks):
    g.set_alive_all(synth.churn_mask(m, 2 + k))
    st = g.tick()
    moved += st["claimed"] + st["spilled"]
t = time.perf_counter() - t0
rec("churn stream: ticks (all rows decided)", n * ticks, t, 16, "%d committed ticks" % ticks)
rec("churn stream: evicted-and-re-placed", moved, t, 16, "objects actually moved per second")
g.close()
capc = (cfg["cap"].astype(np.float64) * 0.72).astype(np.uint64)   # 0.9x of total load: cuts + spills + unplaced
g = mk(cap=capc)
solve_variant("solve cold, capacity 0.9x load", g, 10, "contended: cut + water-fill path")
g.close()
skew = np.minimum((np.random.default_rng(1).pareto(1.1, n)).astype(np.int64), m - 1).astype(np.uint32)
g = rio_gp.GpuPlacement(n, m); g.set_nodes(cfg["cap"], cfg["alive"]); g.set_objects(n, cfg["load"], skew)
solve_variant("solve cold, Pareto-skewed affinity", g, 10, "hot nodes oversubscribed")
g.close()
# --- place_pending: batches of requests from host buffers ---
g = mk()
for k in (1, 1000, 100_000, 1_000_000):
    ii = hidx[:k]
    rq = cfg["aff"][ii]
    g.set_assign(np.full(n, 0xFFFFFFFF, np.uint32))
    g.get_nodes()  # rebuilds `used` after the raw set_assign, outside the timed call
    t0 = time.perf_counter(); g.place_pending(ii, rq); t = time.perf_counter() - t0
    rec("place_pending batch=%d (host buffers)" % k, k, t, 28, "PCIe-inclusive, cold rows")
g.close()
# --- place_pending_dev: request / result arrays resident in HBM, kernel-side time (HIP events on the library stream) ---
g = mk()
onode = torch.empty(n, dtype=torch.int32, device="cuda")
oflag = torch.empty(n, dtype=torch.int32, device="cuda")
req = torch.from_numpy(cfg["aff"].astype(np.int64)).to(torch.int32).cuda()
perm = torch.from_numpy((synth.r(np.arange(n, dtype=np.uint64), 9) % np.uint64(n)).astype(np.int64)).to(torch.int32).cuda()
reqp = req[perm.long()].contiguous()
torch.cuda.synchronize()
for k in (1000, 100_000, 1_000_000, 10_000_000):
    ts = []
    for rep in range(4):
        g.set_assign(np.full(n, 0xFFFFFFFF, np.uint32))
        g.get_nodes()
        g.sync()
        g.timer_begin()
        L.rio_gp_place_pending_dev(h2 := g.handle, k, vp(perm.data_ptr()), vp(reqp.data_ptr()), vp(onode.data_ptr()), vp(oflag.data_ptr()))
        ts.append(g.timer_end() * 1e-3)
    rec("place_pending_dev batch=%d (cold rows)" % k, k, float(np.mean(ts[1:])), 28,
        "device-resident requests, wall clock of the call (validated on the device, one verdict wait)")
g.close()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", tag + "_ops.json"), "w"), indent=1)
