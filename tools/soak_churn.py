#!/usr/bin/env python3
"""Soak: a long churn stream (liveness flips, load/affinity edits, removals, micro place_pending batches between ticks),
every tick compared with the CPU oracle — looks for rare scheduling-dependent differences that short tests cannot see.
Usage: soak_churn.py [ticks=400] [rows=400000] [nodes=1024] [seed=1]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, pyoracle, synth

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 400
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
m = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(seed)
cfg = synth.config("c3", n_override=n)
load, aff = cfg["load"].copy(), (synth.r(np.arange(n), 1) % np.uint64(m)).astype(np.uint32)
cap = synth.uniform_cap(load, m, headroom=1.12)
alive = np.ones(m, np.uint8)
g = rio_gp.GpuPlacement(n, m)
g.set_nodes(cap, alive)
g.set_objects(n, load, aff)
cur = synth.warm_assign(n, m)
g.set_assign(cur)
ref = cur.copy()
t0 = time.time()
slow = moved = 0
for k in range(ticks):
    # membership churn: ~8 % of the nodes flip
    flip = rng.random(m) < 0.08
    alive = np.where(flip, 1 - alive, alive).astype(np.uint8)
    if alive.sum() < m // 2:
        alive[:] = 1
    g.set_alive_all(alive)
    # a few objects change load / affinity, a few are removed
    e = rng.integers(0, n, 200).astype(np.uint32)
    load[e] = rng.integers(0, 5000, 200).astype(np.uint32)
    aff[e] = rng.integers(0, m, 200).astype(np.uint32)
    g.set_object_attrs(e, load[e], aff[e])
    rm = rng.integers(0, n, int(rng.choice([1, 3, 100, 400, 5000]))).astype(np.uint32)   # micro / medium batches
    g.remove_batch(rm)
    ref[rm] = 0xFFFFFFFF
    want, used, ost = pyoracle.tick(ref, load, aff, cap, alive)
    st = g.tick()
    got = g.get_assign()
    if not np.array_equal(got, want) or st != ost or not np.array_equal(g.get_nodes()[2], used):
        bad = np.flatnonzero(got != want)
        print(json.dumps({"tick": k, "mismatch_rows": int(len(bad)), "first": bad[:5].tolist(), "gpu_stats": st, "oracle_stats": ost}))
        sys.exit(3)
    ref = want
    slow += st["slow_path"]
    moved += st["claimed"] + st["spilled"]
    # a micro-batch of requests between ticks (k_pp_small or the general path), against the oracle's place_pending
    q = int(rng.choice([1, 3, 8, 200, 300, 1500, 5000]))   # one workgroup | mapped pinned memory | staging copies
    idx = rng.integers(0, n, q).astype(np.uint32)
    live = np.flatnonzero(alive)
    req = live[rng.integers(0, len(live), q)].astype(np.uint32)
    node, flag = g.place_pending(idx, req)
    wnode, wflag = pyoracle.place_pending(ref, load, cap, alive, used, idx, req)
    if not (np.array_equal(node, wnode) and np.array_equal(flag, wflag) and np.array_equal(g.get_assign(), ref)):
        print(json.dumps({"tick": k, "place_pending_mismatch": True}))
        sys.exit(4)
    if k % 5 == 4:
        # asynchronous ticks with nothing changed in between: the first may need the fix-up (requests above, unplaced rows),
        # the later ones run without the speculative fix-up launches once a fast tick's verdict has landed
        wants = []
        for j in range(4):
            g.tick_async()
            ref, used, ost = pyoracle.tick(ref, load, aff, cap, alive)
            wants.append(ost)
            if j == 1:
                time.sleep(0.002)
        if g.tick_wait() != wants or not np.array_equal(g.get_assign(), ref) or not np.array_equal(g.get_nodes()[2], used):
            print(json.dumps({"tick": k, "async_ticks_mismatch": True}))
            sys.exit(6)
    lq = rng.integers(0, n, int(rng.choice([1, 4, 5, 256, 257, 3000, 20000]))).astype(np.uint32)
    if not np.array_equal(g.lookup_batch(lq), ref[lq]):
        print(json.dumps({"tick": k, "lookup_mismatch": True}))
        sys.exit(5)
print(json.dumps({"ticks": ticks, "rows": n, "nodes": m, "seed": seed, "slow_ticks": slow, "objects_moved": int(moved),
                  "all_ticks_equal_oracle": True, "wall_s": round(time.time() - t0, 1)}))
g.close()
