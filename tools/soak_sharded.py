#!/usr/bin/env python3
"""Soak of the row-sharded solve between PROCESSES (one per rank, all on the GPU of this box): peer-to-peer windows,
hundreds of committed ticks with membership churn, every rank checks its own rows and the global `used` vector against
the whole-table oracle after every tick, plus bursts of back-to-back asynchronous solves (window-slot reuse).
Usage: soak_sharded.py [world=3] [ticks=150] [rows=150000] [nodes=96] [seed=1]
Sharing ONE GPU, the ranks' kernels must be co-resident (a rank that spins for a peer whose scan cannot get its 16 waves on
every CU never sees that peer): keep world x ceil(nodes/4) x 4 waves well under half the chip — 3 ranks at 1 024 nodes, 5
at <= 128 — a limit of this test set-up, not of one process per GPU."""
import os, sys, json, socket, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np


def proc(rank, world, port, ticks, n, m, seed, out_dir):
    import torch
    import torch.distributed as dist
    import rio_gp, sharded, pyoracle, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        rng = np.random.default_rng(seed)  # the same stream on every rank: identical tables and masks
        load = rng.integers(0, 4000, n).astype(np.uint32)
        aff = rng.integers(0, m, n).astype(np.uint32)
        cur = rng.integers(0, m, n).astype(np.uint32)
        cap = np.full(m, int(load.astype(np.uint64).sum() * 1.1 / m), np.uint64)
        b = sharded.shard_bounds(n, world)
        lo, hi = b[rank], b[rank + 1]
        g = rio_gp.GpuPlacement(max(hi - lo, 1), m)
        alive = np.ones(m, np.uint8)
        g.set_nodes(cap, alive, m=m)
        g.set_objects(hi - lo, load[lo:hi], aff[lo:hi])
        g.set_assign(cur[lo:hi])
        eng = sharded.HipShardEngine(g, 0, torch.cuda.Stream(torch.device("cuda", 0)))
        sol = sharded.ShardedSolver([eng], sharded.P2PExchange(eng))
        ref = cur.copy()
        slow = 0
        for k in range(ticks):
            flip = rng.random(m) < 0.1
            alive = np.where(flip, 1 - alive, alive).astype(np.uint8)
            if alive.sum() < m // 2:
                alive[:] = 1
            g.set_alive_all(alive)
            if k % 7 == 0:                       # burst of uncommitted solves: slots and tags are reused back to back
                for _ in range(9):
                    sol.solve_async()
                sol.solve_wait()
            if k % 3 == 2:
                # asynchronous committed ticks (rio_gp_shard_tick_async): three in a row, a liveness flip before each of the
                # last two, nothing waits on the host until the records are collected
                wants = []
                for j in range(3):
                    if j:
                        flip = rng.random(m) < 0.05
                        alive = np.where(flip, 1 - alive, alive).astype(np.uint8)
                        if alive.sum() < m // 2:
                            alive[:] = 1
                        g.set_alive_all(alive)
                    sol.tick_async()
                    ref, used, ost = pyoracle.tick(ref, load, aff, cap, alive)
                    wants.append(ost)
                sts = sol.tick_wait()
                if not (sts == wants and np.array_equal(g.get_assign(), ref[lo:hi]) and np.array_equal(g.get_nodes()[2], used)):
                    raise SystemExit("rank %d: asynchronous ticks at %d differ from the oracle" % (rank, k))
                slow += sum(x["slow_path"] for x in sts)
                continue
            st = sol.tick()
            want, used, ost = pyoracle.tick(ref, load, aff, cap, alive)
            if not (np.array_equal(g.get_assign(), want[lo:hi]) and st == ost and np.array_equal(g.get_nodes()[2], used)):
                raise SystemExit("rank %d: tick %d differs from the oracle" % (rank, k))
            ref = want
            slow += st["slow_path"]
        json.dump({"rank": rank, "ticks": ticks, "slow_ticks": slow}, open(os.path.join(out_dir, "r%d.json" % rank), "w"))
        g.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    import tempfile
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 150_000
    m = int(sys.argv[4]) if len(sys.argv) > 4 else 96
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    d = tempfile.mkdtemp()
    t0 = time.time()
    mp.spawn(proc, args=(world, port, ticks, n, m, seed, d), nprocs=world, join=True)
    res = [json.load(open(os.path.join(d, "r%d.json" % r))) for r in range(world)]
    print(json.dumps({"world": world, "ticks": ticks, "rows": n, "nodes": m, "seed": seed, "slow_ticks": res[0]["slow_ticks"],
                      "all_ranks_equal_oracle_every_tick": True, "wall_s": round(time.time() - t0, 1)}))
