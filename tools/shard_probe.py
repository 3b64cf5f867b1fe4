#!/usr/bin/env python3
"""Where does a row-sharded fast-path step spend its time on ONE rank?  Host enqueue time vs device time of
rio_gp_shard_solve_async, and whether the step (RCCL all-gather included) can be captured into a hipGraph."""
import os, sys, time, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import torch.distributed as dist
import rio_gp, synth, sharded
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
cfg = synth.config("c3")
g = rio_gp.GpuPlacement(cfg["n"], cfg["m"]); g.set_nodes(cfg["cap"], cfg["alive"]); g.set_objects(cfg["n"], cfg["load"], cfg["aff"])
eng = sharded.HipShardEngine(g, 0)
ex = sharded.NativeRcclExchange(eng)
sol = sharded.ShardedSolver([eng], ex)
L = sharded._lib()
res = {}
for _ in range(20): sol.solve_async()
sol.solve_wait()
K = 200
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K): L.rio_gp_shard_solve_async(g.handle)
t1 = time.perf_counter(); sol.solve_wait(); torch.cuda.synchronize(); t2 = time.perf_counter()
res["native_enqueue_us"] = (t1 - t0) / K * 1e6; res["native_total_us"] = (t2 - t0) / K * 1e6
# plain (unsharded) async solve for comparison
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K): g.solve_async()
t1 = time.perf_counter(); g.solve_wait(); t2 = time.perf_counter()
res["plain_enqueue_us"] = (t1 - t0) / K * 1e6; res["plain_total_us"] = (t2 - t0) / K * 1e6
print(json.dumps(res))
dist.destroy_process_group()
