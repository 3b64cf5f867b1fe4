#!/usr/bin/env python3
"""bench.py — placement decisions/sec + achieved HBM GB/s of the whole-table solve.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on
rank 0.  A "step" is one whole-table solve (rio_gp_solve_async: every row of the table gets a
placement decision) over the synthetic table of BASELINE.json config 3 — 10 M objects x 1 024
nodes, Zipf(1.1) loads, cold start (every object pending) — already resident in HBM.
  value        = decisions of all ranks / max-over-ranks wall time of the K steps
  roofline     = k_scan (the streaming kernel, >90 % of a step): algorithmic 16 B/decision
                 (read cur+load+aff, write assign) / its per-launch HIP-event time
  cpu_baseline = the CPU oracle port of the reference's per-object path, on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

ALGO_BYTES_PER_DECISION = 16
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", help="c3 (headline) | c3w | c2 | c4shard | c5 (config 5: churn ticks, N=1 only)")
    ap.add_argument("--objects", type=int, default=0, help="override rows per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", default="p2p", choices=("p2p", "native", "torch"),
                    help="sharded runs: how the per-rank load records travel — p2p: stores into the peers' HBM windows over "
                         "xGMI, no collective call; native: ncclAllGather issued by the library; torch: torch.distributed "
                         "all_gather.  Falls back p2p -> native -> torch if a path cannot be set up")
    ap.add_argument("--backend", default="nccl", help="control-plane backend; 'gloo' only for --same-device flow tests")
    ap.add_argument("--same-device", action="store_true",
                    help="flow test on a 1-GPU box: every rank uses device 0 (RCCL refuses that, so use --backend gloo; "
                         "the peer-to-peer windows work between processes on one device); numbers are meaningless")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="sharded runs: keep the all-gather on the scan stream (no overlap with the next solve's scan)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="N=1 only: run the row-sharded path (RCCL group of one rank) to price its extra kernels and launches")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000)
    ap.add_argument("--no-cold", action="store_true",
                    help="skip the beyond-the-Infinity-Cache data point (the same kernel over a 4x table, 640 MB of columns)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "r03_traffic.json"),
                    help="PMC-derived HBM bytes per k_scan launch (tools/pmc_traffic.py); null if absent")
    return ap.parse_args()


def cpu_baseline(cfg, sample):
    """Time the oracle port of the reference's per-object path on this box's host cores (bounded)."""
    import pyoracle
    n = min(sample, cfg["n"])
    aff = np.ascontiguousarray(cfg["aff"][:n])
    cores = os.cpu_count() or 1
    t1, _ = pyoracle.bench_policy(n, cfg["m"], aff, threads=1)                       # cold: miss -> first touch -> update
    nT = min(n, 1_000_000)
    tT, _ = pyoracle.bench_policy(nT, cfg["m"], aff[:nT], threads=cores)             # all cores on ONE shared map
    nW = min(n, 20_000)
    tW, _ = pyoracle.bench_policy(nW, cfg["m"], aff[:nW], threads=1, warm=True)      # warm: sticky hit + O(M) is_active
    v1, vT = n / t1, nT / tT
    best_v, best_c = (v1, 1) if v1 >= vT else (vT, cores)
    # "best reasonable CPU": the array solver (same algorithm as the GPU), one thread, full table
    t0 = time.perf_counter()
    pyoracle.tick(cfg["cur"], cfg["load"], cfg["aff"], cfg["cap"], cfg["alive"])
    t_arr = time.perf_counter() - t0
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {
        "value": best_v, "unit": "decisions/s", "cores": best_c, "kind": "port",
        "sample": "cold get_or_create_placement per object (service.rs:193-254 restated in C++: string keys, "
                  "unordered_map behind a shared_mutex, %d-member LocalStorage) over the first %d objects of the "
                  "workload on 1 thread, and over the first %d objects on all %d hardware threads sharing one map; "
                  "value = the faster of the two" % (cfg["m"], n, nT, cores),
        "value_1thread": v1, "value_allcores": vT, "host_cores": cores, "cpu_model": model,
        "warm_sticky_hits_1thread": {"value": nW / tW, "unit": "decisions/s", "rows": nW,
                                     "note": "every call hits and pays the O(M) is_active member scan (cluster/storage/mod.rs:95-110)"},
        "array_oracle_1thread": {"value": cfg["n"] / t_arr, "unit": "decisions/s", "rows": cfg["n"],
                                 "note": "oracle/placement_oracle.c orc_tick: the GPU algorithm run sequentially on dense arrays"},
    }


def bench_churn(a, g, cfg, saved_stdout):
    """BASELINE.json config 5: the config-3 table, warm; every step = one liveness push (10 % of the nodes down, the
    previous casualties back) + one committed whole-table tick (evict + re-place through the fix-up path)."""
    import synth
    n, m = cfg["n"], cfg["m"]
    g.set_assign(synth.warm_assign(n, m))
    g.tick()
    masks = [synth.churn_mask(m, 2 + k) for k in range(a.warmup + a.steps)]
    for k in range(a.warmup):
        g.set_alive_all(masks[k])
        g.tick()
    g.sync()
    moved = slow = 0
    t0 = time.perf_counter()
    for k in range(a.steps):
        g.set_alive_all(masks[a.warmup + k])
        st = g.tick()
        moved += st["claimed"] + st["spilled"]
        slow += st["slow_path"]
    dt = time.perf_counter() - t0
    out = {
        "metric": "placement decisions/sec, 10M objects x 1 024 nodes with 10 % node-failure churn per tick", "value": n * a.steps / dt,
        "unit": "decisions/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "config 5: %d objects x %d nodes, Zipf(1.1) load, cap 1.25x, warm; per step 10 %% of the nodes "
                               "flip and one committed tick evicts and re-places their objects" % (n, m),
                   "step": "rio_gp_set_alive_all + rio_gp_tick (synchronous: the host reads every tick's counters)",
                   "slow_path_steps": slow},
        "objects_moved_per_s": moved / dt, "stats_last_step": st,
        "roofline": {"bound": "hbm", "achieved": ALGO_BYTES_PER_DECISION * n * a.steps / dt / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": ALGO_BYTES_PER_DECISION * n * a.steps / dt / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": "whole tick (8 dependent launches + host turn-around; latency-bound, DESIGN.md section 5)"},
    }
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    print(json.dumps(out), flush=True)


def main():
    a = parse()
    # stdout carries exactly ONE line, the JSON: libraries that write to C stdout (RCCL prints a version banner from
    # every rank) are pointed at stderr for the whole run; fd 1 is restored only for rank 0's final print
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    import torch
    import rio_gp
    import synth
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    rio_gp.build()
    if a.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or a.force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    n_over = a.objects or None
    if a.workload == "c5" and world > 1:
        raise SystemExit("--workload c5 is a single-GPU line (the row-sharded churn tick is covered by tools/soak_sharded.py)")
    per_rank = synth.config("c3" if a.workload == "c5" else a.workload, n_override=n_over, start=0)  # shapes only
    n_local = per_rank["n"]
    # weak scaling: every rank owns n_local consecutive rows of ONE table of world*n_local rows (rank order =
    # index order); capacities are set from the GLOBAL load, exactly as the unsharded config would
    cfg = synth.config(a.workload, n_override=n_local, start=rank * n_local) if world > 1 else per_rank
    n, m = cfg["n"], cfg["m"]
    if dist is not None:
        tot = torch.tensor([int(cfg["load"].astype(np.uint64).sum())], device="cuda", dtype=torch.int64)
        dist.all_reduce(tot)  # set-up only, not the data path
        cfg["cap"] = np.full(m, -((-int(tot.item()) * 1250) // (1000 * m)), dtype=np.uint64)

    g = rio_gp.GpuPlacement(n, m, device=local_rank)
    g.set_nodes(cfg["cap"], cfg["alive"])
    g.set_objects(n, cfg["load"], cfg["aff"])
    if a.workload == "c3w":
        g.set_assign(cfg["cur"])
    if a.workload == "c5":
        return bench_churn(a, g, cfg, saved_stdout)

    def barrier():
        if dist is not None:
            dist.barrier()
        g.sync()
        torch.cuda.synchronize()

    if dist is None:
        step, wait = g.solve_async, g.solve_wait
        for _ in range(a.warmup):
            step()
        if a.warmup:
            st, n_slow = wait()
    else:
        # row-sharded solve: k_scan -> exchange of the (2m+8)-word record -> global resolve; the verdicts are read once
        # at the end, as at N=1.  Exchange ladder p2p -> native -> torch: a path that cannot be set up, or whose warm-up
        # fails on ANY rank (agreed through an all-reduce), is dropped for the next one on EVERY rank.
        import sharded
        eng = sharded.HipShardEngine(g, local_rank)
        ladder = ["p2p", "native", "torch"]
        ladder = ladder[ladder.index(a.exchange):]
        sol = None
        for kind in ladder:
            ok, ex, why = 1, None, ""
            try:
                ex = {"p2p": sharded.P2PExchange, "native": sharded.NativeRcclExchange}[kind](eng) if kind != "torch" \
                    else sharded.DistExchange()
                sol = sharded.ShardedSolver([eng], ex, spill_rounds=2, pipeline=(kind == "torch" and not a.no_pipeline))
                for _ in range(max(a.warmup, 2)):
                    sol.solve_async()
                st, n_slow = sol.solve_wait()
                # an exchange that delivers wrong records must not survive the warm-up: every row decided exactly
                # once and every unit of load accounted for, on the GLOBAL table
                if st["kept"] + st["claimed"] + st["spilled"] + st["unplaced"] != n * world or \
                        st["load_kept"] + st["load_claimed"] + st["load_spilled"] + st["load_unplaced"] != int(tot.item()):
                    raise RuntimeError("exchange '%s' produced inconsistent global stats: %r" % (kind, st))
            except Exception as e:  # set-up failure raises on every rank; a warm-up failure may be local
                ok, why = 0, str(e)
            flag = torch.tensor([ok], device="cuda", dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                a.exchange = kind
                break
            print("rank %d: exchange '%s' dropped (%s)" % (rank, kind, why or "failed on another rank"), file=sys.stderr)
            if kind == "p2p" and ex is not None:
                ex.close()
            sol = None
        if sol is None:
            raise SystemExit("no exchange path could be set up")
        step, wait = sol.solve_async, sol.solve_wait

    barrier()
    t0 = time.perf_counter()
    g.timer_begin()
    for _ in range(a.steps):
        step()
    gpu_ms = g.timer_end()
    st, n_slow = wait()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert st["kept"] + st["claimed"] + st["spilled"] + st["unplaced"] == n * world
    total_decisions = n * world * a.steps

    # per-launch duration of the dominant kernel, HIP events on the library's own stream
    scan_ms, res_ms = [], []
    if n_slow == 0:
        for _ in range(max(10, min(a.steps, 100))):
            s_ms, r_ms = g.solve_profiled()
            scan_ms.append(s_ms)
            res_ms.append(r_ms)
    probe = None
    if rank == 0 and n_slow == 0:
        try:  # what this chip's memory system gives a plain grid-stride kernel with the same 3-in/1-out mix
            ms = g.stream_probe(0, 20)
            probe = {"pattern": "grid-stride 2048x256, read cur/load/aff + write one column, no other work",
                     "ms": ms, "GBps": ALGO_BYTES_PER_DECISION * n / ms / 1e6}
        except Exception as e:  # measurement aid only
            probe = {"error": str(e)}
    cold = None
    if rank == 0 and world == 1 and n_slow == 0 and not a.no_cold and a.workload == "c3":
        # The headline table (160 MB of columns) fits the 256 MiB Infinity Cache, so repeated solves are partly served
        # by it.  Same kernel, same per-row inputs tiled 4x (640 MB of columns, capacities scaled): every launch streams
        # from HBM.  Reported next to the headline, never instead of it.
        try:
            k = 4
            loadk, affk = np.tile(cfg["load"], k), np.tile(cfg["aff"], k)
            gb = rio_gp.GpuPlacement(k * n, m, device=local_rank)
            gb.set_nodes(synth.uniform_cap(loadk, m), cfg["alive"])
            gb.set_objects(k * n, loadk, affk)
            for _ in range(5):
                gb.solve_profiled()
            cs = [gb.solve_profiled()[0] for _ in range(30)]
            pm = gb.stream_probe(0, 10)
            gb.close()
            cms = float(np.mean(cs))
            cold = {"rows": k * n, "column_bytes": 16 * k * n, "kernel_ms": cms,
                    "achieved": ALGO_BYTES_PER_DECISION * k * n / (cms * 1e-3) / 1e9, "unit": "GB/s",
                    "frac": ALGO_BYTES_PER_DECISION * k * n / (cms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "stream_probe_GBps": ALGO_BYTES_PER_DECISION * k * n / pm / 1e6,
                    "note": "k_scan over the headline rows tiled 4x: beyond the 256 MiB Infinity Cache"}
        except Exception as e:  # measurement aid only
            cold = {"error": str(e)}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    scan_avg = float(np.mean(scan_ms)) if scan_ms else None
    achieved = (ALGO_BYTES_PER_DECISION * n / (scan_avg * 1e-3) / 1e9) if scan_avg else None
    traffic = None
    if a.traffic_json and os.path.exists(a.traffic_json):
        tj = json.load(open(a.traffic_json))
        traffic = tj.get("hbm_bytes_per_launch") if tj.get("n_rows") == n else None  # measured for this row count only
    out = {
        "metric": "placement decisions/sec + achieved HBM GB/s, 10M objects x 1 024 nodes",
        "value": total_decisions / dt, "unit": "decisions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": "config 3: %d objects x %d nodes per GPU, Zipf(1.1) load, cap 1.25x, cold start "
                               "(all pending)" % (n, m) if a.workload == "c3" else a.workload,
                   "objects_per_gpu": n, "nodes": m, "parallelism": "rows sharded x%d" % world,
                   "step": "rio_gp_solve_async = k_scan + k_resolve on one stream; verdicts read at the end" if dist is None
                           else {"p2p": "row-sharded solve: k_scan -> k_resolve_xchg (every workgroup stores its four nodes' local sums straight into "
                                        "every peer's HBM window over xGMI as data-tagged 8-byte words, %d B/rank, polls the same words of "
                                        "every rank and resolves its nodes); one stream, two launches, no collective call, no flag; "
                                        "verdicts read at the end",
                                 "native": "row-sharded solve: k_scan + k_resolve + pack -> ncclAllGather of %d B/rank issued by the "
                                           "library on a second stream -> k_shard_import; verdicts read at the end",
                                 "torch": "row-sharded solve: k_scan + k_resolve + pack -> torch.distributed all_gather (RCCL) of %d "
                                          "B/rank -> k_shard_import; verdicts read at the end"}[a.exchange] % (8 * (2 * m + 8)),
                   "exchange": None if dist is None else a.exchange,
                   "slow_path_steps": n_slow},
        "gpu_ms_per_step_events": gpu_ms / a.steps,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
                     "kernel": "k_scan", "kernel_ms": scan_avg,
                     "kernel_ms_p10_p90": [float(np.percentile(scan_ms, 10)), float(np.percentile(scan_ms, 90))] if scan_ms else None,
                     "algorithmic_bytes_per_launch": ALGO_BYTES_PER_DECISION * n,
                     "resolve_kernel_ms": float(np.mean(res_ms)) if res_ms else None,
                     "frac_of_measured_copy_peak_6290": (achieved / 6290.0) if achieved else None,
                     "whole_step_achieved_GBps": ALGO_BYTES_PER_DECISION * n / (gpu_ms / a.steps * 1e-3) / 1e9,
                     "stream_probe": probe, "beyond_infinity_cache": cold},
        "stats_last_step": st,
    }
    if not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, a.cpu_sample)
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    os.dup2(saved_stdout, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
