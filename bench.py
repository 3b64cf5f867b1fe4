#!/usr/bin/env python3
"""bench.py — placement decisions/sec + achieved HBM GB/s of the whole-table solve.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
A "step" is one whole-table pass of the placement solver (every row of the table gets a decision), inputs resident in HBM.

  N = 1   BASELINE.json config 3 (the configuration the metric is quoted on): 10 M objects x 1 024 nodes, Zipf(1.1) loads.
          value = a pipelined stream of COMMITTED ticks (rio_gp_tick_async: k_scan + k_resolve (+ fix-up) + commit), starting
          from the cold table; once the stream is quiet (nothing left to fix, nothing changed) the ticks' scans are chained over
          two streams, wave range by wave range, and k_resolve runs beside the next scan (config.step).  The same line carries the un-committed cold re-solves (round 2's headline, named as such), the
          synchronous dependent tick, config 2, config 4 on one GPU, the config-5 churn stream (synchronous + pipelined,
          per-kernel spans, parity of the whole 110-tick stream against the oracle chain) and — the headline ticks never cut or
          spill — the solve in which CAPACITY BINDS: `config3_contended` (0.72 x the capacities) and `config3_skew` (Lomax(1.1)
          affinities), one committed cold solve per step, parity at 10 M rows in the run (binding_record).
  N > 1   BASELINE.json config 4 as north_star states it: ONE table of 100 M objects x 4 096 nodes, rows sharded over
          the N ranks (12.5 M rows per GPU at N = 8), "scaling": "strong"; the weak-scaled config 3 (10 M rows per
          GPU of one N x 10 M-row table) is measured in the same run and reported under "weak_config3".  value = a stream
          of COMMITTED ticks of that table (rio_gp_shard_tick_async over the peer-to-peer windows, ShardedSolver.tick over
          the collective rungs) — the same quantity the N = 1 line carries for config 4 on one GPU.
  every N `scaling_points` = {strong_config4_committed_tick, weak_config3_committed_tick}: the two curves' points under
          the same keys and the same definition strings at every N (the un-committed re-solves rounds 2-3 reported
          at N > 1 are kept under `cold_resolve_uncommitted`).

  value        = decisions of all ranks / max-over-ranks wall time of the K steps; the timed region ends with the
                 library's own wait (rio_gp_tick_wait into a preallocated array), the counters become dictionaries afterwards
  parity       = the solved assignment column, the per-node `used` vector and the counters compared bit for bit with
                 the CPU oracle's solve of the same table, in this very run (exit code 3 on a mismatch); N = 1:
                 parity.against_reference_port = the same column against the map the cpu_baseline leg's string-level
                 restatement of LocalObjectPlacement + get_or_create_placement built, read back object by object
  roofline     = k_scan (the streaming kernel, >85 % of a step): algorithmic 16 B/decision (read
                 cur+load+aff, write assign) / its per-launch HIP-event time, on the cold table; `traffic` = HBM bytes per
                 launch from rocprofv3 PMC passes run by this script; frac_dram_bound / frac_committed_tick /
                 frac_dependent_tick = the same ratio for the whole step beyond the Infinity Cache and for whole ticks (a chained
                 tick costs less than k_scan's own launch duration: the ramp-up and the tail of consecutive scans overlap — the
                 roofline's kernel_ms is the plain k_scan alone on one stream, which is what a rocprofv3 trace shows for it)
  cpu_baseline = the CPU oracle port of the reference's per-object path, on a bounded sample: 1 thread, and as many threads as
                 the process may use CPUs (the cgroup's quota: 16 on this pool, where 256 hardware threads are visible); `value`
                 is the faster of the two, both are in the record
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

# the host driver of these boxes supports dmabuf IPC only: without this the peer-to-peer windows (hipIpcGetMemHandle) and RCCL
# fail across processes.  Set before any HIP runtime comes up; an operator's own value wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

ALGO_BYTES_PER_DECISION = 16
# ONE definition of the quantity every line reports, whatever N: the strings below are copied verbatim into the N = 1 and the
# N > 1 lines (`scaling_points`), so that a 1/2/4/8 curve is built from like quantities
DEF_TICK = ("committed ticks (solve + fix-up if needed + commit, each tick consuming the previous tick's table), a stream that starts "
            "from the cold table (tick 1: every object claims its requester); the timed ticks follow the warm-ups: every row gets "
            "its decision every tick and is kept; decisions/s = rows of the WHOLE table x ticks / max-over-ranks wall time")
DEF_STRONG = "BASELINE config 4, strong scaling: ONE table of 100 M objects x 4 096 nodes, rows sharded contiguously over the N ranks; " + DEF_TICK
DEF_WEAK = "BASELINE config 3, weak scaling: 10 M objects x 1 024 nodes PER GPU (one N x 10 M-row table, capacities from the global load); " + DEF_TICK
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
C4_ROWS, C4_NODES = 100_000_000, 4096


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None,
                    help="default: c3 at N=1, c4 (ONE 100 M x 4 096 table split over the ranks) at N>1 | c3 | c3w | c2 | "
                         "c4 | c5 (config 5: churn ticks, N=1 only)")
    ap.add_argument("--objects", type=int, default=0, help="override rows per GPU (weak-scaled workloads)")
    ap.add_argument("--total-objects", type=int, default=0, help="override the TOTAL row count of the strong-scaled c4 table")
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
                    help="N=1 only: run the row-sharded path (group of one rank) to price its extra kernels and launches")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000)
    ap.add_argument("--no-cold", action="store_true",
                    help="skip the beyond-the-Infinity-Cache data point (the same kernel over a 4x table, 640 MB of columns)")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run comparison with the CPU oracle")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 PMC passes (roofline.traffic falls back to profiles/)")
    ap.add_argument("--no-c4", action="store_true", help="N=1: skip the config-4-on-one-GPU data point (100 M x 4 096)")
    ap.add_argument("--no-c2", action="store_true", help="N=1: skip the config-2 record (1 M x 256)")
    ap.add_argument("--no-binding", action="store_true",
                    help="N=1: skip the two records of the solve in which capacity binds (config3_contended, config3_skew)")
    ap.add_argument("--no-c5", action="store_true", help="N=1: skip the config-5 record (10 %% churn per tick, 110 ticks + oracle replay)")
    ap.add_argument("--no-weak", action="store_true", help="N>1: skip the weak-scaled config-3 second measurement")
    ap.add_argument("--no-sharded-churn", action="store_true", help="N>1: skip the committed / churn tick streams of the sharded table")
    import glob
    newest = (sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_traffic.json"))) or [os.path.join(ROOT, "profiles", "round5_traffic.json")])[-1]
    ap.add_argument("--traffic-json", default=newest,
                    help="fallback for roofline.traffic when the in-run PMC passes are skipped or fail (default: the newest profiles/round*_traffic.json)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ CPU baseline

def usable_cpus():
    """CPUs this process may keep busy at once: the affinity mask, cut by a cgroup CPU quota (cpu.max / cfs_quota_us).  The GPU
    boxes of the pool show 256 hardware threads and grant 16 CPUs: more threads than that are throttled as a group."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, -(-q // per)))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(cfg, sample, array_oracle_seconds=None, gpu_cold_column=None):
    """Time the oracle port of the reference's per-object path on this box's host cores (bounded).  The map the
    single-thread leg builds is not thrown away: every object is looked up again and the answers are compared with the
    column the GPU solved from the same cold table (returned as the second value: parity.against_reference_port)."""
    import pyoracle
    n = min(sample, cfg["n"])
    aff = np.ascontiguousarray(cfg["aff"][:n])
    cores = usable_cpus()   # the multi-thread leg runs on what the container may use, not on every thread it can see
    t1, port_nodes = pyoracle.policy_readback(n, cfg["m"], aff)                      # cold: miss -> first touch -> update
    against_port = None
    if gpu_cold_column is not None:
        eq = bool(np.array_equal(port_nodes, gpu_cold_column[:n]))
        against_port = {"rows": int(n), "equal": eq,
                        "against": "the string-level restatement of LocalObjectPlacement + Service::get_or_create_placement "
                                   "(oracle/local_placement_oracle.cpp; local.rs:22-49, service.rs:193-254): the map the "
                                   "cpu_baseline leg built with one call per object (requester = the object's affinity node, "
                                   "every member active), read back with one lookup per object, against the GPU's solve of "
                                   "the same cold table (capacity 1.25x: no node is cut, so the solver IS the reference policy)"}
        if not eq:
            bad = np.flatnonzero(port_nodes != gpu_cold_column[:n])
            against_port["first_mismatches"] = [[int(i), int(gpu_cold_column[i]), int(port_nodes[i])] for i in bad[:5]]
    nT = min(n, 1_000_000)
    tT, _ = pyoracle.bench_policy(nT, cfg["m"], aff[:nT], threads=cores)             # all cores on ONE shared map
    nW = min(n, 20_000)
    tW, _ = pyoracle.bench_policy(nW, cfg["m"], aff[:nW], threads=1, warm=True)      # warm: sticky hit + O(M) is_active
    v1, vT = n / t1, nT / tT
    best_v, best_c = (v1, 1) if v1 >= vT else (vT, cores)
    if array_oracle_seconds is None:  # "best reasonable CPU": the array solver (same algorithm as the GPU), one thread
        t0 = time.perf_counter()
        pyoracle.tick(cfg["cur"], cfg["load"], cfg["aff"], cfg["cap"], cfg["alive"])
        array_oracle_seconds = time.perf_counter() - t0
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return against_port, {
        "value": best_v, "unit": "decisions/s", "cores": best_c, "kind": "port",
        "sample": "cold get_or_create_placement per object (service.rs:193-254 restated in C++: string keys, "
                  "unordered_map behind a shared_mutex, %d-member LocalStorage) over the first %d objects of the "
                  "workload on 1 thread, and over the first %d objects on %d threads (the CPUs this process may use) sharing one map; "
                  "value = the faster of the two" % (cfg["m"], n, nT, cores),
        "value_1thread": v1, "value_allcores": vT, "host_cores": cores, "hardware_threads_visible": os.cpu_count(), "cpu_model": model,
        "warm_sticky_hits_1thread": {"value": nW / tW, "unit": "decisions/s", "rows": nW,
                                     "note": "every call hits and pays the O(M) is_active member scan (cluster/storage/mod.rs:95-110)"},
        "array_oracle_1thread": {"value": cfg["n"] / array_oracle_seconds, "unit": "decisions/s", "rows": cfg["n"],
                                 "note": "oracle/placement_oracle.c orc_tick: the GPU algorithm run sequentially on dense arrays "
                                         "(the same run the parity check compares against)"},
    }


# ------------------------------------------------------------------------------------------------ parity (in-run)

def parity_single(g, cfg, rounds=2):
    """One more solve of the benchmark table, compared with the CPU oracle bit for bit: assignment column, counters,
    and (after the commit) the per-node `used` vector.  Returns (record, oracle_seconds)."""
    import pyoracle
    t0 = time.perf_counter()
    want, used, ost = pyoracle.tick(cfg["cur"], cfg["load"], cfg["aff"], cfg["cap"], cfg["alive"], rounds)
    t_orc = time.perf_counter() - t0
    st = g.solve()
    got = g.get_solved()
    eq_a = bool(np.array_equal(got, want))
    g.commit()
    eq_u = bool(np.array_equal(g.get_nodes()[2], used))
    eq_s = st == ost
    rec = {"checked_rows": int(cfg["n"]), "equal": eq_a and eq_u and eq_s, "assign_equal": eq_a, "used_equal": eq_u,
           "stats_equal": eq_s, "against": "oracle/placement_oracle.c orc_tick on the same table, same run",
           "oracle_seconds": t_orc}
    if not eq_a:
        bad = np.flatnonzero(got != want)
        rec["first_mismatches"] = [[int(i), int(got[i]), int(want[i])] for i in bad[:5]]
    if not eq_s:
        rec["stats_gpu"], rec["stats_oracle"] = st, ost
    rec["_solved_column"] = got   # (popped by the caller: compared with the reference port's map)
    return rec, t_orc


def parity_sharded(dist, backend, g, sol, workload, n_total, m, bounds, rank, world, cap):
    """Every rank's solved rows and its (global) `used` vector against the oracle's solve of the WHOLE table, computed
    on rank 0 from the same generator."""
    import torch
    import synth
    st = sol.solve()
    mine = g.get_solved()
    sol.commit()
    used = g.get_nodes()[2]
    nmax = max(bounds[r + 1] - bounds[r] for r in range(world))
    dev = "cuda" if backend == "nccl" else "cpu"
    pad = np.full(nmax, 0xFFFFFFFF, np.uint32)
    pad[:len(mine)] = mine
    t_rows = torch.from_numpy(pad.astype(np.int64)).to(dev)
    t_used = torch.from_numpy(used.astype(np.int64)).to(dev)
    rows_all = [torch.empty_like(t_rows) for _ in range(world)]
    used_all = [torch.empty_like(t_used) for _ in range(world)]
    dist.all_gather(rows_all, t_rows)
    dist.all_gather(used_all, t_used)
    if rank != 0:
        return None
    import pyoracle
    glob = synth.config(workload, n_override=n_total, start=0)
    t0 = time.perf_counter()
    want, wused, ost = pyoracle.tick(glob["cur"], glob["load"], glob["aff"], cap, glob["alive"], 2)
    t_orc = time.perf_counter() - t0
    got = np.concatenate([rows_all[r].cpu().numpy().astype(np.uint32)[:bounds[r + 1] - bounds[r]] for r in range(world)])
    eq_a = bool(np.array_equal(got, want))
    eq_u = all(bool(np.array_equal(used_all[r].cpu().numpy().astype(np.uint64), wused)) for r in range(world))
    keys = ("n_objects", "kept", "evicted", "claimed", "spilled", "unplaced", "load_kept", "load_claimed", "load_spilled",
            "load_unplaced", "cut_nodes", "slow_path", "rounds_run")
    eq_s = all(st[k] == ost[k] for k in keys)
    rec = {"checked_rows": int(n_total), "equal": eq_a and eq_u and eq_s, "assign_equal": eq_a, "used_equal_on_every_rank": eq_u,
           "stats_equal": eq_s, "against": "oracle/placement_oracle.c orc_tick on the WHOLE %d-row table (rank 0), every rank's "
                                            "shard gathered" % n_total, "oracle_seconds": t_orc}
    if not eq_s:
        rec["stats_gpu"], rec["stats_oracle"] = st, ost
    return rec


def sharded_ticks(a, dist, torch, g, sol, kind, workload, n_total, m, bounds, rank, world, cap, ticks=10, phase=None):
    """Committed ticks of the row-sharded table, after the parity step left it committed (warm): (i) churn-free ticks
    (every row kept: scan + exchange + verdict + commit), (ii) BASELINE config 5 on the sharded table: per tick a liveness
    push (10 % of the nodes down, a different 10 % each tick) and one committed tick with its fix-up exchanges —
    synchronous (`ShardedSolver.tick`: every phase returns to the host) and, over the peer-to-peer windows, asynchronous
    (`rio_gp_shard_tick_async`: the whole chain enqueued, guarded on the device, counters read at the end).  Max over ranks.
    The final table is checked against the oracle chained over the same ticks and masks (rank 0, whole table) when that
    takes seconds, not minutes."""
    import synth
    dev = "cuda" if a.backend == "nccl" else "cpu"
    schedule = []                       # the liveness mask every executed tick ran under (None: unchanged)
    masks = [synth.churn_mask(m, k + 1) for k in range(2 * ticks + 4)]

    class PhaseFailed(Exception):
        pass

    def timed(step, k, warm=0):
        """`warm` untimed calls, then k timed ones.  Every rank ends the phase with exactly ONE collective that carries its time
        and whether it failed, so a rank whose exchange timed out cannot leave the others waiting in a collective of their own:
        all ranks give the phase up together."""
        dt, fail, why, out = 0.0, 0.0, "", None
        try:
            fin = getattr(step, "finish", None)
            for i in range(warm):
                step(i)
            if warm and fin is not None:
                fin()
            g.sync()
            t0 = time.perf_counter()
            out = [step(warm + i) for i in range(k)]
            out = [x for x in out if x is not None]
            if fin is not None:
                out = fin()
            g.sync()
            dt = time.perf_counter() - t0
        except Exception as e:
            fail, why = 1.0, repr(e)[:200]
        t = torch.tensor([dt, fail], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if float(t[1].item()):
            raise PhaseFailed(why or "failed on another rank")
        return float(t[0].item()), out

    def keep_sync(i):
        schedule.append(None)
        return sol.tick()

    def churn_sync(i):
        g.set_alive_all(masks[i]); schedule.append(masks[i])
        return sol.tick()

    def rec_of(dt, sts, k, step):
        return {"ms_per_tick": dt / k * 1e3, "value": n_total * k / dt, "unit": "decisions/s",
                "objects_moved_per_s": sum(x["claimed"] + x["spilled"] for x in sts) / dt,
                "slow_path_ticks": int(sum(x["slow_path"] for x in sts)), "stats_last_tick": sts[-1], "step": step}

    phase = phase if phase is not None else []
    phase.append("committed ticks, synchronous")
    dt, sts = timed(keep_sync, ticks, warm=1)
    rec = {"ticks": ticks, "committed_tick_no_churn": rec_of(dt, sts, ticks,
           "ShardedSolver.tick: solve_async + solve_wait (verdict, global counters) + commit, the host in the loop")}
    phase.append("churn ticks, synchronous")
    dt, sts = timed(churn_sync, ticks, warm=2)
    rec["churn"] = rec_of(dt, sts, ticks,
                          "rio_gp_set_alive_all + ShardedSolver.tick: scan, exchange, verdict, cut + export, exchange, merge, then per "
                          "water-fill round spill + export, exchange, merge; counters; commit — synchronous, every phase returns to the host")
    last = sts[-1]
    if kind == "p2p":
        class KeepAsync:
            def __call__(self, i):
                schedule.append(None); sol.tick_async()
            def finish(self):
                return sol.tick_wait()

        class ChurnAsync:
            def __call__(self, i):
                mk = masks[ticks + 2 + i]   # (continues the synchronous stream's masks)
                g.set_alive_all(mk); schedule.append(mk); sol.tick_async()
            def finish(self):
                return sol.tick_wait()
        phase.append("committed ticks, asynchronous")
        dt, sts = timed(KeepAsync(), ticks, warm=1)
        rec["committed_tick_no_churn_async"] = rec_of(dt, sts, ticks,
            "rio_gp_shard_tick_async: scan, one-launch exchange, the guarded fix-up chain (nothing to do), commit; nothing waits "
            "on the host, counters read at the end")
        phase.append("churn ticks, asynchronous")
        dt, sts = timed(ChurnAsync(), ticks)
        rec["churn_async"] = rec_of(dt, sts, ticks,
            "rio_gp_set_alive_all + rio_gp_shard_tick_async: the same chain with its cut pass, water-fill rounds and three "
            "peer-to-peer exchanges per tick, enqueued back to back")
        last = sts[-1]
    phase.append("parity of the tick streams")
    if a.no_parity or n_total * (len(schedule) + 2) > 600_000_000:
        rec["parity"] = None
        return rec
    mine, used = g.get_assign(), g.get_nodes()[2]
    nmax = max(bounds[r + 1] - bounds[r] for r in range(world))
    pad = np.full(nmax, 0xFFFFFFFF, np.uint32)
    pad[:len(mine)] = mine
    t_rows, t_used = torch.from_numpy(pad.astype(np.int64)).to(dev), torch.from_numpy(used.astype(np.int64)).to(dev)
    rows_all, used_all = [torch.empty_like(t_rows) for _ in range(world)], [torch.empty_like(t_used) for _ in range(world)]
    dist.all_gather(rows_all, t_rows)
    dist.all_gather(used_all, t_used)
    if rank != 0:
        return rec
    import pyoracle
    glob = synth.config(workload, n_override=n_total, start=0)
    t0 = time.perf_counter()
    alive = glob["alive"]
    ref, wused, ost = pyoracle.tick(glob["cur"], glob["load"], glob["aff"], cap, alive, 2)   # the parity step's commit
    for mk in schedule:
        alive = alive if mk is None else mk
        ref, wused, ost = pyoracle.tick(ref, glob["load"], glob["aff"], cap, alive, 2)
    got = np.concatenate([rows_all[r].cpu().numpy().astype(np.uint32)[:bounds[r + 1] - bounds[r]] for r in range(world)])
    keys = ("n_objects", "kept", "evicted", "claimed", "spilled", "unplaced", "load_kept", "load_claimed", "load_spilled",
            "load_unplaced", "cut_nodes", "slow_path", "rounds_run")
    eq_a = bool(np.array_equal(got, ref))
    eq_u = all(bool(np.array_equal(used_all[r].cpu().numpy().astype(np.uint64), wused)) for r in range(world))
    eq_s = all(last[k] == ost[k] for k in keys)
    rec["parity"] = {"checked_rows": int(n_total), "ticks_replayed": len(schedule) + 1, "equal": eq_a and eq_u and eq_s,
                     "assign_equal": eq_a, "used_equal_on_every_rank": eq_u, "stats_equal": eq_s,
                     "against": "oracle/placement_oracle.c orc_tick chained over the same ticks and liveness masks on the WHOLE "
                                "table (rank 0), after the last stream", "oracle_seconds": time.perf_counter() - t0}
    if not eq_s:
        rec["parity"]["stats_gpu"], rec["parity"]["stats_oracle"] = last, ost
    return rec


# ------------------------------------------------------------------------------------------------ PMC traffic (in-run)

def pmc_traffic_in_run(n_rows, timeout=240):
    """HBM bytes per k_scan launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — they do not fit one pass,
    MI355X_MICROARCH.md §rocprofv3) over tools/pmc_workload.py, calibrated on the stream probes of the same passes
    (tools/pmc_traffic.py).  Runs as child processes next to this one; returns (doc, source) or (None, reason)."""
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    import pmc_traffic
    work = tempfile.mkdtemp(prefix="rio_pmc_")
    env = dict(os.environ, TMPDIR=work)
    dirs = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, counter.lower())
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.join(ROOT, "tools", "pmc_workload.py"), "c3", str(n_rows)]
            r = subprocess.run(cmd, cwd=work, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, r.stderr.decode(errors="replace")[-300:])
            dirs[counter] = d
        doc = pmc_traffic.compute(dirs["FETCH_SIZE"], dirs["WRITE_SIZE"], n_rows)
        if "hbm_bytes_per_launch" not in doc:
            return None, "PMC passes produced no k_scan / probe counters"
        return doc, ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate --kernel-trace "
                     "passes over tools/pmc_workload.py: 30 solves of the same table + stream probes of known traffic that "
                     "calibrate both counters; MI355X_MICROARCH.md §HBM)")
    except Exception as e:  # measurement aid: never fails the bench
        return None, "PMC passes failed: %r" % (e,)
    finally:
        shutil.rmtree(work, ignore_errors=True)


# ------------------------------------------------------------------------------------------------ config 5 (churn)

def churn_record(a, g, cfg, rio_gp, local_rank, steps=None, warmup=None):
    """BASELINE.json config 5: the config-3 table, warm; every step = one liveness push (10 % of the nodes down, the
    previous casualties back) + one committed whole-table tick (evict + re-place through the fix-up path).  Two timed
    streams over the same masks: synchronous ticks (the host reads every tick's counters before the next push) and pipelined
    ticks (rio_gp_tick_async: the counters are read back later; results identical).  Parity: the oracle replays the whole
    stream at full size.  Returns the record (also used as the `config5_churn` entry of the default line)."""
    import pyoracle
    import synth
    steps = steps or a.steps
    warmup = a.warmup if warmup is None else warmup
    n, m = cfg["n"], cfg["m"]
    warm = synth.warm_assign(n, m)
    total = warmup + steps
    masks = [synth.churn_mask(m, 2 + k) for k in range(total)]

    def reset():
        g.set_alive_all(np.ones(m, np.uint8))
        g.set_assign(warm)
        g.tick()

    reset()
    for k in range(warmup):
        g.set_alive_all(masks[k])
        g.tick()
    g.sync()
    moved = slow = 0
    t0 = time.perf_counter()
    for k in range(steps):
        g.set_alive_all(masks[warmup + k])
        st = g.tick()
        moved += st["claimed"] + st["spilled"]
        slow += st["slow_path"]
    dt = time.perf_counter() - t0
    final_sync = g.get_assign()
    used_sync = g.get_nodes()[2]
    parity = None
    if not a.no_parity:
        ref = warm.copy()
        t1 = time.perf_counter()
        ref, used, ost = pyoracle.tick(ref, cfg["load"], cfg["aff"], cfg["cap"], np.ones(m, np.uint8), 2)
        for k in range(total):
            ref, used, ost = pyoracle.tick(ref, cfg["load"], cfg["aff"], cfg["cap"], masks[k], 2)
        parity = {"checked_rows": int(n), "ticks_replayed": total + 1, "equal": bool(np.array_equal(final_sync, ref)) and
                  bool(np.array_equal(used_sync, used)) and st == ost,
                  "against": "oracle/placement_oracle.c orc_tick chained over the same %d liveness masks; assignment column, "
                             "`used` and the last tick's counters after the final tick" % total,
                  "oracle_seconds": time.perf_counter() - t1}
    reset()
    for k in range(warmup):
        g.set_alive_all(masks[k])
        g.tick_async()
    g.tick_wait()
    g.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        g.set_alive_all(masks[warmup + k])
        g.tick_async()
    sts = g.tick_wait()
    dtp = time.perf_counter() - t0
    frac = lambda sec: ALGO_BYTES_PER_DECISION * n * steps / sec / 1e9 / HBM_PEAK_GBPS
    # SURVEY.md section 8d's own accounting for config 5: the clean_server scan (4 B per object) + 4 B per evicted row + a placement
    # decision (16 B) per re-placed row — what the reference's clean_server + per-request path would move for the same
    # effect; ~60 MB per tick at 10 M rows where the full-tick accounting above counts 160 MB
    ev_rows = sum(x["evicted"] for x in sts)
    re_rows = sum(x["claimed"] + x["spilled"] + x["unplaced"] for x in sts)
    survey_bytes = 4 * n * steps + 4 * ev_rows + 16 * re_rows
    piped = {"ms_per_tick": dtp / steps * 1e3, "value": n * steps / dtp, "unit": "decisions/s", "frac_of_roofline": frac(dtp),
             "frac_survey_accounting": survey_bytes / dtp / 1e9 / HBM_PEAK_GBPS,
             "survey_accounting": {"bytes_per_tick": survey_bytes / steps, "evicted_rows_per_tick": ev_rows / steps,
                                   "replaced_rows_per_tick": re_rows / steps,
                                   "definition": "SURVEY.md 8d, config 5: 4 B/object scan + 4 B per evicted row + 16 B per re-placed row"},
             "equal_to_synchronous_stream": bool(np.array_equal(g.get_assign(), final_sync)) and
             bool(np.array_equal(g.get_nodes()[2], used_sync)) and sts[-1] == st,
             "step": "rio_gp_set_alive_all + rio_gp_tick_async: nothing waits on the host between ticks, every tick's "
                     "counters are read afterwards (rio_gp_tick_wait)"}
    spans = None
    try:  # where a tick's time goes, measured by the kernels themselves (lab build: wall_clock64 phase traces)
        gl = rio_gp.LabPlacement(n, m, device=local_rank)
        gl.set_nodes(cfg["cap"], cfg["alive"])
        gl.set_objects(n, cfg["load"], cfg["aff"])
        gl.set_assign(warm)
        gl.tick()
        for k in range(60):      # the stream's steady state: the first ticks after the warm table are ~10 us cheaper
            gl.set_alive_all(masks[k % total])
            gl.tick_async()
        gl.tick_wait()
        gl.ktrace(True)
        for k in range(60, 66):
            gl.set_alive_all(masks[k % total])
            gl.tick()
        names = {3: "k_inc_scan", 5: "k_rebal", 0: "k_resolve<SEARCH>", 1: "k_fill round 0", 2: "k_fill round 1"}
        tabs = {t: gl.ktrace(True, t).astype(np.int64) for t in names}
        gl.ktrace(False)
        gl.close()
        base = min(int(t[t[:, 0] > 0][:, 0].min()) for t in tabs.values() if (t[:, 0] > 0).any())
        spans = {names[k]: {"first_start_us": round((int(t[t[:, 0] > 0][:, 0].min()) - base) / 100.0, 1),
                            "last_end_us": round((int(t[:, 7].max()) - base) / 100.0, 1)}
                 for k, t in tabs.items() if (t[:, 0] > 0).any()}
    except Exception as e:  # measurement aid only
        spans = {"error": repr(e)}
    return {
        "workload": "config 5: %d objects x %d nodes, Zipf(1.1) load, cap 1.25x, warm; per tick 10 %% of the nodes flip and one "
                    "committed tick evicts and re-places their objects (~1 M rows, hundreds of cut nodes)" % (n, m),
        "ticks": steps, "warmup": warmup, "slow_path_ticks": slow,
        "synchronous": {"ms_per_tick": dt / steps * 1e3, "value": n * steps / dt, "unit": "decisions/s", "frac_of_roofline": frac(dt),
                        "frac_survey_accounting": survey_bytes / dt / 1e9 / HBM_PEAK_GBPS,
                        "step": "rio_gp_set_alive_all + rio_gp_tick (the host reads every tick's counters)"},
        "pipelined": piped, "objects_moved_per_s": moved / dt, "stats_last_tick": st, "parity": parity,
        "kernel_spans_on_device_us": spans,
        "launches_per_tick": "k_inc_scan (cur/load/aff streamed, the column updated in place, no histogram; reads the pushed liveness "
                             "bitmap from mapped pinned memory) + k_rebal (pending rows dealt out evenly, per-block histograms) + "
                             "k_resolve<SEARCH> + k_fill (round 0) + k_fill (round 1)",
    }


def binding_record(a, cfg, rio_gp, local_rank, which, reps=30, warmup=4):
    """The solve in which CAPACITY BINDS, on the driver's line (north_star: "min-cost bin-packing placements for all pending
    activations at once").  The headline ticks of config 3 never cut or spill; these two tables do:
      contended  config 3 with 0.72 x its capacities (0.9 x the load fits): ~1 020 of the 1 024 nodes are cut, ~1 M rows
                 go on to the water-fill, most of them stay unplaced;
      skew       config 3 with Lomax(1.1) affinities: a few servers asked for by most objects, ~94 % of the rows water-filled.
    A step = ONE committed whole-table solve of the COLD table (rio_gp_tick: scan, resolve, exact cuts, water-fill rounds,
    commit, the host reads the counters); the table is put back to all-NONE between two steps by a device-to-device copy that
    is NOT timed.  Second figure: round 5's method (the same cold table re-solved back to back, rio_gp_solve, never committed).
    Parity: the committed column, `used` and the counters against orc_tick of the same table at full size, in this run."""
    import pyoracle
    import synth
    import hipbuf
    n, m = cfg["n"], cfg["m"]
    if which == "contended":
        cap, aff = synth.contended_cap(cfg), cfg["aff"]
        what = "config 3 (%d objects x %d nodes, Zipf(1.1) load) with 0.72 x the capacities: 0.9 x the total load fits" % (n, m)
    else:
        cap, aff = cfg["cap"], synth.skew_affinity(n, m)
        what = "config 3 (%d objects x %d nodes, Zipf(1.1) load, cap 1.25x) with Lomax(1.1) affinities (synth.skew_affinity)" % (n, m)
    g = rio_gp.GpuPlacement(n, m, device=local_rank)
    g.set_nodes(cap, cfg["alive"])
    g.set_objects(n, cfg["load"], aff)
    cold = hipbuf.DevBuf(cfg["cur"])
    st = rio_gp.Stats()
    wall = []
    for k in range(warmup + reps):
        g.set_assign_dev(n, cold.ptr)     # (waits for the copy: nothing of it is inside the timed call)
        t0 = time.perf_counter()
        g.tick_struct(st)                 # solve + fix-up + commit; returns with the counters
        if k >= warmup:
            wall.append(time.perf_counter() - t0)
    last = st.as_dict()
    parity = None
    if not a.no_parity:
        t1 = time.perf_counter()
        want, used, ost = pyoracle.tick(cfg["cur"], cfg["load"], aff, cap, cfg["alive"], 2)
        eq_a = bool(np.array_equal(g.get_assign(), want))
        eq_u = bool(np.array_equal(g.get_nodes()[2], used))
        parity = {"checked_rows": int(n), "equal": eq_a and eq_u and last == ost, "assign_equal": eq_a, "used_equal": eq_u,
                  "stats_equal": last == ost, "against": "oracle/placement_oracle.c orc_tick on the same table, same run "
                  "(the table as the last timed committed solve left it)", "oracle_seconds": time.perf_counter() - t1}
        if last != ost:
            parity["stats_gpu"], parity["stats_oracle"] = last, ost
    g.set_assign_dev(n, cold.ptr)
    for _ in range(warmup):
        g.solve()
    g.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ust = g.solve()
    unc = (time.perf_counter() - t0) / reps
    g.close()
    cold.free()
    sec = float(np.median(wall))
    fr = lambda s_: ALGO_BYTES_PER_DECISION * n / s_ / 1e9 / HBM_PEAK_GBPS
    return {"workload": what, "steps": reps, "warmup": warmup,
            "us_per_solve": sec * 1e6, "us_per_solve_p10_p90": [float(np.percentile(wall, 10)) * 1e6, float(np.percentile(wall, 90)) * 1e6],
            "value": n / sec, "unit": "decisions/s", "frac": fr(sec),
            "step": "rio_gp_tick on the cold table: ONE committed whole-table solve per step, synchronous (median of the steps; "
                    "the reset of the table between steps is not timed)",
            "slow_path": int(last["slow_path"]), "cut_nodes": int(last["cut_nodes"]), "stats": last,
            "uncommitted_back_to_back": {"us_per_solve": unc * 1e6, "frac": fr(unc), "equal_counters": ust == last,
                                         "step": "rio_gp_solve of the same cold table, back to back, never committed (round 5's method)"},
            "launches": ("k_scan + k_resolve + k_cut_apply (exact cuts applied + the undecided rows packed, one pass) + k_cut_settle + "
                         "k_fill x 2 over the packed rows (the route when <= 25 % of the rows are pending)" if which == "contended" else
                         "k_scan + k_resolve + k_cut_find + k_fill (round 0: re-mark, water-fill over the table) + k_fill (round 1) "
                         "(the route when most of the table is pending: packing would cost more than it saves)"),
            "parity": parity}


def bench_churn(a, g, cfg, saved_stdout, rio_gp, local_rank):
    """--workload c5: config 5 as the bench line of its own."""
    rec = churn_record(a, g, cfg, rio_gp, local_rank)
    n = cfg["n"]
    out = {
        "metric": "placement decisions/sec, 10M objects x 1 024 nodes with 10 % node-failure churn per tick",
        "value": rec["pipelined"]["value"], "unit": "decisions/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": rec["pipelined"]["ms_per_tick"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": rec["workload"], "step": rec["pipelined"]["step"], "slow_path_steps": rec["slow_path_ticks"]},
        "config5_churn": rec, "parity": rec["parity"],
        "roofline": {"bound": "hbm", "achieved": ALGO_BYTES_PER_DECISION * rec["pipelined"]["value"] / 1e9, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": rec["pipelined"]["frac_of_roofline"], "traffic": None,
                     "traffic_source": "not measured for this workload (profiles/ holds the per-kernel summaries of the fix-up path)",
                     "kernel": "whole committed tick (5 dependent launches; latency-bound, DESIGN.md section 5)"},
    }
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    print(json.dumps(out), flush=True)
    if rec["parity"] is not None and not rec["parity"]["equal"]:
        sys.exit(3)


# ------------------------------------------------------------------------------------------------ sharded set-up

def make_sharded_solver(a, dist, g, local_rank, rank, n_global, load_total):
    """Exchange ladder p2p -> native -> torch: a path that cannot be set up, or whose warm-up fails on ANY rank (agreed
    through an all-reduce), is dropped for the next one on EVERY rank."""
    import torch
    import sharded
    eng = sharded.HipShardEngine(g, local_rank)
    ladder = ["p2p", "native", "torch"]
    ladder = ladder[ladder.index(a.exchange):]
    tried = []
    for kind in ladder:
        ok, ex, why, sol = 1, None, "", None
        try:
            ex = {"p2p": sharded.P2PExchange, "native": sharded.NativeRcclExchange}[kind](eng) if kind != "torch" \
                else sharded.DistExchange()
            sol = sharded.ShardedSolver([eng], ex, spill_rounds=2, pipeline=(kind == "torch" and not a.no_pipeline))
            if kind == os.environ.get("RIO_GP_BENCH_FAIL_RUNG") and str(rank) == os.environ.get("RIO_GP_BENCH_FAIL_RANK", "0"):
                raise RuntimeError("injected failure of rung '%s' on rank %d (flow test)" % (kind, rank))
            for _ in range(max(a.warmup, 2)):
                sol.solve_async()
            st, n_slow = sol.solve_wait()
            # an exchange that delivers wrong records must not survive the warm-up: every row decided exactly once and
            # every unit of load accounted for, on the GLOBAL table
            if st["kept"] + st["claimed"] + st["spilled"] + st["unplaced"] != n_global or \
                    st["load_kept"] + st["load_claimed"] + st["load_spilled"] + st["load_unplaced"] != load_total:
                raise RuntimeError("exchange '%s' produced inconsistent global stats: %r" % (kind, st))
        except Exception as e:  # set-up failure raises on every rank; a warm-up failure may be local
            ok, why = 0, str(e)
        dev = "cuda" if a.backend == "nccl" else "cpu"
        flag = torch.tensor([ok], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        tried.append({"path": kind, "ok": bool(int(flag.item())), "why": why[:200]})
        if int(flag.item()) == 1:
            return sol, kind, tried
        print("rank %d: exchange '%s' dropped (%s)" % (rank, kind, why or "failed on another rank"), file=sys.stderr)
        if kind == "p2p" and ex is not None and hasattr(ex, "close"):
            ex.close()
    raise SystemExit("no exchange path could be set up")


def timed_steps(a, g, dist, torch, step, wait, wait_c=None):
    """K steps between two barriers (N > 1: dist.barrier) + device synchronisation on both sides.  The closing
    synchronisation of the timed region is the library's own wait — `wait_c`, one C call that waits for the handle's stream
    (the only stream with work on it) and copies the counters into a preallocated array — when the caller has one: nothing
    of the host's bookkeeping (Python dictionaries of K ticks' counters, a second wait on an event, torch's synchronise over
    streams that carry nothing) sits between the last step and the clock.  `wait` then only converts what `wait_c` fetched."""
    def barrier():
        if dist is not None:
            dist.barrier()
        g.sync()
        torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    g.timer_begin()
    for _ in range(a.steps):
        step()
    g.timer_stop()              # the closing event is recorded behind the last step; it is read after the clock
    if wait_c is not None:
        wait_c()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        st, n_slow = wait()
    else:
        st, n_slow = wait()
        barrier()
        dt = time.perf_counter() - t0
    gpu_ms = g.timer_end()
    if dist is not None:
        dev = "cuda" if a.backend == "nccl" else "cpu"
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, gpu_ms, st, n_slow


def global_cap(dist, torch, a, load, m):
    """cap[j] = ceil(1.25 * sum of the GLOBAL load / m), the same on every rank (set-up only, not the data path)."""
    dev = "cuda" if a.backend == "nccl" else "cpu"
    tot = torch.tensor([int(load.astype(np.uint64).sum())], device=dev, dtype=torch.int64)
    dist.all_reduce(tot)
    total = int(tot.item())
    return np.full(m, -((-total * 1250) // (1000 * m)), dtype=np.uint64), total


def run_sharded(a, dist, torch, rio_gp, synth, workload, rank, world, local_rank):
    """One sharded measurement: build this rank's shard, pick the exchange, time K steps, check parity."""
    strong = workload == "c4"
    if strong:
        n_total = a.total_objects or C4_ROWS
        import sharded
        bounds = sharded.shard_bounds(n_total, world)
        start, n_local = bounds[rank], bounds[rank + 1] - bounds[rank]
    else:
        n_local = a.objects or synth.DEFAULT_ROWS[workload]
        n_total = n_local * world
        bounds = [r * n_local for r in range(world + 1)]
        start = rank * n_local
    cfg = synth.config(workload, n_override=n_local, start=start)
    n, m = cfg["n"], cfg["m"]
    cfg["cap"], load_total = global_cap(dist, torch, a, cfg["load"], m)
    g = rio_gp.GpuPlacement(max(n, 1), m, device=local_rank)
    g.set_nodes(cfg["cap"], cfg["alive"])
    g.set_objects(n, cfg["load"], cfg["aff"])
    sol, kind, tried = make_sharded_solver(a, dist, g, local_rank, rank, n_total, load_total)
    none_col = np.full(n, 0xFFFFFFFF, np.uint32)
    # ---- (1) the headline: a stream of COMMITTED ticks of the sharded table, from the cold table (tick 1: every object claims
    #      its requester), W warm-ups then K timed ticks — the quantity the N = 1 line reports for the same table.  Over the
    #      peer-to-peer windows a tick is rio_gp_shard_tick_async (nothing waits on the host; at most 48 in flight before
    #      their records are fetched); over the collective rungs it is ShardedSolver.tick (the host sequences the exchanges).
    asynchronous = kind == "p2p"
    inflight, fetched = [0], []

    def tick_step():
        if asynchronous:
            if inflight[0] == 48:
                fetched.append(sol.tick_wait_local()); inflight[0] = 0
            sol.tick_async(); inflight[0] += 1
        else:
            fetched.append(sol.tick())

    def tick_fetch():
        if asynchronous and inflight[0]:
            fetched.append(sol.tick_wait_local()); inflight[0] = 0

    def tick_stats():
        sts = []
        for x in fetched:
            sts.extend(sol.tick_reduce(x) if asynchronous else [x])
        del fetched[:]
        return sts, sum(t["slow_path"] for t in sts)

    for _ in range(a.warmup):
        tick_step()
    tick_fetch()
    tick_stats()
    dt, gpu_ms, sts, n_slow = timed_steps(a, g, dist, torch, tick_step, tick_stats, wait_c=tick_fetch)
    st = sts[-1]
    assert st["kept"] + st["claimed"] + st["spilled"] + st["unplaced"] == n_total
    # ---- (2) round 2's quantity, kept under its own name: the cold table re-solved back to back WITHOUT committing
    g.set_assign(none_col)
    cdt, cgpu_ms, cst, cslow = timed_steps(a, g, dist, torch, sol.solve_async, sol.solve_wait)
    parity = None
    if not a.no_parity:
        parity = parity_sharded(dist, a.backend, g, sol, workload, n_total, m, bounds, rank, world, cfg["cap"])
    ticks = None
    if not a.no_sharded_churn:
        if a.no_parity:
            sol.tick()   # the committed (warm) table the tick streams start from
        phase = []
        try:
            ticks = sharded_ticks(a, dist, torch, g, sol, kind, workload, n_total, m, bounds, rank, world, cfg["cap"],
                                  ticks=max(2, min(a.steps, 10)), phase=phase)
        except Exception as e:  # a second measurement: it must not take the line down with it
            ticks = {"error": repr(e)[:300], "during": phase[-1] if phase else None}
    # who ran where, and what RCCL was asked for (control plane: torch.distributed's group when its backend is nccl = RCCL;
    # data path: the library's own communicator of the `native` rung — none over the peer-to-peer windows)
    import sharded as _sh
    info = {"rank": rank, "device": int(local_rank), "exchange": kind,
            "native_rung_comm_ranks": int(_sh._lib().rio_gp_shard_comm_ranks(g.handle))}
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        info["device_name"] = pr.name
        info["pci_bus_id"] = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
    except Exception:
        pass
    infos = [None] * world
    dist.all_gather_object(infos, info)
    if not a.same_device and len({(x.get("pci_bus_id"), x["device"]) for x in infos}) != world:
        raise SystemExit("bench.py: the %d ranks do not run on %d distinct devices: %r" % (world, world, infos))
    rec = {"value": n_total * a.steps / dt, "ms_per_step": dt / a.steps * 1e3, "gpu_ms_per_step_events": gpu_ms / a.steps,
           "rows_total": n_total, "rows_this_rank": n, "nodes": m, "exchange": kind, "exchange_ladder": tried,
           "slow_path_steps": n_slow, "stats_last_step": st, "parity": parity, "committed_ticks": ticks,
           "asynchronous": asynchronous, "ranks": infos,
           "rccl": {"control_plane_backend": a.backend,
                    "control_plane_communicators": 1 if a.backend == "nccl" else 0,
                    "data_path_communicators": 1 if kind == "native" else 0,
                    "data_path_comm_ranks": [x["native_rung_comm_ranks"] for x in infos],
                    "note": "p2p rung: no collective on the data path (stores into the peers' IPC-mapped windows); native rung: ONE "
                            "ncclComm of the library's own, world ranks; torch rung: the control plane's group carries the records"},
           "cold_resolve_uncommitted": {"value": n_total * a.steps / cdt, "unit": "decisions/s", "ms_per_step": cdt / a.steps * 1e3,
                                        "gpu_ms_per_step_events": cgpu_ms / a.steps, "slow_path_steps": cslow,
                                        "stats_last_step": cst,
                                        "note": "rio_gp_shard_solve_async of the SAME cold table back to back, never committed "
                                                "(what rounds 2-3 reported as `value` at N > 1)"},
           "whole_step_achieved_GBps": ALGO_BYTES_PER_DECISION * n / (gpu_ms / a.steps * 1e-3) / 1e9}
    g.close()
    return rec


def peer_matrix(torch, world, same_device):
    """hipDeviceCanAccessPeer for every pair of the ranks' devices (what the peer-to-peer windows need)."""
    try:
        if same_device:
            return "all ranks on device 0 (--same-device flow test)"
        nd = torch.cuda.device_count()
        return [[int(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(min(nd, world))] for i in range(min(nd, world))]
    except Exception as e:
        return "unavailable: %r" % (e,)


# ------------------------------------------------------------------------------------------------ main

def visible_devices():
    """HIP devices this process can see (hipGetDeviceCount through ctypes: no torch import in the launcher)."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return int(n.value) if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


def spawn_ranks(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment): become the launcher —
    re-exec under torch.distributed.run with one rank per GPU (rank i -> device i) and the same arguments, so that the command
    prints ONE line with n_gpus = N.  Fewer than N visible devices is an error, never a silent one-GPU line (unless
    --same-device, the one-GPU flow test)."""
    have = visible_devices()
    if not a.same_device and have < a.gpus:
        sys.stderr.write("bench.py: --gpus %d asked for, %d HIP device(s) visible: refusing to run (a line measured on fewer GPUs than "
                         "it names would be wrong; --same-device runs every rank on device 0 as a flow test)\n" % (a.gpus, have))
        raise SystemExit(2)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without a launcher: starting %d ranks myself: %s\n" % (a.gpus, a.gpus, " ".join(cmd)))
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)  # (does not return)
    # stdout carries exactly ONE line, the JSON: libraries that write to C stdout (RCCL prints a version banner from
    # every rank) are pointed at stderr for the whole run; fd 1 is restored only for rank 0's final print
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:  # (a launcher that started another number of ranks than the line would name)
        raise SystemExit("--gpus %d must equal WORLD_SIZE %d" % (a.gpus, world))
    import torch
    import rio_gp
    import synth
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    rio_gp.build()
    if a.same_device:
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks, %d visible device(s): one GPU per rank or --same-device" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    workload = a.workload or ("c3" if world == 1 else "c4")
    dist = None
    if world > 1 or a.force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
    if workload == "c5" and world > 1:
        raise SystemExit("--workload c5 is a single-GPU line (the row-sharded churn tick is covered by tools/soak_sharded.py)")

    if dist is not None:
        # ---- row-sharded runs: the primary measurement, then (N>1, c4) the weak-scaled config 3 in the same run
        prim = run_sharded(a, dist, torch, rio_gp, synth, workload, rank, world, local_rank)
        weak = None
        if workload == "c4" and world > 1 and not a.no_weak:
            weak = run_sharded(a, dist, torch, rio_gp, synth, "c3", rank, world, local_rank)
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return
        strong = workload == "c4"
        m = prim["nodes"]
        out = {
            "metric": "placement decisions/sec + achieved HBM GB/s, 10M objects x 1 024 nodes" if not strong else
                      "placement decisions/sec + achieved HBM GB/s, 100M objects x 4 096 nodes row-sharded across the GPUs",
            "value": prim["value"], "unit": "decisions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": prim["ms_per_step"], "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": ("config 4: %d objects x %d nodes, %d per GPU (ONE table, rows sharded over %d ranks), Zipf(1.1) "
                                    "load, cap 1.25x of the global load, cold start" % (prim["rows_total"], m, prim["rows_this_rank"], world))
                       if strong else ("%s: %d objects x %d nodes per GPU of one %d-row table, cap from the global load, cold start"
                                       % (workload, prim["rows_this_rank"], m, prim["rows_total"])),
                       "objects_per_gpu": prim["rows_this_rank"], "objects_total": prim["rows_total"], "nodes": m,
                       "parallelism": "rows sharded x%d" % world,
                       "fast_path_of_a_tick": {"p2p": "row-sharded solve: k_scan -> k_resolve_xchg (every workgroup stores its eight nodes' local sums "
                                       "straight into every peer's HBM window over xGMI as data-tagged 8-byte words, polls the same "
                                       "words of every rank and resolves its nodes); one stream, two launches, no collective call, "
                                       "no flag; verdicts read at the end",
                                "native": "row-sharded solve: k_scan + k_resolve + pack -> ncclAllGather of %d B/rank issued by the "
                                          "library on a second stream -> k_shard_import; verdicts read at the end" % (8 * (2 * m + 8)),
                                "torch": "row-sharded solve: k_scan + k_resolve + pack -> torch.distributed all_gather (RCCL) of %d "
                                         "B/rank -> k_shard_import; verdicts read at the end" % (8 * (2 * m + 8))}[prim["exchange"]],
                       "value_is": DEF_STRONG if strong else DEF_WEAK,
                       "tick": "rio_gp_shard_tick_async: scan, one-launch exchange over the peer-to-peer windows, the guarded fix-up "
                               "chain, commit; nothing waits on the host, at most 48 ticks in flight" if prim["asynchronous"] else
                               "ShardedSolver.tick: solve_async + solve_wait (verdict, exchanges, global counters) + commit; the host "
                               "sequences the exchanges of the '%s' rung" % prim["exchange"],
                       "ranks": prim["ranks"], "rccl": prim["rccl"],
                       "exchange": prim["exchange"], "exchange_ladder": prim["exchange_ladder"],
                       "peer_access": peer_matrix(torch, world, a.same_device), "slow_path_steps": prim["slow_path_steps"]},
            "gpu_ms_per_step_events": prim["gpu_ms_per_step_events"],
            "parity": prim["parity"],
            "roofline": {"bound": "hbm", "achieved": prim["whole_step_achieved_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": prim["whole_step_achieved_GBps"] / HBM_PEAK_GBPS, "traffic": None,
                         "traffic_source": "not measured in sharded runs (PMC passes are single-process)",
                         "kernel": "whole sharded step on rank 0 (k_scan + exchange/resolve), HIP events on the library's stream",
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_DECISION * prim["rows_this_rank"]},
            "stats_last_step": prim["stats_last_step"],
            "cold_resolve_uncommitted": prim["cold_resolve_uncommitted"],
            "committed_ticks": prim["committed_ticks"],
        }
        pt = lambda r_: {"value": r_["value"], "unit": "decisions/s", "ms_per_tick": r_["ms_per_step"], "n_gpus": world,
                         "rows_total": r_["rows_total"], "exchange": r_["exchange"]}
        out["scaling_points"] = {"strong_config4_committed_tick": dict(pt(prim), definition=DEF_STRONG) if strong else None,
                                 "weak_config3_committed_tick": dict(pt(weak), definition=DEF_WEAK) if weak is not None else
                                 (dict(pt(prim), definition=DEF_WEAK) if workload == "c3" else None)}
        if weak is not None:
            out["weak_config3"] = {"metric": "placement decisions/sec, 10M objects x 1 024 nodes per GPU (weak scaling)",
                                   "value": weak["value"], "unit": "decisions/s", "scaling": "weak", "ms_per_step": weak["ms_per_step"],
                                   "objects_per_gpu": weak["rows_this_rank"], "objects_total": weak["rows_total"], "nodes": weak["nodes"],
                                   "exchange": weak["exchange"], "slow_path_steps": weak["slow_path_steps"], "parity": weak["parity"],
                                   "value_is": DEF_WEAK, "cold_resolve_uncommitted": weak["cold_resolve_uncommitted"],
                                   "committed_ticks": weak["committed_ticks"]}
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        bad = [p for p in (prim["parity"], weak["parity"] if weak else None) if p is not None and not p["equal"]]
        for r_ in (prim, weak):
            cp = ((r_ or {}).get("committed_ticks") or {}).get("parity") if r_ else None
            if cp is not None and not cp["equal"]:
                bad.append(cp)
        if bad:
            sys.exit(3)
        return

    # ---- N = 1 -----------------------------------------------------------------------------------------------
    cfg = synth.config("c3" if workload == "c5" else workload, n_override=(a.objects or None))
    n, m = cfg["n"], cfg["m"]
    g = rio_gp.GpuPlacement(n, m, device=local_rank)
    g.set_nodes(cfg["cap"], cfg["alive"])
    g.set_objects(n, cfg["load"], cfg["aff"])
    if workload == "c3w":
        g.set_assign(cfg["cur"])
    if workload == "c5":
        return bench_churn(a, g, cfg, saved_stdout, rio_gp, local_rank)

    # ---- (1) the headline: a pipelined stream of COMMITTED ticks (rio_gp_tick_async: solve + fix-up if needed + commit, each
    #      tick consuming the previous one's table; counters read afterwards).  The stream starts from the cold table (tick 1:
    #      every object claims its requester), the K timed steps follow W warm-ups: every row gets its decision every tick.
    for _ in range(a.warmup):
        g.tick_async()
    if a.warmup:
        g.tick_wait()
    tick_arr = (rio_gp.Stats * max(a.steps, 1))()
    tick_got = []
    dt, gpu_ms, sts, _ = timed_steps(a, g, None, torch, g.tick_async, lambda: (g.stats_list(tick_arr, tick_got[0]), 0),
                                     wait_c=lambda: tick_got.append(g.tick_wait_into(tick_arr)))
    st = sts[-1]
    n_slow = sum(x["slow_path"] for x in sts)
    assert st["kept"] + st["claimed"] + st["spilled"] + st["unplaced"] == n
    total_decisions = n * a.steps
    # the synchronous form of the same tick (the host reads every tick's counters before the next one)
    dep = None
    try:
        g.tick()
        g.sync()
        st_c = rio_gp.Stats()
        t0 = time.perf_counter()
        for _ in range(100):
            g.tick_struct(st_c)   # (the C call alone: the host reads the tick's counters, no Python dictionary is built)
        dep = (time.perf_counter() - t0) / 100 * 1e3
    except Exception:
        dep = None

    # ---- (2) the cold table, re-solved WITHOUT committing (round 2's headline, kept under its own name: a server never does
    #      this, but it is the configuration the dominant kernel is profiled on — cold start, every row pending)
    g.set_assign(cfg["cur"])
    for _ in range(a.warmup):
        g.solve_async()
    if a.warmup:
        g.solve_wait()
    g.sync()
    t0 = time.perf_counter()
    g.timer_begin()
    for _ in range(a.steps):
        g.solve_async()
    cold_gpu_ms = g.timer_end()
    cst, cold_slow = g.solve_wait()
    g.sync()
    cold_dt = time.perf_counter() - t0

    # per-launch duration of the dominant kernel, HIP events on the library's own stream
    scan_ms, res_ms = [], []
    if cold_slow == 0:
        for _ in range(max(10, min(a.steps, 100))):
            s_ms, r_ms = g.solve_profiled()
            scan_ms.append(s_ms)
            res_ms.append(r_ms)
    probe = None
    if cold_slow == 0:
        try:  # what this chip's memory system gives a plain grid-stride kernel with the same 3-in/1-out mix (lab build)
            gl = rio_gp.LabPlacement(n, m, device=local_rank)
            gl.set_objects(n, cfg["load"], cfg["aff"])
            ms = gl.stream_probe(0, 20)
            gl.close()
            probe = {"pattern": "grid-stride 2048x256, read cur/load/aff + write one column, no other work",
                     "ms": ms, "GBps": ALGO_BYTES_PER_DECISION * n / ms / 1e6}
        except Exception as e:  # measurement aid only
            probe = {"error": str(e)}
    cold = None
    if cold_slow == 0 and not a.no_cold and workload == "c3":
        # The headline table (160 MB of columns) fits the 256 MiB Infinity Cache, so repeated solves are partly served
        # by it.  Same kernel, same per-row inputs tiled 4x (640 MB of columns, capacities scaled): every launch streams
        # from HBM.  Kernel-only AND whole-step (k_scan + k_resolve) AND committed ticks.
        try:
            k = 4
            loadk, affk = np.tile(cfg["load"], k), np.tile(cfg["aff"], k)
            gb = rio_gp.GpuPlacement(k * n, m, device=local_rank)
            gb.set_nodes(synth.uniform_cap(loadk, m), cfg["alive"])
            gb.set_objects(k * n, loadk, affk)
            for _ in range(5):
                gb.solve_profiled()
            cs = [gb.solve_profiled()[0] for _ in range(30)]
            gl = rio_gp.LabPlacement(k * n, m, device=local_rank)
            gl.set_objects(k * n, loadk, affk)
            pm = gl.stream_probe(0, 10)
            gl.close()
            for _ in range(3):
                gb.solve_async()
            gb.solve_wait()
            gb.sync()
            gb.timer_begin()
            for _ in range(30):
                gb.solve_async()
            wms = gb.timer_end() / 30
            gb.solve_wait()
            for _ in range(3):
                gb.tick_async()
            gb.tick_wait()
            gb.sync()
            gb.timer_begin()
            for _ in range(30):
                gb.tick_async()
            tms = gb.timer_end() / 30
            gb.tick_wait()
            gb.close()
            cms = float(np.mean(cs))
            fr = lambda ms_: ALGO_BYTES_PER_DECISION * k * n / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBPS
            cold = {"rows": k * n, "column_bytes": 16 * k * n, "kernel_ms": cms,
                    "achieved": ALGO_BYTES_PER_DECISION * k * n / (cms * 1e-3) / 1e9, "unit": "GB/s", "frac": fr(cms),
                    "whole_step_ms": wms, "whole_step_frac": fr(wms), "committed_tick_ms": tms, "committed_tick_frac": fr(tms),
                    "stream_probe_GBps": ALGO_BYTES_PER_DECISION * k * n / pm / 1e6,
                    "note": "the headline rows tiled 4x: beyond the 256 MiB Infinity Cache; kernel = k_scan alone (dispatch events), "
                            "whole step = k_scan + k_resolve back to back, committed tick = rio_gp_tick_async (30 pipelined steps "
                            "between two events each)"}
        except Exception as e:  # measurement aid only
            cold = {"error": str(e)}
    parity = None
    t_orc = None
    cold_column = None
    if not a.no_parity:
        g.set_assign(cfg["cur"])
        parity, t_orc = parity_single(g, cfg)   # leaves the table committed (warm)
        cold_column = parity.pop("_solved_column")
    c4one = None
    if not a.no_c4 and workload == "c3" and not a.objects:
        # BASELINE config 4 on ONE GPU (the N=1 point of the strong-scaling curve): 100 M x 4 096, parity at size
        try:
            c4 = synth.config("c4")
            g4 = rio_gp.GpuPlacement(c4["n"], c4["m"], device=local_rank)
            g4.set_nodes(c4["cap"], c4["alive"])
            g4.set_objects(c4["n"], c4["load"], c4["aff"])
            for _ in range(3):
                g4.solve_async()
            g4.solve_wait()
            g4.sync()
            t0 = time.perf_counter()
            for _ in range(20):
                g4.solve_async()
            st4, slow4 = g4.solve_wait()
            t4 = (time.perf_counter() - t0) / 20
            sc4 = [g4.solve_profiled()[0] for _ in range(10)] if slow4 == 0 else []
            par4 = None if a.no_parity else parity_single(g4, c4)[0]
            if par4 is not None:
                par4.pop("_solved_column", None)
            for _ in range(2):
                g4.tick_async()
            g4.tick_wait()
            g4.sync()
            t0 = time.perf_counter()
            for _ in range(10):
                g4.tick_async()
            g4.tick_wait()
            tt4 = (time.perf_counter() - t0) / 10
            fr4 = lambda sec: ALGO_BYTES_PER_DECISION * c4["n"] / sec / 1e9 / HBM_PEAK_GBPS
            # config 5's churn on this table: 10 % of the 4 096 nodes flip per tick, ~10 M rows evicted and re-placed
            churn4 = None
            try:
                masks4 = [synth.churn_mask(c4["m"], 2 + k) for k in range(13)]
                base4 = g4.get_assign() if not a.no_parity else None
                for k in range(3):
                    g4.set_alive_all(masks4[k])
                    g4.tick_async()
                g4.tick_wait()
                g4.sync()
                t0 = time.perf_counter()
                for k in range(3, 13):
                    g4.set_alive_all(masks4[k])
                    g4.tick_async()
                sts4 = g4.tick_wait()
                tc4 = (time.perf_counter() - t0) / 10
                churn4 = {"ms_per_tick": tc4 * 1e3, "value": c4["n"] / tc4, "unit": "decisions/s", "frac_of_roofline": fr4(tc4),
                          "ticks": 10, "stats_last_tick": sts4[-1],
                          "step": "rio_gp_set_alive_all + rio_gp_tick_async (k_inc_scan, k_rebal, k_resolve, k_cut_find, k_fill x 2)"}
                if base4 is not None:
                    import pyoracle
                    t1 = time.perf_counter()
                    ref4 = base4
                    for k in range(13):
                        ref4, used4, ost4 = pyoracle.tick(ref4, c4["load"], c4["aff"], c4["cap"], masks4[k], 2)
                    churn4["parity"] = {"checked_rows": int(c4["n"]), "ticks_replayed": 13,
                                        "equal": bool(np.array_equal(g4.get_assign(), ref4)) and bool(np.array_equal(g4.get_nodes()[2], used4))
                                        and sts4[-1] == ost4,
                                        "against": "oracle/placement_oracle.c orc_tick chained over the same 13 liveness masks",
                                        "oracle_seconds": time.perf_counter() - t1}
                    del ref4, base4
            except Exception as e:  # measurement aid only
                churn4 = {"error": repr(e)}
            g4.close()
            c4one = {"workload": "config 4 on one GPU: %d objects x %d nodes, Zipf(1.1), cap 1.25x" % (c4["n"], c4["m"]),
                     "committed_tick": {"value": c4["n"] / tt4, "unit": "decisions/s", "ms_per_tick": tt4 * 1e3, "frac": fr4(tt4)},
                     "cold_resolve_uncommitted": {"value": c4["n"] / t4, "unit": "decisions/s", "ms_per_step": t4 * 1e3,
                                                  "frac": fr4(t4), "slow_path_steps": slow4},
                     "k_scan_ms": float(np.mean(sc4)) if sc4 else None,
                     "k_scan_frac": fr4(float(np.mean(sc4)) * 1e-3) if sc4 else None, "parity": par4,
                     "churn_tick_pipelined": churn4}
            del c4
        except Exception as e:  # measurement aid only
            c4one = {"error": repr(e)}
    c2rec = None
    if not a.no_c2 and workload == "c3" and not a.objects:
        # BASELINE config 2: 1 M x 256, load 1, cap 4 883, cold start — parity at size + the same three rates
        try:
            c2 = synth.config("c2")
            g2 = rio_gp.GpuPlacement(c2["n"], c2["m"], device=local_rank)
            g2.set_nodes(c2["cap"], c2["alive"])
            g2.set_objects(c2["n"], c2["load"], c2["aff"])
            par2 = None if a.no_parity else parity_single(g2, c2)[0]
            if par2 is not None:
                par2.pop("_solved_column", None)
            g2.set_assign(c2["cur"])
            for _ in range(10):
                g2.solve_async()
            g2.solve_wait()
            g2.sync()
            t0 = time.perf_counter()
            for _ in range(200):
                g2.solve_async()
            st2, slow2 = g2.solve_wait()
            t2 = (time.perf_counter() - t0) / 200
            sc2 = [g2.solve_profiled()[0] for _ in range(50)] if slow2 == 0 else []
            for _ in range(10):
                g2.tick_async()
            g2.tick_wait()
            g2.sync()
            t0 = time.perf_counter()
            for _ in range(200):
                g2.tick_async()
            g2.tick_wait()
            tt2 = (time.perf_counter() - t0) / 200
            g2.close()
            fr2 = lambda sec: ALGO_BYTES_PER_DECISION * c2["n"] / sec / 1e9 / HBM_PEAK_GBPS
            c2rec = {"workload": "config 2: %d objects x %d nodes, load 1, cap %d (uniform), cold start" % (c2["n"], c2["m"], int(c2["cap"][0])),
                     "committed_tick": {"value": c2["n"] / tt2, "unit": "decisions/s", "ms_per_tick": tt2 * 1e3, "frac": fr2(tt2)},
                     "cold_resolve_uncommitted": {"value": c2["n"] / t2, "unit": "decisions/s", "ms_per_step": t2 * 1e3,
                                                  "frac": fr2(t2), "slow_path_steps": slow2},
                     "k_scan_ms": float(np.mean(sc2)) if sc2 else None,
                     "k_scan_frac": fr2(float(np.mean(sc2)) * 1e-3) if sc2 else None,
                     "note": "16 MB of columns: every launch is a handful of microseconds — latency, not bandwidth", "parity": par2}
        except Exception as e:  # measurement aid only
            c2rec = {"error": repr(e)}
    bind = {}
    if not a.no_binding and workload == "c3" and not a.objects:
        for which in ("contended", "skew"):
            try:
                bind[which] = binding_record(a, cfg, rio_gp, local_rank, which)
            except Exception as e:  # a second measurement: it must not take the line down with it
                bind[which] = {"error": repr(e)}
    c5rec = None
    if not a.no_c5 and workload == "c3" and not a.objects:
        try:
            c5rec = churn_record(a, g, cfg, rio_gp, local_rank, steps=100, warmup=10)
        except Exception as e:
            c5rec = {"error": repr(e)}

    scan_avg = float(np.mean(scan_ms)) if scan_ms else None
    achieved = (ALGO_BYTES_PER_DECISION * n / (scan_avg * 1e-3) / 1e9) if scan_avg else None
    traffic, traffic_source, traffic_detail = None, "not measured", None
    if not a.no_pmc and cold_slow == 0 and workload == "c3":
        doc, src = pmc_traffic_in_run(n)
        if doc is not None:
            traffic, traffic_source = doc["hbm_bytes_per_launch"], src
            traffic_detail = {k: doc.get(k) for k in ("k_scan", "k_resolve", "calibration")}
        else:
            traffic_source = "in-run PMC passes unavailable (%s)" % src
    if traffic is None and a.traffic_json and os.path.exists(a.traffic_json):
        tj = json.load(open(a.traffic_json))
        if tj.get("n_rows") == n:  # measured for this row count only
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_source += "; value replayed from %s (rocprofv3 PMC passes of an earlier run of this build)" % \
                os.path.relpath(a.traffic_json, ROOT)
    fr = lambda sec: ALGO_BYTES_PER_DECISION * n / sec / 1e9 / HBM_PEAK_GBPS
    tick_s = dt / a.steps
    out = {
        "metric": "placement decisions/sec + achieved HBM GB/s, 10M objects x 1 024 nodes",
        "value": total_decisions / dt, "unit": "decisions/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": tick_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": "config 3: %d objects x %d nodes, Zipf(1.1) load, cap 1.25x; a stream of committed ticks that "
                               "starts from the cold table (all pending)" % (n, m) if workload == "c3" else workload,
                   "objects_per_gpu": n, "nodes": m, "parallelism": "rows sharded x1",
                   "step": "rio_gp_tick_async = k_scan + k_resolve (+ the fix-up, guarded on the device) + commit; nothing waits on the "
                           "host between ticks, every tick's counters are read at the end.  Once a tick's verdict says nothing is "
                           "left to fix and nothing has changed since, the following ticks are k_scan + k_resolve only: the scans "
                           "alternate between two streams and hand their rows over wave range by wave range (wave w of tick "
                           "k + 1 waits for wave w of tick k, not for its launch), k_resolve runs on a third stream behind "
                           "its scan",
                   "exchange": None, "slow_path_steps": n_slow},
        "gpu_ms_per_step_events": gpu_ms / a.steps,
        "committed_tick_frac": fr(tick_s),
        "dependent_tick_ms": dep,
        "dependent_tick_frac": fr(dep * 1e-3) if dep else None,
        "cold_resolve_uncommitted": {"value": n * a.steps / cold_dt, "unit": "decisions/s", "ms_per_step": cold_dt / a.steps * 1e3,
                                     "gpu_ms_per_step_events": cold_gpu_ms / a.steps, "frac": fr(cold_dt / a.steps),
                                     "slow_path_steps": cold_slow,
                                     "note": "round 2's headline: rio_gp_solve_async of the SAME cold table back to back, never committed "
                                             "(k_scan + k_resolve per step) — what the kernel figures below are profiled on, not what a server does"},
        "parity": parity,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
                     "traffic_source": traffic_source, "traffic_detail": traffic_detail,
                     "kernel": "k_scan", "kernel_ms": scan_avg,
                     "gpu_ms_per_step_events": gpu_ms / a.steps,   # the committed-tick stream between two HIP events on the library's stream
                     "kernel_ms_p10_p90": [float(np.percentile(scan_ms, 10)), float(np.percentile(scan_ms, 90))] if scan_ms else None,
                     "algorithmic_bytes_per_launch": ALGO_BYTES_PER_DECISION * n,
                     "resolve_kernel_ms": float(np.mean(res_ms)) if res_ms else None,
                     "frac_note": "frac = k_scan on the 160 MB table, which fits the 256 MiB Infinity Cache (MALL-assisted); the DRAM-bound "
                                  "and whole-tick fractions follow",
                     "frac_dram_bound": cold.get("whole_step_frac") if isinstance(cold, dict) else None,
                     "frac_dram_bound_kernel": cold.get("frac") if isinstance(cold, dict) else None,
                     "frac_committed_tick": fr(tick_s),
                     "frac_committed_tick_dram_bound": cold.get("committed_tick_frac") if isinstance(cold, dict) else None,
                     "frac_dependent_tick": fr(dep * 1e-3) if dep else None,
                     "frac_of_measured_copy_peak_6290": (achieved / 6290.0) if achieved else None,
                     "stream_probe": probe, "beyond_infinity_cache": cold},
        "scaling_points": {
            "strong_config4_committed_tick": (dict(c4one["committed_tick"], n_gpus=1, rows_total=C4_ROWS, exchange=None,
                                                   definition=DEF_STRONG) if isinstance(c4one, dict) and "committed_tick" in c4one else None),
            "weak_config3_committed_tick": ({"value": total_decisions / dt, "unit": "decisions/s", "ms_per_tick": tick_s * 1e3, "n_gpus": 1,
                                             "rows_total": n, "exchange": None, "definition": DEF_WEAK} if workload == "c3" and not a.objects else None)},
        "config2": c2rec,
        "config3_contended": bind.get("contended"),
        "config3_skew": bind.get("skew"),
        "config4_single_gpu": c4one,
        "config5_churn": c5rec,
        "stats_last_step": st,          # last committed tick of the stream: every row kept where the first tick put it
        "stats_cold_step": cst,         # the cold table re-solved: every row pending
    }
    port_parity = None
    if not a.no_cpu_baseline:
        port_parity, out["cpu_baseline"] = cpu_baseline(cfg, a.cpu_sample, t_orc, cold_column if workload in ("c3", "c2") else None)
        if parity is not None and port_parity is not None:
            parity["against_reference_port"] = port_parity
            parity["equal"] = parity["equal"] and port_parity["equal"]
    g.close()
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    os.dup2(saved_stdout, 1)
    print(json.dumps(out), flush=True)
    bad = [p for p in (parity, (c4one or {}).get("parity"), ((c4one or {}).get("churn_tick_pipelined") or {}).get("parity"),
                       (c2rec or {}).get("parity"), (c5rec or {}).get("parity"),
                       (bind.get("contended") or {}).get("parity"), (bind.get("skew") or {}).get("parity"))
           if p is not None and not p["equal"]]
    if bad:
        sys.exit(3)


if __name__ == "__main__":
    main()
