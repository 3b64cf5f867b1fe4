"""Row-sharded solve on the GPU: the HIP shard phases (rio_gp_shard_*), sequenced by the same
ShardedSolver the multi-GPU bench uses, against the whole-table CPU oracle — bit-exact.  One MI355X is
enough: G shards = G handles on the device (LocalExchange), a world_size-1 RCCL group (DistExchange,
"nccl"), and two PROCESSES sharing the GPU over gloo."""
import json
import os
import socket
import sys

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NONE = 0xFFFFFFFF
INF = 0xFFFFFFFFFFFFFFFF


@pytest.fixture(scope="module")
def gp():
    import rio_gp
    rio_gp.build()
    return rio_gp


def make_engines(gp, case, bounds, rounds=2, stream=None):
    import torch
    import sharded
    cur, load, aff, cap, alive = case
    stream = stream or torch.cuda.Stream(torch.device("cuda", 0))
    engines = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        n = hi - lo
        g = gp.GpuPlacement(max(n, 1), max(len(cap), 1), spill_rounds=rounds)
        g.set_nodes(cap, alive, m=len(cap))
        g.set_objects(n, load[lo:hi], aff[lo:hi])
        if n:
            g.set_assign(cur[lo:hi])
        engines.append(sharded.HipShardEngine(g, 0, stream))
    return engines


def check(gp, oracle, case, bounds, rounds=2):
    import sharded
    engines = make_engines(gp, case, bounds, rounds)
    sol = sharded.ShardedSolver(engines, sharded.LocalExchange(len(engines)), spill_rounds=rounds)
    st = sol.solve()
    got = np.concatenate([e.g.get_solved() if e.g.num_objects else np.zeros(0, np.uint32) for e in engines])
    want, used, ost = oracle.tick(*case, rounds)
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
    assert st == ost
    sol.commit()
    for e in engines:
        assert np.array_equal(e.g.get_nodes()[2], used)  # every rank holds the GLOBAL used vector
        e.g.close()
    return ost


def cases():
    from test_sharded_protocol import CASES, random_case
    return [random_case(100 + i, **kw) for i, kw in enumerate(CASES)]


@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_hip_shards_equal_whole_table_oracle(gp, oracle, G):
    import sharded
    for case in cases():
        check(gp, oracle, case, sharded.shard_bounds(len(case[0]), G))


def test_hip_ragged_and_empty_shards(gp, oracle):
    from test_sharded_protocol import random_case
    case = random_case(7, n=3000, m=12, cap_scale=0.85, dead_frac=0.1, zero_load=0.1)
    for bounds in ([0, 0, 1, 1, 2999, 3000, 3000], [0, 1500, 1500, 3000], [0, 7, 2000, 2001, 3000]):
        check(gp, oracle, case, bounds)


@pytest.mark.parametrize("rounds", [1, 4])
def test_hip_shard_spill_rounds(gp, oracle, rounds):
    from test_sharded_protocol import random_case
    import sharded
    case = random_case(11, n=50_000, m=20, cap_scale=0.95, dead_frac=0.15)
    check(gp, oracle, case, sharded.shard_bounds(50_000, 4), rounds)


def test_hip_shards_config3_shape(gp, oracle):
    import sharded
    cfg = synth.config("c3", n_override=2_000_000)
    case = (cfg["cur"], cfg["load"], cfg["aff"], cfg["cap"], cfg["alive"])
    st = check(gp, oracle, case, sharded.shard_bounds(cfg["n"], 8))
    assert st["slow_path"] == 0
    # the same table squeezed: hundreds of cut nodes, forced nodes on the upper ranks, spill, unplaced rows
    cap = (cfg["cap"] * np.uint64(9)) // np.uint64(10)
    alive = cfg["alive"].copy()
    alive[3::17] = 0
    st = check(gp, oracle, (synth.warm_assign(cfg["n"], cfg["m"]), cfg["load"], cfg["aff"], cap, alive),
               sharded.shard_bounds(cfg["n"], 8))
    assert st["cut_nodes"] > 0 and st["evicted"] > 0


def test_hip_shard_rccl_world1_async(gp, oracle):
    """The bench's path: DistExchange over "nccl" (= RCCL), kernels and collective ordered on one torch stream,
    several solves in flight, one verdict read at the end."""
    import torch
    import torch.distributed as dist
    import sharded
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg = synth.config("c3", n_override=1_000_000)
        case = (cfg["cur"], cfg["load"], cfg["aff"], cfg["cap"] * np.uint64(2), cfg["alive"])
        engines = make_engines(gp, case, [0, cfg["n"]])
        want, used, ost = oracle.tick(*case, 2)
        for pipeline in (False, True):  # True: exchange + global resolve on a second stream (the bench's default)
            sol = sharded.ShardedSolver(engines, sharded.DistExchange(), pipeline=pipeline)
            for _ in range(9):
                sol.solve_async()
            st, n_slow = sol.solve_wait()
            assert n_slow == 0 and st == ost
            assert np.array_equal(engines[0].g.get_solved(), want)
        # a pipelined solve that needs the fix-up: same answer as the whole-table oracle
        tight = (case[0], case[1], case[2], case[3] // np.uint64(3), case[4])
        engines[0].g.set_nodes(tight[3], tight[4])
        sol.solve_async()
        sol.solve_async()
        st, n_slow = sol.solve_wait()
        w2, u2, o2 = oracle.tick(*tight, 2)
        assert n_slow == 2 and st == o2 and np.array_equal(engines[0].g.get_solved(), w2)
        engines[0].g.set_nodes(case[3], case[4])
        sol.solve_async()
        st, n_slow = sol.solve_wait()
        assert n_slow == 0 and st == ost
        sol.commit()
        assert np.array_equal(engines[0].g.get_nodes()[2], used)
        engines[0].g.close()
    finally:
        dist.destroy_process_group()


def test_hip_shard_native_rccl_world1(gp, oracle):
    """rio_gp_shard_comm_init + rio_gp_shard_solve_async: the library issues ncclAllGather itself."""
    import torch
    import torch.distributed as dist
    import sharded
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=0, world_size=1)  # control channel only (moves the ncclUniqueId)
    try:
        cfg = synth.config("c3", n_override=1_000_000)
        case = (cfg["cur"], cfg["load"], cfg["aff"], cfg["cap"] * np.uint64(2), cfg["alive"])
        eng = make_engines(gp, case, [0, cfg["n"]])[0]
        sol = sharded.ShardedSolver([eng], sharded.NativeRcclExchange(eng))
        want, used, ost = oracle.tick(*case, 2)
        for _ in range(9):
            sol.solve_async()
        st, n_slow = sol.solve_wait()
        assert n_slow == 0 and st == ost and np.array_equal(eng.g.get_solved(), want)
        tight = (case[0], case[1], case[2], case[3] // np.uint64(3), case[4])
        eng.g.set_nodes(tight[3], tight[4])
        sol.solve_async()
        st, n_slow = sol.solve_wait()
        w2, u2, o2 = oracle.tick(*tight, 2)
        assert n_slow == 1 and st == o2 and np.array_equal(eng.g.get_solved(), w2)
        sol.commit()
        assert np.array_equal(eng.g.get_nodes()[2], u2)
        eng.g.close()
    finally:
        dist.destroy_process_group()


def test_hip_shard_p2p_world1(gp, oracle):
    """Peer-to-peer window path with a single rank: export/connect/handshake, the one-launch tagged exchange
    (k_resolve_xchg), Y exchanges."""
    import torch
    import torch.distributed as dist
    import sharded
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        cfg = synth.config("c3", n_override=1_000_000)
        case = (cfg["cur"], cfg["load"], cfg["aff"], cfg["cap"] * np.uint64(2), cfg["alive"])
        eng = make_engines(gp, case, [0, cfg["n"]])[0]
        sol = sharded.ShardedSolver([eng], sharded.P2PExchange(eng))
        want, used, ost = oracle.tick(*case, 2)
        for _ in range(11):
            sol.solve_async()
        st, n_slow = sol.solve_wait()
        assert n_slow == 0 and st == ost and np.array_equal(eng.g.get_solved(), want)
        tight = (case[0], case[1], case[2], case[3] // np.uint64(3), case[4])
        eng.g.set_nodes(tight[3], tight[4])
        sol.solve_async()
        st, n_slow = sol.solve_wait()
        w2, u2, o2 = oracle.tick(*tight, 2)
        assert n_slow == 1 and st == o2 and np.array_equal(eng.g.get_solved(), w2)
        eng.g.close()
    finally:
        dist.destroy_process_group()


ASYNC_TICKS = 9


def _async_mask(m, k):
    """liveness of tick k of the asynchronous stream: every third tick changes nothing, the others flip a few nodes"""
    alive = np.ones(m, np.uint8)
    if k % 3 != 2:
        alive[np.random.default_rng(500 + k - (k % 3 == 1)).choice(m, 5, replace=False)] = 0    # (k, k+1 share a mask twice)
    return alive


def _proc(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import rio_gp
    import sharded
    from test_sharded_protocol import random_case
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        case = random_case(31, n=200_000, m=50, cap_scale=0.92, dead_frac=0.1, zero_load=0.05)
        b = sharded.shard_bounds(len(case[0]), world)
        eng = make_engines(rio_gp, case, [b[rank], b[rank + 1]])[0]
        if os.environ.get("RIO_TEST_EXCHANGE") == "p2p":   # IPC-mapped windows between the two processes
            sol = sharded.ShardedSolver([eng], sharded.P2PExchange(eng))
            for _ in range(5):
                sol.solve_async()   # back-to-back steps: slot reuse and sequence flags across processes
            sol.solve_wait()
        else:
            sol = sharded.ShardedSolver([eng], sharded.DistExchange(stage_through_host=True))
        st = sol.tick()
        a1, u1 = eng.g.get_assign(), eng.g.get_nodes()[2]
        extra = {}
        if os.environ.get("RIO_TEST_EXCHANGE") == "p2p":
            # asynchronous committed ticks (rio_gp_shard_tick_async): a churn stream with nothing waiting on the host —
            # liveness flips, ticks that need the whole fix-up chain and ticks that need none of it, back to back
            m = len(case[3])
            sts = []
            for k in range(ASYNC_TICKS):
                eng.g.set_alive_all(_async_mask(m, k))
                sol.tick_async()
                if k == 3:
                    sts += sol.tick_wait()      # a wait in the middle of the stream
            sts += sol.tick_wait()
            keys = sorted(sts[0])
            extra = dict(a2=eng.g.get_assign(), used2=eng.g.get_nodes()[2],
                         st2=np.array([[s_[k] for k in keys] for s_ in sts], np.uint64))
        np.savez(os.path.join(out_dir, "g%d.npz" % rank), a=eng.g.get_assign() if not extra else a1, used=eng.g.get_nodes()[2] if not extra else u1,
                 st=np.array([st[k] for k in sorted(st)], np.uint64), **extra)
        eng.g.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange,world", [("gloo", 2), ("p2p", 2), ("p2p", 3), ("p2p", 5), ("p2p", 8)])
def test_hip_shards_several_processes_one_gpu(gp, oracle, tmp_path, exchange, world, monkeypatch):
    """One process per rank, all on the one GPU of the box: real cross-process windows, every rank's claim prefix over
    the lower ranks, forced nodes, back-to-back steps (slot reuse), then a committed tick with the fix-up exchanges."""
    import torch.multiprocessing as mp
    from test_sharded_protocol import random_case
    monkeypatch.setenv("RIO_TEST_EXCHANGE", exchange)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_proc, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    case = random_case(31, n=200_000, m=50, cap_scale=0.92, dead_frac=0.1, zero_load=0.05)
    want, used, ost = oracle.tick(*case, 2)
    parts = [np.load(os.path.join(str(tmp_path), "g%d.npz" % r)) for r in range(world)]
    assert np.array_equal(np.concatenate([z["a"] for z in parts]), want)
    for z in parts:
        assert np.array_equal(z["used"], used)
        assert [int(v) for v in z["st"]] == [ost[k] for k in sorted(ost)]
    if exchange == "p2p":     # the asynchronous stream that followed, against the oracle chained over the same masks
        cur, load, aff, cap, alive = case
        ref, osts = want, []
        for k in range(ASYNC_TICKS):
            ref, used, ost = oracle.tick(ref, load, aff, cap, _async_mask(len(cap), k), 2)
            osts.append([ost[key] for key in sorted(ost)])
        assert np.array_equal(np.concatenate([z["a2"] for z in parts]), ref)
        for z in parts:
            assert np.array_equal(z["used2"], used)
            assert z["st2"].astype(np.int64).tolist() == osts
        assert any(o[sorted(ost).index("slow_path")] for o in osts)


def _proc_lost(rank, world, port, out_dir):
    """Three ranks share the peer-to-peer windows; the last one disappears between two rio_gp_shard_tick_async calls."""
    for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import time
    import torch
    import torch.distributed as dist
    import rio_gp
    import sharded
    from test_sharded_protocol import random_case
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    case = random_case(33, n=120_000, m=40, cap_scale=0.92, dead_frac=0.1, zero_load=0.05)
    cur, load, aff, cap, alive = case
    b = sharded.shard_bounds(len(cur), world)
    eng = make_engines(rio_gp, case, [b[rank], b[rank + 1]])[0]
    survivors = dist.new_group(list(range(world - 1)))   # (made while everybody is still there)
    ex = sharded.P2PExchange(eng)
    sol = sharded.ShardedSolver([eng], ex)
    for k in range(2):                                   # a healthy stream first
        eng.g.set_alive_all(_async_mask(len(cap), k))
        sol.tick_async()
    healthy = sol.tick_wait()
    dist.barrier()
    if rank == world - 1:
        os._exit(0)                                      # gone: no close, no goodbye — its windows' words never change again
    rec = {"healthy_ticks": len(healthy)}
    t0 = time.time()
    eng.tick_async()                                     # two ticks enqueued against a peer that is not there
    eng.tick_async()
    try:
        eng.tick_wait()
        rec["wait_rc"] = 0
    except rio_gp.ObjectPlacementError as e:
        rec["wait_rc"], rec["wait_kind"], rec["wait_text"] = e.rc, e.kind, str(e)
    rec["wait_seconds"] = time.time() - t0
    try:                                                 # the session is marked: the next call fails at once, nothing is enqueued
        t1 = time.time()
        eng.tick_async()
        rec["next_rc"] = 0
    except rio_gp.ObjectPlacementError as e:
        rec["next_rc"], rec["next_text"], rec["next_seconds"] = e.rc, str(e), time.time() - t1
    ex.close()                                           # rio_gp_shard_p2p_close: unmaps the dead rank's window as well
    rec["ready_after_close"] = int(sharded._lib().rio_gp_shard_p2p_ready(eng.g.handle))
    # ... and the handle is as good as new: the survivors' rows as a table of their own, solved over the exchange rung that
    # works between processes on one device (torch.distributed; with a GPU per rank that is where the RCCL rungs come in)
    eng.g.set_alive_all(alive)
    eng.g.set_assign(cur[b[rank]:b[rank + 1]])
    sol2 = sharded.ShardedSolver([eng], sharded.DistExchange(group=survivors, stage_through_host=True))
    st = sol2.tick()
    np.savez(os.path.join(out_dir, "lost%d.npz" % rank), a=eng.g.get_assign(), used=eng.g.get_nodes()[2],
             st=np.array([st[k] for k in sorted(st)], np.uint64), rec=np.array([json.dumps(rec)]))
    eng.g.close()
    os._exit(0)                                          # (no collective teardown with a rank that is gone)


def test_peer_lost_mid_stream(gp, oracle, tmp_path):
    """A rank that disappears MID-stream on the peer-to-peer rung (round-5 verdict: only set-up failures had been injected).
    The survivors' kernels poll IPC-mapped windows for words that will never come: the first wait runs into the 3 s time-out
    and raises the error word, every wait behind it gives up at once, rio_gp_shard_tick_wait returns RIO_GP_EUPSTREAM
    (ObjectPlacementError::Upstream) and marks the session out of step, the next call fails immediately with the text that
    names the way out, rio_gp_shard_p2p_close works — no hang, no device fault — and the same handles then solve the
    survivors' rows over another exchange rung, equal to the oracle."""
    import torch.multiprocessing as mp
    from test_sharded_protocol import random_case
    world = 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_proc_lost, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    cur, load, aff, cap, alive = random_case(33, n=120_000, m=40, cap_scale=0.92, dead_frac=0.1, zero_load=0.05)
    import sharded
    b = sharded.shard_bounds(len(cur), world)
    nsurv = b[world - 1]
    want, used, ost = oracle.tick(cur[:nsurv], load[:nsurv], aff[:nsurv], cap, alive, 2)
    parts = [np.load(os.path.join(str(tmp_path), "lost%d.npz" % r)) for r in range(world - 1)]
    for z in parts:
        rec = json.loads(str(z["rec"][0]))
        assert rec["healthy_ticks"] == 2
        assert rec["wait_rc"] == gp.EUPSTREAM and rec["wait_kind"] == "Upstream" and "timed out" in rec["wait_text"], rec
        assert 2.5 < rec["wait_seconds"] < 12, rec          # ONE time-out for the two ticks' fourteen launches each, not one per wait
        assert rec["next_rc"] == gp.EUPSTREAM and "lost step" in rec["next_text"] and rec["next_seconds"] < 0.5, rec
        assert rec["ready_after_close"] == 0
        assert np.array_equal(z["used"], used)
        assert [int(v) for v in z["st"]] == [ost[k] for k in sorted(ost)]
    assert np.array_equal(np.concatenate([z["a"] for z in parts]), want)


def test_soak_three_processes_peer_to_peer(gp):
    """40 committed ticks with membership churn and bursts of back-to-back asynchronous solves between three processes on
    the one GPU; every rank checks its rows and the global `used` vector against the whole-table oracle after every tick
    (tools/soak_sharded.py; profiles/archive/r03_soak_sharded.json holds longer runs)."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_sharded.py"), "3", "40", "90000", "96", "11"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["all_ranks_equal_oracle_every_tick"] is True

