"""Row-sharded whole-table solve across the GPUs of one node.

One process per GPU; rank r owns the contiguous rows [off_r, off_r + n_r) of the object table
(shard order = index order), the node table is replicated.  Rows couple only through per-node
load vectors, so the data path has exactly one kind of exchange: an all-gather of a small u64
record per rank (X = 2m+8 words on every solve; Y = m+2 words per fix-up step) — RCCL over xGMI
through torch.distributed (backend "nccl" IS RCCL on ROCm), <= 64 KiB per rank, latency-bound.
Every cross-rank reduction is an integer sum taken in rank order inside the shard kernels, so
the composed result equals the unsharded solve (and the CPU oracle) bit for bit.

This module only sequences the C-ABI phases (rio_gp_shard_*, include/rio_gpu_placement.h) and
the collectives; it computes nothing itself and has no CPU fallback.  The engine interface is
small on purpose: tests drive the same `ShardedSolver` with a numpy engine over gloo
(tests/shard_engine_cpu.py) to check the protocol itself on machines without a GPU.

Reference anchor: the coupling being sharded is the per-server "where do new objects go"
decision of Service::get_or_create_placement (rio-rs/src/service.rs:193-254); the reference
has no multi-device form of it.
"""
import ctypes as C

import numpy as np
import torch

import rio_gp

STAT_KEYS = ("n_objects", "kept", "evicted", "claimed", "spilled", "unplaced",
             "load_kept", "load_claimed", "load_spilled", "load_unplaced")


class ShardInfo(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("cut_nodes", "spill_rows", "local_fixup", "kept", "evicted",
                                          "claimants", "load_kept", "load_claim")]


class ShardTickInfo(C.Structure):
    _fields_ = [("local", rio_gp.Stats), ("cut_nodes", C.c_uint64), ("spill_rows", C.c_uint64),
                ("slow_path", C.c_uint32), ("rounds_run", C.c_uint32)]


_ready = False


def _lib():
    global _ready
    L = rio_gp.lib()
    if not _ready:
        vp = C.c_void_p
        L.rio_gp_set_stream.argtypes = [vp, vp]
        L.rio_gp_shard_words1.argtypes = [vp]
        L.rio_gp_shard_words1.restype = C.c_uint32
        L.rio_gp_shard_words2.argtypes = [vp]
        L.rio_gp_shard_words2.restype = C.c_uint32
        L.rio_gp_shard_scan.argtypes = [vp, vp]
        L.rio_gp_shard_resolve.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp]
        L.rio_gp_shard_verdict.argtypes = [vp, C.POINTER(ShardInfo), C.POINTER(C.c_uint32)]
        L.rio_gp_shard_cut.argtypes = [vp, C.c_int, vp]
        L.rio_gp_shard_merge.argtypes = [vp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.rio_gp_shard_spill.argtypes = [vp, C.c_uint32, C.c_int, vp]
        L.rio_gp_shard_finish.argtypes = [vp, C.POINTER(rio_gp.Stats)]
        L.rio_gp_shard_comm_unique_id.argtypes = [vp, C.c_char_p]
        L.rio_gp_shard_comm_init.argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.c_char_p]
        L.rio_gp_shard_solve_async.argtypes = [vp]
        L.rio_gp_shard_exchange.argtypes = [vp, vp, vp, C.c_uint64]
        L.rio_gp_shard_p2p_export.argtypes = [vp, C.c_uint32, vp]
        L.rio_gp_shard_p2p_connect.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
        L.rio_gp_shard_p2p_ready.argtypes = [vp]
        L.rio_gp_shard_p2p_close.argtypes = [vp]
        L.rio_gp_shard_tick_async.argtypes = [vp]
        L.rio_gp_shard_tick_wait.argtypes = [vp, C.POINTER(ShardTickInfo), C.c_uint32, C.POINTER(C.c_uint32)]
        _ready = True
    return L


class HipShardEngine:
    """One rank's shard: a rio_gp handle whose calls are enqueued on a torch stream, so that the
    RCCL all-gather torch.distributed issues is ordered against the kernels on the device (no host
    synchronisation on the fast path)."""

    def __init__(self, placement, device=0, stream=None):
        self.g = placement
        self.device = torch.device("cuda", device)
        self.stream = stream if stream is not None else torch.cuda.Stream(self.device)
        self.g._chk(_lib().rio_gp_set_stream(self.g.handle, C.c_void_p(self.stream.cuda_stream)))
        self.m = self.g.num_nodes
        self.words1 = int(_lib().rio_gp_shard_words1(self.g.handle))
        self.words2 = int(_lib().rio_gp_shard_words2(self.g.handle))

    def ctx(self):
        return torch.cuda.stream(self.stream)

    def new_buffer(self, words):
        with self.ctx():
            return torch.zeros(int(words), dtype=torch.int64, device=self.device)

    def scan(self, x):
        self.g._chk(_lib().rio_gp_shard_scan(self.g.handle, C.c_void_p(x.data_ptr())))

    def resolve(self, rank, n_ranks, xg, on_stream=None):
        st = C.c_void_p(on_stream.cuda_stream) if on_stream is not None else None
        self.g._chk(_lib().rio_gp_shard_resolve(self.g.handle, rank, n_ranks, C.c_void_p(xg.data_ptr()), st))

    def verdict(self):
        info, ns = ShardInfo(), C.c_uint32(0)
        self.g._chk(_lib().rio_gp_shard_verdict(self.g.handle, C.byref(info), C.byref(ns)))
        d = {k: int(getattr(info, k)) for k, _ in ShardInfo._fields_}
        d["n_slow"] = int(ns.value)
        return d

    def cut(self, run_local_fixup, y):
        self.g._chk(_lib().rio_gp_shard_cut(self.g.handle, int(bool(run_local_fixup)), C.c_void_p(y.data_ptr())))

    def merge(self, yg):
        rows, load = C.c_uint64(0), C.c_uint64(0)
        self.g._chk(_lib().rio_gp_shard_merge(self.g.handle, C.c_void_p(yg.data_ptr()), C.byref(rows), C.byref(load)))
        return int(rows.value), int(load.value)

    def spill(self, rnd, last, y):
        self.g._chk(_lib().rio_gp_shard_spill(self.g.handle, rnd, int(bool(last)), C.c_void_p(y.data_ptr())))

    def finish(self):
        st = rio_gp.Stats()
        self.g._chk(_lib().rio_gp_shard_finish(self.g.handle, C.byref(st)))
        return st.as_dict()

    def commit(self):
        self.g.commit()

    def tick_async(self):
        """One committed tick, nothing waits on the host (peer-to-peer windows only): rio_gp_shard_tick_async."""
        self.g._chk(_lib().rio_gp_shard_tick_async(self.g.handle))

    def tick_wait(self, cap=64):
        """Records of the ticks enqueued since the last wait, oldest first: this rank's counters + the global verdict."""
        buf, n = (ShardTickInfo * cap)(), C.c_uint32(0)
        self.g._chk(_lib().rio_gp_shard_tick_wait(self.g.handle, buf, cap, C.byref(n)))
        out = []
        for k in range(min(cap, int(n.value))):
            d = buf[k].local.as_dict()
            d.update(cut_nodes=int(buf[k].cut_nodes), spill_rows=int(buf[k].spill_rows), slow_path=int(buf[k].slow_path),
                     rounds_run=int(buf[k].rounds_run))
            out.append(d)
        return out

    def sync(self):
        self.g.sync()


class DistExchange:
    """The data path's only collective: all-gather of one small record per rank through
    torch.distributed ("nccl" = RCCL over xGMI on GPUs; "gloo" in the CPU tests)."""

    def __init__(self, group=None, stage_through_host=False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # gloo cannot all-gather device tensors: tests that run several HIP shards as separate processes on ONE
        # GPU bounce the record through host memory (never the bench path)
        self.stage = stage_through_host

    def ranks(self, n_local):
        if n_local != 1:
            raise ValueError("one shard per process")
        return [self.rank]

    def all_gather(self, parts):
        inp = parts[0]
        out = torch.empty(self.world * inp.numel(), dtype=inp.dtype, device=inp.device)
        return self.all_gather_into(out, parts)

    def all_gather_into(self, out, parts):
        if self.stage and out.is_cuda:
            h_in = parts[0].cpu()  # synchronises with the producing stream
            h_out = torch.empty(out.numel(), dtype=out.dtype)
            self.dist.all_gather_into_tensor(h_out, h_in, group=self.group)
            out.copy_(h_out)
            return out
        self.dist.all_gather_into_tensor(out, parts[0], group=self.group)
        return out


class NativeRcclExchange:
    """The library issues the RCCL all-gathers itself (rio_gp_shard_comm_init / _solve_async / _exchange): one C call
    per fast-path solve, the exchange on the library's second stream.  torch.distributed is used once, to move the
    128-byte ncclUniqueId from rank 0 to the other ranks — a host in another language would use its own control
    channel for that."""

    def __init__(self, engine, group=None):
        import os
        import torch.distributed as dist
        self.e = engine
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self.path = path.encode() if os.path.exists(path) else None
        uid = (C.c_uint8 * 128)()
        if self.rank == 0:
            rc = _lib().rio_gp_shard_comm_unique_id(uid, self.path)
            if rc:
                raise rio_gp.ObjectPlacementError("Upstream", (_lib().rio_gp_last_error(None) or b"").decode(), rc)
        box = [bytes(uid)]
        dist.broadcast_object_list(box, src=0, group=group)
        buf = (C.c_uint8 * 128).from_buffer_copy(box[0])
        engine.g._chk(_lib().rio_gp_shard_comm_init(engine.g.handle, self.rank, self.world, buf, self.path))

    def ranks(self, n_local):
        if n_local != 1:
            raise ValueError("one shard per process")
        return [self.rank]

    def native_solve_async(self):
        self.e.g._chk(_lib().rio_gp_shard_solve_async(self.e.g.handle))

    def all_gather(self, parts):
        inp = parts[0]
        out = torch.empty(self.world * inp.numel(), dtype=inp.dtype, device=inp.device)
        return self.all_gather_into(out, parts)

    def all_gather_into(self, out, parts):
        inp = parts[0]
        self.e.g._chk(_lib().rio_gp_shard_exchange(self.e.g.handle, C.c_void_p(inp.data_ptr()), C.c_void_p(out.data_ptr()),
                                                  inp.numel()))
        return out


class P2PExchange(NativeRcclExchange):
    """Peer-to-peer windows over xGMI (rio_gp_shard_p2p_*): each record is stored straight into the peers' HBM and
    consumed behind sequence flags — no collective call on the data path.  torch.distributed only carries the
    64-byte IPC handles at set-up.  Raises if any rank cannot map its peers or the handshake fails, on EVERY rank
    (the outcome is agreed through the control channel), so that the caller can fall back consistently."""

    def __init__(self, engine, group=None):
        import torch.distributed as dist
        self.e = engine
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        h = engine.g.handle
        mine = (C.c_uint8 * 64)()
        rc = _lib().rio_gp_shard_p2p_export(h, self.world, mine)
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(mine) if rc == 0 else None, group=group)
        ok = rc == 0 and all(x is not None for x in handles)
        if ok:
            blob = (C.c_uint8 * (64 * self.world)).from_buffer_copy(b"".join(handles))
            ok = _lib().rio_gp_shard_p2p_connect(h, self.rank, self.world, blob) == 0
        flags = [None] * self.world
        dist.all_gather_object(flags, bool(ok), group=group)
        if not all(flags):
            _lib().rio_gp_shard_p2p_close(h)
            raise rio_gp.ObjectPlacementError("Upstream", "peer-to-peer windows unavailable on ranks %s: %s" % (
                [r for r, f in enumerate(flags) if not f], (_lib().rio_gp_last_error(h) or b"").decode()), 2)

    def close(self):
        _lib().rio_gp_shard_p2p_close(self.e.g.handle)


class LocalExchange:
    """G shards driven by ONE process on one device (tests, single-GPU what-if runs): the "all-gather" is
    a concatenation in rank order — the same records, the same reduction order."""

    def __init__(self, n_shards):
        self.world = n_shards
        self.rank = 0

    def ranks(self, n_local):
        if n_local != self.world:
            raise ValueError("LocalExchange drives all shards")
        return list(range(self.world))

    def all_gather(self, parts):
        return torch.cat(list(parts))

    def all_gather_into(self, out, parts):
        torch.cat(list(parts), out=out)
        return out


class ShardedSolver:
    """Sequences the phases of one row-sharded solve over this process's engines (normally one)."""

    def __init__(self, engines, exchange, spill_rounds=2, pipeline=False):
        """pipeline=True (one HIP engine per process): the all-gather and the global resolve of solve k run on a
        second stream and overlap the scan of solve k+1 — for back-to-back INDEPENDENT solves (bench.py); a tick
        that consumes the previous tick's commit cannot overlap and pays the exchange latency in full."""
        self.engines = list(engines)
        self.ex = exchange
        self.ranks = exchange.ranks(len(self.engines))
        self.R = exchange.world
        self.rounds = spill_rounds
        e0 = self.engines[0]
        self.X = [e.new_buffer(e.words1) for e in self.engines]
        self.Y = [e.new_buffer(e.words2) for e in self.engines]
        self.S = [e.new_buffer(len(STAT_KEYS)) for e in self.engines]
        self._ctx = getattr(e0, "ctx", None)
        # ring of gathered-X buffers for back-to-back asynchronous solves (bench)
        self.XG = [e0.new_buffer(self.R * e0.words1) for _ in range(4)]
        self._k = 0
        self.pipeline = bool(pipeline) and self._ctx is not None and len(self.engines) == 1
        if self.pipeline:
            self.side = torch.cuda.Stream(e0.device)
            self.XR = [e0.new_buffer(e0.words1) for _ in range(4)]   # ring of X records
            self.done = [None] * 4                                   # side-stream event of the solve that used slot q

    def _gather(self, parts, out=None):
        if self._ctx is not None:
            with self._ctx():
                return self.ex.all_gather(parts) if out is None else self.ex.all_gather_into(out, parts)
        return self.ex.all_gather(parts) if out is None else self.ex.all_gather_into(out, parts)

    def _copy_in(self, dst, src):
        if self._ctx is not None:
            with self._ctx():
                dst.copy_(src)
        else:
            dst.copy_(src)

    # -- fast path, asynchronous: scan -> all-gather X -> resolve; nothing waits on the host --
    def solve_async(self):
        if hasattr(self.ex, "native_solve_async"):
            return self.ex.native_solve_async()
        if self.pipeline:
            return self._solve_async_pipelined()
        xg = self.XG[self._k % len(self.XG)]
        self._k += 1
        for e, x in zip(self.engines, self.X):
            e.scan(x)
        self._gather(self.X, out=xg)
        for e, r in zip(self.engines, self.ranks):
            e.resolve(r, self.R, xg)

    def _solve_async_pipelined(self):
        e, q = self.engines[0], self._k % 4
        self._k += 1
        x, xg = self.XR[q], self.XG[q]
        if self.done[q] is not None:          # slot q's previous exchange must have consumed x / produced xg
            e.stream.wait_event(self.done[q])
        e.scan(x)                             # k_scan + k_resolve + pack on the engine's stream
        ready = e.stream.record_event()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            self.ex.all_gather_into(xg, [x])  # RCCL, ordered against the side stream
            e.resolve(self.ranks[0], self.R, xg, on_stream=self.side)
            self.done[q] = self.side.record_event()

    # -- finish the LAST enqueued solve: verdict, fix-up exchanges if it needs them, global stats --
    def solve_wait(self):
        vs = [e.verdict() for e in self.engines]
        v = vs[0]
        slow = v["cut_nodes"] > 0 or v["spill_rows"] > 0
        rounds_run = 0
        if slow:
            for e, y, ve in zip(self.engines, self.Y, vs):
                e.cut(ve["local_fixup"] > 0 and v["cut_nodes"] > 0, y)
            yg = self._gather(self.Y)
            pend = [e.merge(yg) for e in self.engines][0]
            for r in range(self.rounds):
                if pend[0] == 0:
                    break
                rounds_run += 1
                for e, y in zip(self.engines, self.Y):
                    e.spill(r, r + 1 == self.rounds, y)
                yg = self._gather(self.Y)
                pend = [e.merge(yg) for e in self.engines][0]
        local = [e.finish() for e in self.engines]
        for s_t, st in zip(self.S, local):
            self._copy_in(s_t, torch.tensor([st[k] for k in STAT_KEYS], dtype=torch.int64))
        sg = self._gather(self.S).cpu().numpy().reshape(self.R, len(STAT_KEYS))
        tot = sg.astype(np.uint64).sum(axis=0)
        stats = {k: int(tot[i]) for i, k in enumerate(STAT_KEYS)}
        stats.update(cut_nodes=v["cut_nodes"], slow_path=int(slow), rounds_run=rounds_run)
        return stats, v["n_slow"]

    def solve(self):
        self.solve_async()
        return self.solve_wait()[0]

    def commit(self):
        for e in self.engines:
            e.commit()

    def tick(self):
        st = self.solve()
        self.commit()
        return st

    # -- committed ticks with nothing waiting on the host (HIP engines over peer-to-peer windows) --
    def tick_async(self):
        for e in self.engines:
            e.tick_async()

    def tick_wait_local(self):
        """Wait for the ticks enqueued since the last wait on THIS process's engines and fetch their local records (the C
        call alone: no exchange between the ranks) — what a timing loop ends on; tick_reduce turns them into global counters."""
        return [e.tick_wait() for e in self.engines]

    def tick_wait(self):
        """Global counters of every tick enqueued since the last wait (oldest first): each rank's records, summed over the
        ranks with ONE all-gather for all the ticks."""
        return self.tick_reduce(self.tick_wait_local())

    def tick_reduce(self, local):
        n = len(local[0])
        if n == 0:
            return []
        nk = len(STAT_KEYS)
        per = max(1, self.engines[0].words1 // nk)      # ticks per exchange: a record fits a window row
        sg = np.zeros((n, nk), np.uint64)
        for lo in range(0, n, per):
            hi = min(n, lo + per)
            dev = [e.new_buffer((hi - lo) * nk) for e in self.engines]
            for d, recs in zip(dev, local):
                self._copy_in(d, torch.tensor([[r[k] for k in STAT_KEYS] for r in recs[lo:hi]], dtype=torch.int64).reshape(-1))
            sg[lo:hi] = self._gather(dev).cpu().numpy().reshape(self.R, hi - lo, nk).astype(np.uint64).sum(axis=0)
        out = []
        for k in range(n):
            st = {key: int(sg[k][i]) for i, key in enumerate(STAT_KEYS)}
            st.update(cut_nodes=local[0][k]["cut_nodes"], slow_path=local[0][k]["slow_path"], rounds_run=local[0][k]["rounds_run"])
            out.append(st)
        return out


def shard_bounds(n, n_ranks):
    """Contiguous, balanced row blocks: rank r owns [b[r], b[r+1])."""
    return [(r * n) // n_ranks for r in range(n_ranks + 1)]
