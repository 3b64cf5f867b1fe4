"""Randomised operation sequences through the C ABI against the CPU oracle, bit for bit after EVERY operation.

One scenario = a random table (sizes drawn around the kernels' thresholds: tile, workgroup and batch-size boundaries), random
capacities (unbounded / tight / zero), loads, affinities (nodes, NONE, RIO_GP_AFF_INACTIVE), liveness, the reference's
self-assignment switch on or off — then 12-30 operations drawn from everything the dense layer offers: committed and
uncommitted solves (synchronous, asynchronous streams), liveness flips, update / remove / lookup batches, clean_server(s),
place_pending at every batch-size regime (one workgroup, three launches, plain kernels, window-sorted), new loads /
affinities / capacities.  A second kind of scenario drives the ROW-SHARDED solve (G handles on the one device, random shard
boundaries with empty shards, streams of committed ticks with liveness changes) against the whole-table oracle.  The fix-up policies the product picks adaptively are also forced through the lab build's knobs, by
seed.  The seeds are fixed: a failure names the seed and the operation.

    python tests/test_gpu_fuzz.py <seconds> [first_seed]     # a longer campaign (tools/round5_pass.sh runs one)
"""
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NONE = 0xFFFFFFFF
AFF_INACTIVE = 0xFFFFFFFE
INF = 0xFFFFFFFFFFFFFFFF

_SIZES = (1, 2, 3, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 16383, 16384, 16385, 65536, 262143, 262144, 262145)
_BATCHES = (1, 2, 4, 5, 255, 256, 257, 1000, 1024, 1025, 4095, 4096, 4097, 20000, 65535, 65536, 65537, 131072)


def _pick(rng, table, hi):
    """A size: half the time one of the boundary values (below hi), else log-uniform in [1, hi]."""
    if rng.random() < 0.5:
        c = [v for v in table if v <= hi]
        return int(c[rng.integers(len(c))])
    return int(np.exp(rng.uniform(0, np.log(hi))))


class Scenario:
    def __init__(self, gp, oracle, seed, big=False):
        self.gp, self.oracle, self.seed = gp, oracle, seed
        rng = self.rng = np.random.default_rng(0x5EED0000 + seed)
        self.n = n = _pick(rng, _SIZES + (524288, 1_000_000), 2_000_000 if big else 300_000)
        if big and rng.random() < 0.3:   # a campaign spends a third of its scenarios on tables of 10^5 .. 2 x 10^6 rows
            self.n = n = int(np.exp(rng.uniform(np.log(100_000), np.log(2_000_000))))
        self.m = m = int(rng.choice([1, 2, 3, 7, 31, 32, 33, 64, 100, 255, 256, 257, 1000, 1024, 1025, 4096, 5000, 8191, 8192]))
        self.rounds = int(rng.choice([1, 2, 2, 2, 3]))
        self.sa = bool(rng.random() < 0.25)
        self.flags = gp.CFG_REF_SELF_ASSIGN if self.sa else 0
        self.oflags = oracle.REF_SELF_ASSIGN if self.sa else 0
        kind = rng.integers(4)
        self.load = {0: lambda: rng.integers(0, 30, n), 1: lambda: np.ones(n), 2: lambda: rng.zipf(1.3, n).clip(0, 60000),
                     3: lambda: rng.integers(0, 2, n) * rng.integers(0, 1000, n)}[int(kind)]().astype(np.uint32)
        self.aff = rng.integers(0, m, n).astype(np.uint32)
        r = rng.random(n)
        self.aff[r < 0.08] = NONE
        self.aff[(r >= 0.08) & (r < 0.11)] = AFF_INACTIVE
        if rng.random() < 0.3:      # a hot affinity node
            self.aff[rng.random(n) < 0.4] = int(rng.integers(m))
        self.cap = self._caps()
        self.alive = (rng.random(m) > rng.choice([0.0, 0.1, 0.5])).astype(np.uint8)
        start = rng.integers(3)
        if start == 0:
            self.ref = np.full(n, NONE, np.uint32)
        else:
            self.ref = rng.integers(0, m, n).astype(np.uint32)
            self.ref[rng.random(n) < (0.02 if start == 1 else 0.5)] = NONE
        self.lab = seed % 3 == 2
        # half of the lab scenarios: the quiet asynchronous ticks overlap / chain whatever the table's size (the product's rule
        # needs 2^22 rows) — their scans alternate between two streams and hand the rows over workgroup by workgroup
        self.chain_small = self.lab and (seed // 3) % 2 == 1
        if self.chain_small:
            os.environ["RIO_GP_OVERLAP_MIN_ROWS"] = "1"
            if (seed // 6) % 2:   # ... in the form big tables use (k_resolve on the side stream behind the scan's stop event)
                os.environ["RIO_GP_CHAIN_INLINE_BELOW"] = "0"
        try:
            g = self.g = gp.GpuPlacement(n, m, spill_rounds=self.rounds, flags=self.flags, lab=self.lab)
        finally:
            os.environ.pop("RIO_GP_OVERLAP_MIN_ROWS", None)
            os.environ.pop("RIO_GP_CHAIN_INLINE_BELOW", None)
        g.set_nodes(self.cap, self.alive, m=m)
        g.set_objects(n, self.load, self.aff)
        g.set_assign(self.ref)
        if self.lab:
            k = (seed // 3) % 6
            g.set_compact(("never", "always", "never", "always", "auto", "auto")[k], cut_pack=("never", "never", "always", "never", "auto", "auto")[k],
                          inc=("auto", "always", "never", "never", "always", "never")[k],
                          cut_apply=("auto", "never", "always")[(seed // 54) % 3])   # (the one-pass / the two-pass whole-table fix-up)
            g.set_speculate(("never", "always", "auto")[(seed // 18) % 3])
        self.log = []
        self.count = {}

    def _caps(self):
        rng, m = self.rng, self.m
        total = int(self.load.sum())
        k = rng.integers(4)
        if k == 0:
            return np.full(m, INF, np.uint64)
        if k == 1:
            return rng.integers(0, total // m + 5, m).astype(np.uint64)            # tight: about half fits
        if k == 2:
            c = rng.integers(0, 3 * (total // m) + 5, m).astype(np.uint64)          # roomy, some zero, some unbounded
            c[rng.random(m) < 0.1] = 0
            c[rng.random(m) < 0.1] = INF
            return c
        return np.full(m, (total * 5 // 4) // m + 1, np.uint64)                     # BASELINE's 1.25x

    # ---- checks -------------------------------------------------------------------------------------------------------
    def check_table(self, what):
        got = self.g.get_assign()
        assert np.array_equal(got, self.ref), (self.seed, what, self.log[-6:], np.flatnonzero(got != self.ref)[:8])
        used = self.oracle.recompute_used(self.ref, self.load, self.m)
        gu = self.g.get_nodes()[2]
        assert np.array_equal(gu, used), (self.seed, what, self.log[-6:], np.flatnonzero(gu != used)[:8])

    def otick(self):
        return self.oracle.tick(self.ref, self.load, self.aff, self.cap, self.alive, self.rounds, self.oflags)

    # ---- operations ---------------------------------------------------------------------------------------------------
    def op_tick(self):
        st = self.g.tick()
        want, used, ost = self.otick()
        self.ref = want
        assert st == ost, (self.seed, "tick", self.log[-6:], st, ost)

    def op_solve(self):
        st = self.g.solve()
        want, used, ost = self.otick()
        got = self.g.get_solved()
        assert st == ost, (self.seed, "solve", self.log[-6:], st, ost)
        assert np.array_equal(got, want), (self.seed, "solve", self.log[-6:], np.flatnonzero(got != want)[:8])
        if self.rng.random() < 0.5:
            self.g.commit()
            self.ref = want

    def op_async(self):
        k = int(self.rng.integers(1, 5))
        quiet_run = self.chain_small and self.rng.random() < 0.5   # a longer stream without changes: verdicts land, the ticks chain
        if quiet_run:
            k += 6
        want_st = []
        for i in range(k):
            if i and not quiet_run and self.rng.random() < 0.5:
                self.op_flip()
            self.g.tick_async()
            self.ref, used, ost = self.otick()
            want_st.append(ost)
            if self.rng.random() < 0.3 or (quiet_run and i < 3):
                time.sleep(0.002)
        got = self.g.tick_wait()
        assert got == want_st, (self.seed, "tick_async", self.log[-6:], got, want_st)

    def op_flip(self):
        rng, m = self.rng, self.m
        k = rng.integers(3)
        if k == 0:
            j = int(rng.integers(m))
            self.alive[j] ^= 1
            self.g.set_alive(j, int(self.alive[j]))
        elif k == 1:
            self.alive = (rng.random(m) > 0.1).astype(np.uint8)
            self.g.set_alive_all(self.alive)
        else:
            self.alive = np.ones(m, np.uint8)
            self.g.set_alive_all(self.alive)

    def _idx(self, k=None, big_ok=True):
        rng, n = self.rng, self.n
        if k is None:
            k = _pick(rng, _BATCHES, 140000 if n >= 20000 else 20000)
            if big_ok and n >= 20000 and rng.random() < 0.2:
                k = int(rng.integers(262144, 600000))     # the window-sorted forms
                self.count["batches >= 2^18"] = self.count.get("batches >= 2^18", 0) + 1
        mode = rng.integers(3)
        if mode == 0:
            return rng.integers(0, n, k).astype(np.uint32)
        if mode == 1:                                     # a narrow range: many duplicates
            lo = int(rng.integers(n))
            return (lo + rng.integers(0, max(1, min(n - lo, k // 2 + 1)), k)).astype(np.uint32)
        return (np.arange(k, dtype=np.uint64) * 7 % n).astype(np.uint32)

    def op_update(self):
        idx = self._idx()
        node = self.rng.integers(0, self.m, idx.size).astype(np.uint32)
        node[self.rng.random(idx.size) < 0.05] = NONE      # Option::None deletes (local.rs:36-37)
        self.g.update_batch(idx, node)
        self.oracle.update_batch(self.ref, self.m, idx, node)

    def op_remove(self):
        idx = self._idx()
        self.g.remove_batch(idx)
        self.oracle.remove_batch(self.ref, idx)

    def op_lookup(self):
        idx = self._idx()
        got = self.g.lookup_batch(idx)
        assert np.array_equal(got, self.oracle.lookup_batch(self.ref, idx)), (self.seed, "lookup", self.log[-6:])

    def op_clean(self):
        rng, m = self.rng, self.m
        if rng.random() < 0.5:
            j = int(rng.integers(m))
            ev = self.g.clean_server(j)
            want = int((self.ref == j).sum())
            self.ref[self.ref == j] = NONE
        else:
            dead = sorted(set(int(x) for x in rng.integers(0, m, int(rng.integers(1, 8)))))
            ev = self.g.clean_servers(dead)
            want = self.oracle.clean_servers(self.ref, m, dead)
        assert ev == want, (self.seed, "clean", self.log[-6:], ev, want)

    def op_place(self):
        idx = self._idx()
        req = self.rng.integers(0, self.m, idx.size).astype(np.uint32)
        if self.rng.random() < 0.3:
            req[:] = int(self.rng.integers(self.m))
        used = self.oracle.recompute_used(self.ref, self.load, self.m)
        if self.rng.random() < 0.3:   # the same call over device-resident arrays (validated on the device; sometimes not 16-byte aligned)
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
            from hipbuf import DevBuf
            off = int(self.rng.integers(0, 2)) * int(self.rng.integers(1, 4))
            pad = np.zeros(off, np.uint32)
            d_idx, d_req = DevBuf(np.concatenate([pad, idx])), DevBuf(np.concatenate([pad, req]))
            d_node, d_flag = DevBuf(nbytes=4 * (idx.size + off)), DevBuf(nbytes=4 * (idx.size + off))
            self.g.place_pending_dev(idx.size, d_idx.ptr + 4 * off, d_req.ptr + 4 * off, d_node.ptr + 4 * off, d_flag.ptr + 4 * off)
            node, flag = d_node.to_host()[off:], d_flag.to_host()[off:]
            for x in (d_idx, d_req, d_node, d_flag):
                x.free()
            self.count["place_dev"] = self.count.get("place_dev", 0) + 1
        else:
            node, flag = self.g.place_pending(idx, req)
        wnode, wflag = self.oracle.place_pending(self.ref, self.load, self.cap, self.alive, used, idx, req, self.rounds, self.oflags)
        bad = np.flatnonzero((node != wnode) | (flag != wflag))
        assert bad.size == 0, (self.seed, "place_pending", self.log[-6:], idx.size, bad[:8], node[bad[:8]], wnode[bad[:8]], flag[bad[:8]], wflag[bad[:8]])

    def op_mixed(self):
        """rio_gp_mixed_batch: up to 256 entries of each kind, one round trip — against the oracle's four calls in order."""
        rng = self.rng
        def some():
            return self._idx(k=int(_pick(rng, (1, 2, 4, 5, 31, 64, 200, 256), 256)), big_ok=False) if rng.random() < 0.7 else None
        ui, ri, li, pi = some(), some(), some(), some()
        un = pr = None
        if ui is not None:
            un = rng.integers(0, self.m, ui.size).astype(np.uint32)
            un[rng.random(ui.size) < 0.05] = NONE
        if pi is not None:
            pr = rng.integers(0, self.m, pi.size).astype(np.uint32)
        rc, lo, pn, pf = self.g.mixed_batch(update=None if ui is None else (ui, un), remove=ri, lookup=li,
                                            place=None if pi is None else (pi, pr))
        assert rc == [0, 0, 0, 0], (self.seed, "mixed rc", rc)
        if ui is not None:
            self.oracle.update_batch(self.ref, self.m, ui, un)
        if ri is not None:
            self.oracle.remove_batch(self.ref, ri)
        if li is not None:
            assert np.array_equal(lo, self.oracle.lookup_batch(self.ref, li)), (self.seed, "mixed lookup", self.log[-6:])
        if pi is not None:
            used = self.oracle.recompute_used(self.ref, self.load, self.m)
            wnode, wflag = self.oracle.place_pending(self.ref, self.load, self.cap, self.alive, used, pi, pr, self.rounds, self.oflags)
            bad = np.flatnonzero((pn != wnode) | (pf != wflag))
            assert bad.size == 0, (self.seed, "mixed place_pending", self.log[-6:], pi.size, bad[:8], pn[bad[:8]], wnode[bad[:8]])

    def op_attrs(self):
        idx = np.unique(self._idx(big_ok=False))
        load = self.rng.integers(0, 500, idx.size).astype(np.uint32)
        aff = self.rng.integers(0, self.m, idx.size).astype(np.uint32)
        aff[self.rng.random(idx.size) < 0.1] = NONE
        self.load[idx] = load
        self.aff[idx] = aff
        self.g.set_object_attrs(idx, load, aff)

    def op_caps(self):
        self.cap = self._caps()
        self.g.set_nodes(self.cap, self.alive, m=self.m)

    OPS = (("tick", 5), ("solve", 2), ("async", 3), ("flip", 4), ("update", 2), ("remove", 2), ("lookup", 1), ("clean", 2),
           ("place", 4), ("mixed", 3), ("attrs", 1), ("caps", 1))

    def run(self):
        names = [a for a, w in self.OPS for _ in range(w)]
        k = int(self.rng.integers(12, 31))
        try:
            for _ in range(k):
                op = names[int(self.rng.integers(len(names)))]
                self.log.append(op)
                self.count[op] = self.count.get(op, 0) + 1
                getattr(self, "op_" + op)()
                self.check_table(op)
            self.chained = self.g.chained_scans() if self.lab else 0
        finally:
            self.g.close()
        return k


@pytest.fixture(scope="module")
def gp():
    import rio_gp
    rio_gp.build()
    return rio_gp


@pytest.mark.parametrize("seed", range(int(os.environ.get("RIO_FUZZ_SEEDS", "36"))))
def test_random_operation_sequences(gp, oracle, seed):
    Scenario(gp, oracle, seed).run()


def _sharded_scenario(gp, oracle, seed):
    """The row-sharded solve as G handles on the one device, sequenced by the ShardedSolver the multi-GPU
    bench uses: random shard boundaries (empty and one-row shards included), a stream of committed ticks with liveness
    changes between them, every tick against the WHOLE-table oracle (rows, the global `used` on every rank, the counters)."""
    import sharded
    import torch
    rng = np.random.default_rng(0x5A4D0000 + seed)
    n = _pick(rng, _SIZES, 200_000)
    m = int(rng.choice([1, 2, 7, 33, 64, 257, 1024, 3000, 8192]))
    rounds = int(rng.choice([1, 2, 2, 3]))
    G = int(rng.choice([1, 2, 3, 5, 8]))
    load = (rng.integers(0, 30, n) if rng.random() < 0.5 else rng.zipf(1.3, n).clip(0, 60000)).astype(np.uint32)
    aff = rng.integers(0, m, n).astype(np.uint32)
    aff[rng.random(n) < 0.1] = NONE
    if rng.random() < 0.3:
        aff[rng.random(n) < 0.4] = int(rng.integers(m))
    total = int(load.sum())
    cap = [np.full(m, INF, np.uint64), rng.integers(0, total // m + 5, m).astype(np.uint64),
           np.full(m, (total * 5 // 4) // m + 1, np.uint64)][int(rng.integers(3))]
    alive = (rng.random(m) > rng.choice([0.0, 0.1, 0.5])).astype(np.uint8)
    ref = rng.integers(0, m, n).astype(np.uint32)
    ref[rng.random(n) < rng.choice([0.02, 0.5, 1.0])] = NONE
    cuts = sorted(int(x) for x in rng.integers(0, n + 1, G - 1))
    bounds = [0] + cuts + [n]
    stream = torch.cuda.Stream(torch.device("cuda", 0))
    engines = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        g = gp.GpuPlacement(max(hi - lo, 1), m, spill_rounds=rounds)
        g.set_nodes(cap, alive, m=m)
        g.set_objects(hi - lo, load[lo:hi], aff[lo:hi])
        if hi > lo:
            g.set_assign(ref[lo:hi])
        engines.append(sharded.HipShardEngine(g, 0, stream))
    sol = sharded.ShardedSolver(engines, sharded.LocalExchange(G), spill_rounds=rounds)
    try:
        for step in range(int(rng.integers(3, 7))):
            if step:
                k = rng.integers(3)
                if k == 0:
                    alive = (rng.random(m) > 0.1).astype(np.uint8)
                elif k == 1:
                    alive = np.ones(m, np.uint8)
                else:
                    alive[int(rng.integers(m))] ^= 1
                for e in engines:
                    e.g.set_alive_all(alive)
            st = sol.tick()
            ref, used, ost = oracle.tick(ref, load, aff, cap, alive, rounds)
            got = np.concatenate([e.g.get_assign() if e.g.num_objects else np.zeros(0, np.uint32) for e in engines])
            assert np.array_equal(got, ref), (seed, step, bounds, np.flatnonzero(got != ref)[:8])
            assert st == ost, (seed, step, st, ost)
            for e in engines:
                assert np.array_equal(e.g.get_nodes()[2], used), (seed, step)
    finally:
        for e in engines:
            e.g.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("RIO_FUZZ_SHARD_SEEDS", "24"))))
def test_random_sharded_tick_streams(gp, oracle, seed):
    _sharded_scenario(gp, oracle, seed)


if __name__ == "__main__":
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import json
    try:   # (two HIP runtimes in the process: torch's has to come up first — see tests/conftest.py)
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    import pyoracle
    import rio_gp
    rio_gp.build()
    pyoracle.build()
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0, ops, first, cov = time.time(), 0, seed, {}
    while time.time() - t0 < budget:
        if seed % 10 == 9:   # every tenth scenario: a row-sharded tick stream (G handles on the device)
            _sharded_scenario(rio_gp, pyoracle, seed)
            cov["row-sharded tick streams"] = cov.get("row-sharded tick streams", 0) + 1
            seed += 1
            continue
        sc = Scenario(rio_gp, pyoracle, seed, big=True)
        ops += sc.run()
        for k, v in sc.count.items():
            cov[k] = cov.get(k, 0) + v
        for k, on in (("tables >= 10^5 rows", sc.n >= 100_000), ("tables >= 2^19 rows", sc.n >= 524288), ("self-assign", sc.sa),
                      ("forced policies (lab build)", sc.lab), ("scenarios with chained quiet ticks", sc.chained > 0)):
            cov[k] = cov.get(k, 0) + int(on)
        cov["chained scans"] = cov.get("chained scans", 0) + sc.chained
        seed += 1
    print(json.dumps({"scenarios": seed - first, "first_seed": first, "operations_checked": ops, "seconds": round(time.time() - t0, 1),
                      "mismatches": 0, "coverage": cov}))
