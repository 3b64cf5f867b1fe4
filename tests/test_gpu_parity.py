"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs — bit-exact (integer/index work).  Run on the GPU box: pytest -m gpu."""
import json
import os

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

NONE = 0xFFFFFFFF
INF = 0xFFFFFFFFFFFFFFFF
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sql_backend_golden.json")


@pytest.fixture(scope="module")
def gp():
    import rio_gp
    rio_gp.build()
    return rio_gp


# (packed fix-up, speculative enqueue) — the policies the product picks adaptively, forced through the lab build's knobs:
# fix-up over the whole table with the cut search as a launch of its own | over the pending rows k_scan packed, with the
# search inside k_resolve | whole table with round 0 of k_fill packing the rows that go on to the water-fill; enqueued after
# the host has read the verdict | speculatively behind k_resolve (every kernel guards itself on the device)
FIXUP_VARIANTS = (("never", "never"), ("always", "never"), ("never", "always"), ("always", "always"),
                  ("cutpack", "never"), ("cutpack", "always"),
                  # the two-pass whole-table fix-up (k_cut_find, then the re-marking inside round 0 of k_fill): what the request
                  # path's virtual table and the row-sharded solve still run, kept under test on the real table too
                  ("twopass", "never"), ("twopass-cutpack", "always"),
                  # ... and k_cut_apply where the adaptive rule would not pick it (a solve that does not pack at the cut pass)
                  ("onepass", "always"))


def apply_variant(g, mode):
    """(packed fix-up | whole-table fix-up by k_cut_apply, packing or not | the two-pass form, packing or not, speculation)"""
    kind, spec = mode
    if kind == "cutpack":
        g.set_compact("never", cut_pack="always")
    elif kind == "twopass":
        g.set_compact("never", cut_pack="never", cut_apply="never")
    elif kind == "twopass-cutpack":
        g.set_compact("never", cut_pack="always", cut_apply="never")
    elif kind == "onepass":
        g.set_compact("never", cut_pack="never", cut_apply="always")
    else:
        g.set_compact(kind, cut_pack="never")
    g.set_speculate(spec)


def _mk(gp, n, m, load, aff, cap, alive, cur=None, rounds=2, lab=False):
    g = gp.GpuPlacement(max(n, 1), max(m, 1), spill_rounds=rounds, lab=lab)
    g.set_nodes(cap, alive, m=m)
    g.set_objects(n, load, aff)
    if cur is not None and n:
        g.set_assign(cur)
    return g


def _check_tick(gp, oracle, cur, load, aff, cap, alive, rounds=2):
    n, m = len(cur), len(cap)
    want, used, ost = oracle.tick(cur, load, aff, cap, alive, rounds)
    # Every policy of the fix-up must give the same bytes — and so must the product library left to itself (first variant).
    for mode in (None,) + FIXUP_VARIANTS:
        g = _mk(gp, n, m, load, aff, cap, alive, cur, rounds, lab=mode is not None)
        if mode is not None:
            apply_variant(g, mode)
        st = g.solve()
        got = g.get_solved()
        assert np.array_equal(got, want), (mode, np.flatnonzero(got != want)[:10])
        assert st == ost, mode
        assert np.array_equal(g.get_assign(), cur)  # solve does not publish
        g.commit()
        assert np.array_equal(g.get_assign(), want)
        assert np.array_equal(g.get_nodes()[2], used)
        g.close()
    return ost


def test_backend_is_hip(gp):
    g = gp.GpuPlacement(16, 2)
    assert g.backend() == "hip:gfx950"
    g.close()


def test_host_waits_without_the_runtime(gp, oracle):
    """Synchronous calls spin on a word their last kernel stores into mapped pinned memory (completion word of the
    single-workgroup calls, sequence number in the pinned rows of k_resolve / the last water-fill round, the tagged total
    of k_clean) instead of hipStreamSynchronize.  Thousands of back-to-back calls of every kind, results checked: a word
    that arrives before the data it guards, or a stale sequence number, shows up as a wrong answer here."""
    rng = np.random.default_rng(77)
    n, m = 50_000, 64
    load = rng.integers(1, 50, n).astype(np.uint32)
    g = gp.GpuPlacement(n, m)
    g.set_nodes(np.full(m, 1 << 40, np.uint64), np.ones(m, np.uint8))
    g.set_objects(n, load, None)
    ref = np.full(n, NONE, np.uint32)
    for step in range(1500):
        k = int(rng.integers(1, 7))
        idx = rng.integers(0, n, k).astype(np.uint32)
        node = rng.integers(0, m, k).astype(np.uint32)
        g.update_batch(idx, node)
        oracle.update_batch(ref, m, idx, node)
        q = rng.integers(0, n, int(rng.integers(1, 6))).astype(np.uint32)
        assert np.array_equal(g.lookup_batch(q), ref[q]), step
        if step % 7 == 0:
            rm = rng.integers(0, n, 3).astype(np.uint32)
            g.remove_batch(rm)
            oracle.remove_batch(ref, rm)
        if step % 50 == 0:
            j = int(rng.integers(m))
            assert g.clean_server(j) == int((ref == j).sum()), step
            ref[ref == j] = NONE
    assert np.array_equal(g.get_assign(), ref)
    g.close()
    # the probe kernels behind tools/sync_probe.py (lab build): both ways of waiting complete and take microseconds
    gl = gp.LabPlacement(1024, 4)
    for mode in (20, 21, 22, 23):
        assert 0 < gl.stream_probe(mode, 200) * 1000.0 < 200.0, mode
    gl.close()
    # synchronous ticks, alternating fast path and fix-up path: counters and tables of every tick against the oracle
    n, m = 300_000, 100
    cur, load, aff, cap, alive = _rand_case(np.random.default_rng(5), n, m, cap_scale=1.6, p_alive=1.0)
    g = gp.GpuPlacement(n, m, spill_rounds=2)
    g.set_nodes(cap, alive)
    g.set_objects(n, load, aff)
    g.set_assign(cur)
    for t in range(40):
        if t % 3 == 0:
            alive = (np.random.default_rng(100 + t).random(m) > 0.15).astype(np.uint8)
            g.set_alive_all(alive)
        cur, used, ost = oracle.tick(cur, load, aff, cap, alive, 2)
        assert g.tick() == ost, t
        assert np.array_equal(g.get_assign(), cur), t
    g.close()


# ---- reference known-answer tests, through the GPU (object_placement_backend.rs:11-34 etc.) ----

def test_backend_save_and_load_dense(gp):
    g = gp.GpuPlacement(8, 2)
    g.set_nodes(m=2, cap=None, alive=np.ones(2, np.uint8))
    g.set_objects(8)
    assert g.lookup_batch([1])[0] == NONE                 # no_placement
    g.update_batch([1], [0])                              # save
    assert g.lookup_batch([1])[0] == 0                    # load
    g.update_batch([1], [1])                              # upsert overwrites (sqlite.rs:149-193)
    assert g.lookup_batch([1])[0] == 1
    assert g.clean_server(1) == 1                         # clean_server
    assert g.lookup_batch([1])[0] == NONE
    g.update_batch([1], [0])
    g.update_batch([1], [NONE])                           # update(None) deletes (local.rs:36-37)
    assert g.lookup_batch([1])[0] == NONE
    g.remove_batch([1, 2, 2])                             # remove absent: no-op (local.rs:60-68)
    g.close()


def test_sql_golden_through_gpu(gp):
    """The reference's SQL semantics (golden file) replayed op by op through the HIP CRUD kernels."""
    doc = json.load(open(GOLD))
    for case in doc["cases"][:4]:
        keys, addrs = {}, {}
        for op in case["ops"]:
            if op[0] in ("update", "lookup", "remove"):
                keys.setdefault(op[1] + "." + op[2], len(keys))
            if op[0] == "update":
                addrs.setdefault(op[3], len(addrs))
            if op[0] == "clean_server":
                addrs.setdefault(op[1], len(addrs))
        names = {v: k for k, v in addrs.items()}
        g = gp.GpuPlacement(len(keys), len(addrs))
        g.set_nodes(m=len(addrs), alive=np.ones(len(addrs), np.uint8))
        g.set_objects(len(keys))
        got = []
        for op in case["ops"]:
            if op[0] == "lookup":
                v = int(g.lookup_batch([keys[op[1] + "." + op[2]]])[0])
                got.append(None if v == NONE else names[v])
            elif op[0] == "update":
                g.update_batch([keys[op[1] + "." + op[2]]], [addrs[op[3]]])
            elif op[0] == "remove":
                g.remove_batch([keys[op[1] + "." + op[2]]])
            else:
                g.clean_server(addrs[op[1]])
        assert got == case["expected_lookups"]
        g.close()


# ---- CRUD batches vs oracle ---------------------------------------------------------------------

@pytest.mark.parametrize("seed", [0, 1, 2])
def test_crud_batches_random(gp, oracle, seed):
    rng = np.random.default_rng(seed)
    n, m = int(rng.integers(100, 70000)), int(rng.integers(1, 300))
    load = rng.integers(0, 100, n).astype(np.uint32)
    g = gp.GpuPlacement(n, m)
    g.set_nodes(m=m, alive=np.ones(m, np.uint8))
    g.set_objects(n, load, None)
    ref = np.full(n, NONE, np.uint32)
    for step in range(12):
        k = int(rng.integers(1, 30000))
        idx = rng.integers(0, n, k).astype(np.uint32)          # many duplicates: last writer wins
        node = rng.integers(0, m, k).astype(np.uint32)
        node[rng.random(k) < 0.1] = NONE
        g.update_batch(idx, node)
        assert oracle.update_batch(ref, m, idx, node) == 0
        q = rng.integers(0, n, 5000).astype(np.uint32)
        assert np.array_equal(g.lookup_batch(q), oracle.lookup_batch(ref, q))
        assert np.array_equal(g.get_nodes()[2], oracle.recompute_used(ref, load, m))
        rm = rng.integers(0, n, int(rng.integers(1, 3000))).astype(np.uint32)
        g.remove_batch(rm)
        oracle.remove_batch(ref, rm)
        assert np.array_equal(g.get_nodes()[2], oracle.recompute_used(ref, load, m))  # incremental == scratch
        dead = rng.choice(m, size=min(m, int(rng.integers(1, 4))), replace=False)
        ev = g.clean_servers(dead)
        assert ev == oracle.clean_servers(ref, m, dead)
        assert np.array_equal(g.get_assign(), ref)
        assert np.array_equal(g.get_nodes()[2], oracle.recompute_used(ref, load, m))
    g.close()


@pytest.mark.parametrize("seed", [10, 11])
def test_crud_micro_batches(gp, oracle, seed):
    """Batches of <= 256 entries take the one-launch paths over mapped pinned memory (k_update_small: last writer wins
    INSIDE the batch; remove with the incremental `used`; lookup): same bytes as the oracle, duplicates included."""
    rng = np.random.default_rng(seed)
    n, m = 3000, 17
    load = rng.integers(0, 100, n).astype(np.uint32)
    g = gp.GpuPlacement(n, m)
    g.set_nodes(m=m, alive=np.ones(m, np.uint8))
    g.set_objects(n, load, None)
    ref = np.full(n, NONE, np.uint32)
    for step in range(60):
        k = int(rng.choice([1, 2, 3, 4, 5, 7, 64, 255, 256]))   # <= 4: requests in the kernel arguments
        idx = rng.integers(0, 40 if step % 3 == 0 else n, k).astype(np.uint32)   # every third step: heavy duplication
        node = rng.integers(0, m, k).astype(np.uint32)
        node[rng.random(k) < 0.15] = NONE
        g.update_batch(idx, node)
        assert oracle.update_batch(ref, m, idx, node) == 0
        q = rng.integers(0, n, int(rng.choice([1, 2, 3, 4, 5, int(rng.integers(1, 257))]))).astype(np.uint32)
        assert np.array_equal(g.lookup_batch(q), oracle.lookup_batch(ref, q))
        assert np.array_equal(g.get_nodes()[2], oracle.recompute_used(ref, load, m))
        rm = rng.integers(0, 60 if step % 4 == 0 else n, int(rng.choice([1, 3, 4, 5, int(rng.integers(1, 257))]))).astype(np.uint32)
        g.remove_batch(rm)
        oracle.remove_batch(ref, rm)
        assert np.array_equal(g.get_nodes()[2], oracle.recompute_used(ref, load, m))  # incremental == scratch
        assert np.array_equal(g.get_assign(), ref)
    g.close()


def test_update_remove_batch_sizes_around_every_path_boundary(gp, oracle):
    """update_batch / remove_batch from host buffers: one workgroup up to 256 entries, mapped pinned memory with a completion
    word up to 16 384, staging copies beyond — with duplicates and deletions inside every batch, `used` maintained."""
    rng = np.random.default_rng(12)
    n, m = 200_000, 33
    load = rng.integers(0, 50, n).astype(np.uint32)
    g = gp.GpuPlacement(n, m)
    g.set_nodes(m=m, alive=np.ones(m, np.uint8))
    g.set_objects(n, load, None)
    ref = np.full(n, NONE, np.uint32)
    for k in (256, 257, 1000, 4096, 16383, 16384, 16385, 50_000, 300, 257):
        idx = rng.integers(0, n // 3, k).astype(np.uint32)          # plenty of duplicates: the last writer wins
        node = rng.integers(0, m, k).astype(np.uint32)
        node[rng.random(k) < 0.1] = NONE
        g.update_batch(idx, node)
        assert oracle.update_batch(ref, m, idx, node) == 0
        assert np.array_equal(g.get_assign(), ref), k
        assert np.array_equal(g.get_nodes()[2], oracle.recompute_used(ref, load, m)), k
        rm = rng.integers(0, n // 3, max(1, k // 2)).astype(np.uint32)
        g.remove_batch(rm)
        oracle.remove_batch(ref, rm)
        assert np.array_equal(g.get_assign(), ref), k
        assert np.array_equal(g.get_nodes()[2], oracle.recompute_used(ref, load, m)), k
    g.close()


def test_lookup_batch_sizes_around_every_path_boundary(gp, oracle):
    """lookup_batch takes the one-workgroup kernel up to 256 entries (requests in the kernel arguments up to 4), mapped pinned
    memory with a several-workgroup completion word up to 16 384, staging copies beyond: the sizes on both sides of every
    boundary, twice each (the completion protocol resets its ticket itself)."""
    rng = np.random.default_rng(3)
    n, m = 500_000, 77
    ref = rng.integers(0, m, n).astype(np.uint32)
    ref[rng.random(n) < 0.3] = NONE
    g = gp.GpuPlacement(n, m)
    g.set_nodes(m=m, alive=np.ones(m, np.uint8))
    g.set_objects(n)
    g.set_assign(ref)
    for k in (1, 4, 5, 255, 256, 257, 1000, 1023, 1024, 1025, 4097, 16383, 16384, 16385, 70_000) * 2:
        q = rng.integers(0, n, k).astype(np.uint32)
        assert np.array_equal(g.lookup_batch(q), ref[q]), k
    with pytest.raises(gp.ObjectPlacementError):
        g.lookup_batch(np.array([0] * 300 + [n], np.uint32))   # an index out of range in a medium batch: refused, nothing read
    g.close()


@pytest.mark.parametrize("seed,n,m,k", [(0, 2_000_003, 300, 3_000_000), (1, 600_000, 7, 262_144), (2, 10_000_000, 1024, 10_000_000)])
def test_crud_big_batches_partitioned_by_row_window(gp, oracle, seed, n, m, k):
    """update_batch / remove_batch of >= 2^18 entries take the window-partitioned kernels (k_part_bin, k_part_update,
    k_part_remove): sequential last-writer-wins among heavy duplicates, deletes (node NONE), the incremental `used`
    vector of remove, the row-lifecycle column — same bytes as the oracle and as the plain per-entry kernels."""
    rng = np.random.default_rng(4400 + seed)
    load = rng.integers(0, 100, n).astype(np.uint32)
    g = gp.GpuPlacement(n, m, flags=gp.CFG_ROW_LIFECYCLE)
    g.set_nodes(m=m, alive=np.ones(m, np.uint8))
    g.set_objects(n, load, None)
    plain = gp.LabPlacement(n, m, flags=gp.CFG_ROW_LIFECYCLE)
    plain.set_compact("auto", partitioned_crud=False)
    plain.set_nodes(m=m, alive=np.ones(m, np.uint8))
    plain.set_objects(n, load, None)
    # the product sorts these batches in chunks of 8 192 entries; `small` (the name is history): the 16 384-entry form that
    # request batches of 4 M and more take, forced through the lab build's knob (process-wide in that library: reset below)
    small = gp.LabPlacement(n, m, flags=gp.CFG_ROW_LIFECYCLE)
    small.set_nodes(m=m, alive=np.ones(m, np.uint8))
    small.set_objects(n, load, None)
    gp.lab_lib().rio_gp_debug_set_part_shift(14 | 0x80)
    ref = np.full(n, NONE, np.uint32)
    life = np.zeros(n, bool)
    for step in range(2):
        idx = rng.integers(0, n, k).astype(np.uint32)
        idx[: k // 4] = rng.integers(0, 5000, k // 4).astype(np.uint32)       # a quarter of the batch fights over 5 000 rows
        node = rng.integers(0, m, k).astype(np.uint32)
        node[rng.random(k) < 0.1] = NONE
        for h in (g, plain, small):
            h.update_batch(idx, node)
        assert oracle.update_batch(ref, m, idx, node) == 0
        got = g.get_assign()
        assert np.array_equal(got, ref), np.flatnonzero(got != ref)[:10]
        assert np.array_equal(plain.get_assign(), ref) and np.array_equal(small.get_assign(), ref)
        life[idx] = ref[idx] != NONE
        assert np.array_equal(g.get_objects()[1] != gp.AFF_INACTIVE, life)
        assert np.array_equal(g.get_nodes()[2], oracle.recompute_used(ref, load, m))   # rebuilt from scratch after raw updates
        rm = rng.integers(0, n, max(k // 2, 262_144)).astype(np.uint32)        # duplicates and absent rows included
        for h in (g, plain, small):
            h.remove_batch(rm)
        oracle.remove_batch(ref, rm)
        life[rm] = False
        assert np.array_equal(g.get_assign(), ref) and np.array_equal(plain.get_assign(), ref)
        assert np.array_equal(small.get_assign(), ref)
        assert np.array_equal(g.get_nodes()[2], oracle.recompute_used(ref, load, m))   # incremental == scratch
        assert np.array_equal(small.get_nodes()[2], g.get_nodes()[2])
        assert np.array_equal(g.get_objects()[1] != gp.AFF_INACTIVE, life)
        assert np.array_equal(small.get_objects()[1], g.get_objects()[1])
    gp.lab_lib().rio_gp_debug_set_part_shift(14)
    g.close()
    plain.close()
    small.close()


def test_crud_big_batch_dev_skips_invalid_entries(gp, oracle):
    """Device-resident batch with out-of-range entries: they are skipped and reported (RIO_GP_EINVAL), the valid ones applied —
    the partitioned kernels keep the contract of the plain ones."""
    from hipbuf import DevBuf   # device arrays through hipMalloc / hipMemcpy (tools/hipbuf.py): no torch in the loop
    n, m, k = 1_000_000, 40, 400_000
    rng = np.random.default_rng(9)
    idx = rng.integers(0, n, k).astype(np.uint32)
    node = rng.integers(0, m, k).astype(np.uint32)
    idx[[5, 77777]] = n + 3
    node[[9, 12345]] = m
    g = gp.GpuPlacement(n, m)
    g.set_nodes(m=m, alive=np.ones(m, np.uint8))
    g.set_objects(n)
    d_idx, d_node = DevBuf(idx), DevBuf(node)
    import ctypes as C
    rc = gp.lib().rio_gp_update_batch_dev(g.handle, k, C.c_void_p(d_idx.ptr), C.c_void_p(d_node.ptr))
    assert rc == gp.EINVAL
    ok = (idx < n) & (node < m)
    ref = np.full(n, NONE, np.uint32)
    assert oracle.update_batch(ref, m, idx[ok], node[ok]) == 0
    assert np.array_equal(g.get_assign(), ref)
    rc = gp.lib().rio_gp_remove_batch_dev(g.handle, k, C.c_void_p(d_idx.ptr))
    assert rc == gp.EINVAL
    oracle.remove_batch(ref, idx[idx < n])
    assert np.array_equal(g.get_assign(), ref)
    g.close()
    d_idx.free()
    d_node.free()


def test_invalid_arguments_are_unknown_errors(gp):
    g = gp.GpuPlacement(10, 2)
    g.set_nodes(m=2, alive=np.ones(2, np.uint8))
    g.set_objects(10)
    for call in (lambda: g.lookup_batch([10]), lambda: g.update_batch([0], [2]), lambda: g.remove_batch([99]),
                 lambda: g.place_pending([0], [2]), lambda: g.set_alive(5, 1)):
        with pytest.raises(gp.ObjectPlacementError) as e:
            call()
        assert e.value.kind == "Unknown" and e.value.rc == gp.EINVAL
    assert np.all(g.get_assign() == NONE)       # nothing was mutated
    assert g.clean_server(7) == 0               # unknown address: retain() removes nothing
    g.close()
    # the lab build's knobs reject what they do not know (modes are 0 | 1 | 2 in every field)
    gl = gp.LabPlacement(10, 2)
    for bad in (3, 3 << 5, 7, 1 << 13):
        assert gp.lab_lib().rio_gp_debug_set_compact(gl.handle, bad) == gp.EINVAL
    assert gp.lab_lib().rio_gp_debug_set_speculate(gl.handle, 3) == gp.EINVAL
    assert gp.lab_lib().rio_gp_debug_set_compact(gl.handle, 2 | 16 | (1 << 5)) == 0   # never | plain CRUD | cut-pass packing always
    gl.close()
    assert not hasattr(gp.lib(), "rio_gp_debug_set_compact")   # and the product library has none of them


# ---- whole-table solve ---------------------------------------------------------------------------

def test_tick_known_answers(gp, oracle):
    # the hand-computed cases of tests/test_oracle_solver.py, through the kernels
    st = _check_tick(gp, oracle, np.full(4, NONE, np.uint32), np.array([60, 50, 10, 10], np.uint32),
                     np.zeros(4, np.uint32), np.array([100, 1000], np.uint64), np.ones(2, np.uint8), rounds=1)
    assert st["claimed"] == 1 and st["spilled"] == 3
    cur = np.full(5, NONE, np.uint32)
    _check_tick(gp, oracle, cur, np.array([6, 4, 3, 5, 9], np.uint32), np.zeros(5, np.uint32),
                np.array([100, 5, 8, 8], np.uint64), np.array([0, 1, 1, 1], np.uint8), rounds=2)
    _check_tick(gp, oracle, np.array([0, 0, 1, NONE], np.uint32), np.array([10, 10, 5, 1], np.uint32),
                np.array([1, 1, 0, 0], np.uint32), np.array([5, 100], np.uint64), np.array([1, 0], np.uint8))


def _rand_case(rng, n, m, p_none=0.5, cap_scale=1.0, p_alive=0.85, skew=False, max_load=50):
    cur = rng.integers(0, m, n).astype(np.uint32)
    cur[rng.random(n) < p_none] = NONE
    load = rng.integers(0, max_load, n).astype(np.uint32)
    if skew:
        aff = np.minimum((rng.pareto(1.2, n)).astype(np.int64), m - 1).astype(np.uint32)
    else:
        aff = rng.integers(0, m, n).astype(np.uint32)
    aff[rng.random(n) < 0.03] = NONE
    alive = (rng.random(m) < p_alive).astype(np.uint8)
    cap = rng.integers(0, int(load.sum() * cap_scale / max(m, 1)) + 2, m).astype(np.uint64)
    return cur, load, aff, cap, alive


@pytest.mark.parametrize("seed", range(8))
def test_tick_random_small(gp, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    n, m = int(rng.integers(1, 5000)), int(rng.integers(1, 64))
    _check_tick(gp, oracle, *_rand_case(rng, n, m, cap_scale=float(rng.choice([0.3, 1.0, 2.5]))),
                rounds=int(rng.integers(1, 4)))


@pytest.mark.parametrize("seed,n,m,skew", [(1, 300_000, 1024, False), (2, 1_000_003, 256, False),
                                           (3, 777_777, 4096, True), (4, 65_536, 7, True), (5, 2_000_000, 8192, False)])
def test_tick_random_large(gp, oracle, seed, n, m, skew):
    rng = np.random.default_rng(2000 + seed)
    st = _check_tick(gp, oracle, *_rand_case(rng, n, m, cap_scale=1.2, skew=skew, max_load=70000))
    assert st["slow_path"] == 1


@pytest.mark.parametrize("seed,n,m", [(1, 4000, 9), (2, 300_000, 1024), (3, 1_000_000, 64)])
def test_tick_rows_that_are_not_objects(gp, oracle, seed, n, m):
    """Affinity RIO_GP_AFF_INACTIVE (row lifecycle): such a row is kept if it sits on a live node and takes no part
    otherwise — never claimed, never water-filled, not counted — in every implementation of the fix-up."""
    rng = np.random.default_rng(2500 + seed)
    cur, load, aff, cap, alive = _rand_case(rng, n, m, cap_scale=1.1, max_load=300)
    aff[rng.random(n) < 0.3] = 0xFFFFFFFE
    st = _check_tick(gp, oracle, cur, load, aff, cap, alive)
    assert st["n_objects"] < n and st["n_objects"] == st["kept"] + st["claimed"] + st["spilled"] + st["unplaced"]


def test_tick_fast_path_no_contention(gp, oracle):
    cfg = synth.config("c3", n_override=500_000)
    cfg["cap"] = synth.uniform_cap(cfg["load"], cfg["m"], headroom=2.5)  # few rows per node: keep every node under cap
    st = _check_tick(gp, oracle, cfg["cur"], cfg["load"], cfg["aff"], cfg["cap"], cfg["alive"])
    assert st["slow_path"] == 0 and st["claimed"] == cfg["n"] and st["unplaced"] == 0


def test_tick_all_one_node_worst_case(gp, oracle):
    """Every object claims node 0: one cut, one block holds it, everything else spills."""
    n, m = 400_000, 16
    rng = np.random.default_rng(5)
    load = rng.integers(1, 20, n).astype(np.uint32)
    cap = np.full(m, int(load.sum()) // 10, np.uint64)
    _check_tick(gp, oracle, np.full(n, NONE, np.uint32), load, np.zeros(n, np.uint32), cap, np.ones(m, np.uint8),
                rounds=3)


def test_tick_tiny_capacity_cuts_in_first_block(gp, oracle):
    """Free capacity ~0 everywhere: all M cuts fall into the first sub-chunks."""
    n, m = 300_000, 1024
    rng = np.random.default_rng(6)
    load = rng.integers(1, 9, n).astype(np.uint32)
    _check_tick(gp, oracle, np.full(n, NONE, np.uint32), load, rng.integers(0, m, n).astype(np.uint32),
                rng.integers(0, 40, m).astype(np.uint64), np.ones(m, np.uint8))


@pytest.mark.parametrize("n,m,maxcap,zero_loads", [(600_000, 4096, 60, False), (1_200_000, 8192, 30, True),
                                                      (3_000_000, 2048, 2000, True), (40_000_000, 2048, 30000, True)])
def test_tick_cuts_concentrated_in_few_blocks(gp, oracle, n, m, maxcap, zero_loads):
    """A nearly full cluster: every node's cut lies within its first claimants, so ONE workgroup owns hundreds or
    thousands of cuts (k_cut_fused: coarser sub-chunks so they share a pass, several groups when even that does not
    fit, pipelined per-node row search; at 40 M rows a sub-chunk is ~40 tiles, which takes the coarse group search first).
    Zero-load claimants ride along until the first overflow."""
    rng = np.random.default_rng(60 + m)
    load = rng.integers(0 if zero_loads else 1, 12, n).astype(np.uint32)
    cap = rng.integers(0, maxcap, m).astype(np.uint64)
    cap[rng.integers(0, m, m // 16)] = 0
    alive = np.ones(m, np.uint8)
    alive[rng.integers(0, m, m // 50)] = 0
    _check_tick(gp, oracle, np.full(n, NONE, np.uint32), load, rng.integers(0, m, n).astype(np.uint32), cap, alive)


def test_tick_edge_shapes(gp, oracle):
    one = np.ones(1, np.uint8)
    _check_tick(gp, oracle, np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint32),
                np.array([5], np.uint64), one)                                           # empty table
    _check_tick(gp, oracle, np.array([NONE], np.uint32), np.array([0], np.uint32), np.array([0], np.uint32),
                np.array([0], np.uint64), one)                                           # zero load fits zero cap
    _check_tick(gp, oracle, np.full(257, NONE, np.uint32), np.full(257, 0xFFFFFFFF, np.uint32),
                np.zeros(257, np.uint32), np.array([INF, 7], np.uint64), np.ones(2, np.uint8))  # max loads, inf cap
    _check_tick(gp, oracle, np.full(1000, NONE, np.uint32), np.ones(1000, np.uint32), np.zeros(1000, np.uint32),
                np.full(3, 10, np.uint64), np.zeros(3, np.uint8))                        # no live node at all


def test_tick_equals_reference_policy_capacity_infinite(gp, oracle):
    """cap = inf: the kernels reproduce service.rs:193-254 run object by object (string oracle)."""
    rng = np.random.default_rng(11)
    n, m = 3000, 12
    alive = np.ones(m, np.uint8)
    alive[[3, 7]] = 0
    cur = rng.integers(0, m, n).astype(np.uint32)
    cur[rng.random(n) < 0.5] = NONE
    live = np.flatnonzero(alive)
    aff = live[rng.integers(0, len(live), n)].astype(np.uint32)
    g = _mk(gp, n, m, np.ones(n, np.uint32), aff, np.full(m, INF, np.uint64), alive, cur)
    g.tick()
    got = g.get_assign()
    g.close()
    provider = oracle.LocalObjectPlacement()
    storage = oracle.LocalStorage()
    for j in range(m):
        ip, port = synth.node_address(j).split(":")
        storage.push(ip, port, bool(alive[j]))
    for i in range(n):
        if cur[i] != NONE:
            provider.update("Obj", str(i), synth.node_address(int(cur[i])))
    for i in range(n):
        want = oracle.get_or_create_placement(provider, storage, synth.node_address(int(aff[i])), "Obj", str(i))
        assert want == synth.node_address(int(got[i]))


@pytest.mark.parametrize("seed,n,m", [(0, 3000, 12), (1, 200_000, 64), (2, 700_001, 1024)])
def test_reference_self_assign_ticks_and_requests(gp, oracle, seed, n, m):
    """RIO_GP_CFG_REF_SELF_ASSIGN (service.rs:244-252: the first touch asks nobody whether the requester is an active member):
    pending rows claim their affinity node — requests their requester — whether or not it is alive, against its whole
    capacity; rows on dead nodes are still evicted, the water-fill still places on live nodes only.  Whole-table solves in
    every fix-up variant, committed ticks (the in-place scan included) and request batches through every path (one
    workgroup, three launches, general, window-sorted), with finite capacities so that dead nodes are cut too — against the
    oracle with the same flag; and with unbounded capacities against the string-level restatement of the reference itself."""
    rng = np.random.default_rng(6600 + seed)
    cur, load, aff, cap, alive = _rand_case(rng, n, m, p_none=0.4, cap_scale=1.1, p_alive=0.7, max_load=40)
    flags = gp.CFG_REF_SELF_ASSIGN
    want, used, ost = oracle.tick(cur, load, aff, cap, alive, 2, flags=oracle.REF_SELF_ASSIGN)
    assert int((alive[want[want != NONE]] == 0).sum()) > 0            # rows really are claimed onto dead nodes
    for mode in (None,) + FIXUP_VARIANTS:
        g = gp.GpuPlacement(n, m, flags=flags, lab=mode is not None)
        g.set_nodes(cap, alive)
        g.set_objects(n, load, aff)
        g.set_assign(cur)
        if mode is not None:
            apply_variant(g, mode)
        assert g.solve() == ost, mode
        assert np.array_equal(g.get_solved(), want), mode
        g.commit()
        assert np.array_equal(g.get_nodes()[2], used), mode
        # committed ticks with liveness flips behind it (the in-place scan when the rule picks it: rows on dead nodes are
        # evicted and claim their dead affinity nodes again)
        ref = want
        for t in range(3):
            mask = (np.random.default_rng(70 + t).random(m) < 0.75).astype(np.uint8)
            g.set_alive_all(mask)
            ref, u2, o2 = oracle.tick(ref, load, aff, cap, mask, 2, flags=oracle.REF_SELF_ASSIGN)
            assert g.tick() == o2, (mode, t)
            assert np.array_equal(g.get_assign(), ref) and np.array_equal(g.get_nodes()[2], u2), (mode, t)
        g.close()
    # request batches: requesters are ANY member, dead ones included
    g = gp.GpuPlacement(n, m, flags=flags)
    g.set_nodes(cap, alive)
    g.set_objects(n, load, None)
    ref = np.full(n, NONE, np.uint32)
    uref = np.zeros(m, np.uint64)
    for k in (3, 200, 900, 3000, 4096, 20_000, min(n, 300_000)):
        idx = rng.integers(0, n, k).astype(np.uint32)
        req = rng.integers(0, m, k).astype(np.uint32)
        node, flag = g.place_pending(idx, req)
        wnode, wflag = oracle.place_pending(ref, load, cap, alive, uref, idx, req, flags=oracle.REF_SELF_ASSIGN)
        assert np.array_equal(node, wnode) and np.array_equal(flag, wflag), k
        assert np.array_equal(g.get_assign(), ref) and np.array_equal(g.get_nodes()[2], uref), k
    g.close()
    # unbounded capacities, one request per object, inactive requesters that hold nothing when the batch starts: the
    # reference's own map, read back (tests/test_reference_port_readback.py has the reasoning)
    if n <= 200_000:
        dead = np.flatnonzero(alive == 0)
        senders = np.concatenate([np.flatnonzero(alive), dead[::2]])
        aff2 = senders[rng.integers(0, len(senders), n)].astype(np.uint32)
        cur2 = cur.copy()
        cur2[np.isin(cur2, dead[::2])] = NONE
        ones = np.ones(n, np.uint32)
        _, port = oracle.policy_readback(n, m, aff2, alive, cur2)
        g = gp.GpuPlacement(n, m, flags=flags)
        g.set_nodes(np.full(m, INF, np.uint64), alive)
        g.set_objects(n, ones, aff2)
        g.set_assign(cur2)
        g.tick()
        assert np.array_equal(g.get_assign(), port)
        g.close()


def test_reference_port_read_back_against_the_gpu_at_1m(gp, oracle):
    """ONE hop to the reference's own semantics at a size that means something: 1 M objects x 256 nodes, a warm table with
    unplaced objects, 15 % of the nodes dead.  The string-level restatement of LocalObjectPlacement +
    Service::get_or_create_placement (local.rs:22-68, service.rs:193-254) serves one request per object — objects found on
    dead servers have those servers cleaned and are first-touched on the requester — and is read back object by object;
    the GPU's committed tick over the same table (capacity unbounded, load 1: the solver IS the reference policy) must
    leave the same column.  (bench.py does this at 10 M rows on the cold table: parity.against_reference_port.)"""
    rng = np.random.default_rng(4242)
    n, m = 1_000_000, 256
    alive = (rng.random(m) < 0.85).astype(np.uint8)
    live = np.flatnonzero(alive)
    aff = live[rng.integers(0, len(live), n)].astype(np.uint32)
    cur = rng.integers(0, m, n).astype(np.uint32)
    cur[rng.random(n) < 0.2] = NONE
    load = np.ones(n, np.uint32)
    cap = np.full(m, INF, np.uint64)
    _, port = oracle.policy_readback(n, m, aff, alive, cur)
    g = _mk(gp, n, m, load, aff, cap, alive, cur)
    st = g.tick()
    got = g.get_assign()
    assert np.array_equal(got, port), np.flatnonzero(got != port)[:10]
    assert st["spilled"] == 0 and st["unplaced"] == 0 and st["evicted"] > 100_000 and st["kept"] > 500_000
    # a second tick over the committed table (the in-place scan's turf): liveness changes again, the port serves every
    # object again on its own map
    alive2 = alive.copy()
    alive2[live[:20]] = 0
    live2 = np.flatnonzero(alive2)
    aff2 = aff.copy()
    gone = alive2[aff] == 0
    aff2[gone] = live2[rng.integers(0, len(live2), int(gone.sum()))]
    g.set_object_attrs(np.flatnonzero(gone).astype(np.uint32), None, aff2[gone])
    g.set_alive_all(alive2)
    g.tick()
    _, port2 = oracle.policy_readback(n, m, aff2, alive2, port)
    assert np.array_equal(g.get_assign(), port2)
    g.close()


def test_solve_is_deterministic_and_idempotent(gp, oracle):
    rng = np.random.default_rng(21)
    cur, load, aff, cap, alive = _rand_case(rng, 600_000, 512, cap_scale=1.1, max_load=1000)
    g = _mk(gp, len(cur), len(cap), load, aff, cap, alive, cur)
    g.solve()
    a = g.get_solved()
    for _ in range(3):
        g.solve()
        assert np.array_equal(g.get_solved(), a)           # same bytes run after run
    g.commit()
    g.tick()
    b = g.get_assign()
    placed = a != NONE
    assert np.array_equal(b[placed], a[placed])              # a second tick moves nothing that was placed
    g.close()


def test_async_solves_match_sync(gp, oracle):
    cfg = synth.config("c3", n_override=300_000)
    cfg["cap"] = synth.uniform_cap(cfg["load"], cfg["m"], headroom=3.0)
    g = _mk(gp, cfg["n"], cfg["m"], cfg["load"], cfg["aff"], cfg["cap"], cfg["alive"])
    for _ in range(5):
        g.solve_async()
    st, n_slow = g.solve_wait()
    want, used, ost = oracle.tick(cfg["cur"], cfg["load"], cfg["aff"], cfg["cap"], cfg["alive"])
    assert n_slow == 0 and st == ost and np.array_equal(g.get_solved(), want)
    # and a contended one through the async path
    cap = cfg["cap"] // np.uint64(4)
    g.set_nodes(cap, cfg["alive"])
    g.solve_async()
    st, n_slow = g.solve_wait()
    want, used, ost = oracle.tick(cfg["cur"], cfg["load"], cfg["aff"], cap, cfg["alive"])
    assert n_slow == 1 and st == ost and np.array_equal(g.get_solved(), want)
    g.close()


@pytest.mark.parametrize("spec", [None, "auto", "never", "always"])
def test_churn_stream_adaptive_packed_fixup(gp, oracle, spec):
    """Config-5 shape: committed ticks while a different 10 % of the nodes is down each tick.  From the second tick
    on the adaptive rule switches to the packed fix-up (few rows pending: the cuts are searched inside k_resolve) and
    enqueues it speculatively, without reading the verdict first; every tick must equal the oracle chain.
    spec None = the product library left to itself."""
    cfg = synth.config("c3", n_override=1_500_000)
    n, m = cfg["n"], cfg["m"]
    g = _mk(gp, n, m, cfg["load"], cfg["aff"], cfg["cap"], cfg["alive"], synth.warm_assign(n, m), lab=spec is not None)
    if spec is not None:
        g.set_speculate(spec)
    ref = synth.warm_assign(n, m)
    for tick in range(6):
        alive = synth.churn_mask(m, 2 + tick)
        g.set_alive_all(alive)
        st = g.tick()
        ref, used, ost = oracle.tick(ref, cfg["load"], cfg["aff"], cfg["cap"], alive, 2)
        assert st == ost, tick
        assert np.array_equal(g.get_assign(), ref), tick
        assert np.array_equal(g.get_nodes()[2], used), tick
        assert tick == 0 or (ost["slow_path"] == 1 and ost["evicted"] > 0)
    g.close()


INC_VARIANTS = (("auto", "auto"), ("auto", "always"), ("auto", "never"), ("never", "auto"))


@pytest.mark.parametrize("seed,n,m", [(0, 1, 1), (1, 255, 3), (2, 4097, 64), (3, 70_001, 33), (4, 300_000, 1024),
                                      (5, 1_000_003, 256), (6, 777_777, 4096), (7, 65_536 * 5, 8), (8, 2_000_000, 1000)])
def test_committed_ticks_in_place_scan_every_variant(gp, oracle, seed, n, m):
    """The in-place scan of committed ticks (k_inc_scan: only the assignment column is streamed, the kept load comes from the
    committed `used` vector, the decisions go into the committed column itself, the pending rows are dealt out evenly to the
    fix-up's workgroups by k_rebal) with the fix-up enqueued speculatively or after the verdict, and switched off
    (k_scan<COMPACT>): ten ticks of churn over random tables with unplaced rows,
    rows that are not objects, zero capacities and loads, table sizes around every tile / block boundary.  Every tick's
    table, `used` and counters against the oracle chain."""
    rng = np.random.default_rng(8800 + seed)
    cur, load, aff, cap, alive = _rand_case(rng, n, m, p_none=0.1, cap_scale=1.3, p_alive=0.9, max_load=300 if seed % 2 else 3)
    aff[rng.random(n) < 0.05] = 0xFFFFFFFE   # rows that are not objects
    masks = [(rng.random(m) < 0.88).astype(np.uint8) for _ in range(10)]
    masks[4] = np.ones(m, np.uint8)           # a tick in which everybody is alive
    masks[7] = np.zeros(m, np.uint8)          # ... and one in which nobody is
    for inc, spec in INC_VARIANTS:
        g = _mk(gp, n, m, load, aff, cap, alive, cur, 2, lab=True)
        g.set_compact("always", inc=inc)      # (the adaptive rule would keep small tables on the whole-table fix-up)
        g.set_speculate(spec)
        ref = cur.copy()
        for t, mask in enumerate(masks):
            g.set_alive_all(mask)
            ref, used, ost = oracle.tick(ref, load, aff, cap, mask, 2)
            st = g.tick() if t % 3 else None
            if st is None:                     # every third tick through the asynchronous entry point
                g.tick_async()
                st = g.tick_wait()[-1]
            assert st == ost, (inc, spec, t, st, ost)
            got = g.get_assign()
            assert np.array_equal(got, ref), (inc, spec, t, np.flatnonzero(got != ref)[:10])
            assert np.array_equal(g.get_nodes()[2], used), (inc, spec, t)
        # an un-committed solve right behind the in-place ticks: it must not touch the committed column
        ref2, used2, ost2 = oracle.tick(ref, load, aff, cap, masks[0], 2)
        g.set_alive_all(masks[0])
        assert g.solve() == ost2
        assert np.array_equal(g.get_solved(), ref2) and np.array_equal(g.get_assign(), ref)
        g.commit()
        assert np.array_equal(g.get_assign(), ref2) and np.array_equal(g.get_nodes()[2], used2)
        g.close()


def test_async_ticks_equal_the_synchronous_stream(gp, oracle):
    """rio_gp_tick_async: committed ticks enqueued back to back with liveness pushes in between, nothing waits on the host;
    tables and counters must be exactly those of the same sequence of rio_gp_tick calls (= the oracle chain), including
    more ticks in flight than the verdict ring holds (64) and synchronous calls in the middle of the stream."""
    cfg = synth.config("c3", n_override=600_000)
    n, m = cfg["n"], cfg["m"]
    ref = synth.warm_assign(n, m)
    g = _mk(gp, n, m, cfg["load"], cfg["aff"], cfg["cap"], cfg["alive"], ref)
    want = []
    ticks = 70
    for tick in range(ticks):
        alive = synth.churn_mask(m, 2 + tick) if tick % 5 else np.ones(m, np.uint8)   # every fifth tick: nothing to fix
        g.set_alive_all(alive)
        g.tick_async()
        ref, used, ost = oracle.tick(ref, cfg["load"], cfg["aff"], cfg["cap"], alive, 2)
        want.append(ost)
        if tick == 30:   # a synchronous lookup in the middle of the stream: ordered behind the ticks enqueued so far
            q = np.arange(0, n, 977, dtype=np.uint32)
            assert np.array_equal(g.lookup_batch(q), ref[q])
    got = g.tick_wait()
    assert len(got) == ticks
    for k in range(ticks):
        assert got[k] == want[k], (k, got[k], want[k])
    assert np.array_equal(g.get_assign(), ref)
    assert np.array_equal(g.get_nodes()[2], used)
    assert g.tick_wait() == []
    st = g.tick()                                            # and the synchronous call still works afterwards
    ref, used, ost = oracle.tick(ref, cfg["load"], cfg["aff"], cfg["cap"], alive, 2)
    assert st == ost and np.array_equal(g.get_assign(), ref)
    g.close()


@pytest.mark.parametrize("overlap", [True, False, "chained", "chained-events", "overlapped"])
def test_async_ticks_quiet_stream_and_every_kind_of_change(gp, oracle, overlap, monkeypatch):
    """A tick that took the fast path leaves every object placed: until an input of the solve changes, rio_gp_tick_async
    enqueues k_scan + k_resolve only (no speculative fix-up) — and k_resolve on a stream of its own beside the next tick's scan
    (overlap: the product library as it is; False: the lab build with that switched off; "chained" / "overlapped": the lab build with
    the size threshold of the overlap taken away and the chain in either of its forms — k_resolve in line behind its scan, what
    tables below 5 M rows get (also what the product does with this table: overlap=True), or on the side stream behind the scan's
    stop event, what a 10 M-row table gets — and the same without the chain).  Every kind of change must end that: liveness, removals,
    updates onto other nodes, new loads / affinities, a new table, clean_server, requests — the tick right behind each is
    compared with the oracle chain, with the verdicts of the earlier ticks given time to land (so the quiet rule is
    actually in force when the change arrives)."""
    import time
    cfg = synth.config("c3", n_override=300_000)
    n, m = cfg["n"], cfg["m"]
    load, aff, cap = cfg["load"].copy(), cfg["aff"].copy(), cfg["cap"]
    ref = synth.warm_assign(n, m)
    alive = np.ones(m, np.uint8)
    if isinstance(overlap, str):
        monkeypatch.setenv("RIO_GP_OVERLAP_MIN_ROWS", "1")
    if overlap == "chained-events":   # the form big tables use (k_resolve on the side stream behind the scan's stop event), on this small one
        monkeypatch.setenv("RIO_GP_CHAIN_INLINE_BELOW", "0")
    g = _mk(gp, n, m, load, aff, cap, alive, ref, lab=overlap is not True)
    if not overlap:
        g.set_compact("auto", overlap=False)
    elif overlap == "overlapped":
        g.set_compact("auto", chain=False)
    rng = np.random.default_rng(99)
    want = []

    def quiet_ticks(k=4):
        nonlocal ref
        for _ in range(k):
            g.tick_async()
            ref, used, ost = oracle.tick(ref, load, aff, cap, alive, 2)
            want.append(ost)
            time.sleep(0.003)      # the verdict lands; the next rio_gp_tick_async sees it

    def changes():
        nonlocal ref, alive
        alive = synth.churn_mask(m, 7); g.set_alive_all(alive); yield "liveness: a tenth of the nodes dies"
        alive = np.ones(m, np.uint8); g.set_alive_all(alive); yield "liveness: they come back"
        idx = rng.choice(n, 20_000, replace=False).astype(np.uint32)
        g.remove_batch(idx); ref[idx] = NONE; yield "remove"
        idx = rng.choice(n, 30_000, replace=False).astype(np.uint32)
        node = np.full(30_000, 5, np.uint32)                     # thirty thousand rows onto one node: far over its capacity
        g.update_batch(idx, node); ref[idx] = 5; yield "update (kept rows stay: sticky)"
        alive = np.ones(m, np.uint8); alive[5] = 0; g.set_alive(5, 0); yield "that node dies: its rows spill"
        alive[5] = 1; g.set_alive(5, 1); yield "and comes back"
        idx = rng.choice(n, 10_000, replace=False).astype(np.uint32)
        load[idx] = rng.integers(0, 500, 10_000).astype(np.uint32); aff[idx] = rng.integers(0, m, 10_000).astype(np.uint32)
        g.set_object_attrs(idx, load[idx], aff[idx]); yield "new loads and affinities"
        ref = synth.warm_assign(n, m, stream=3); ref[::7] = NONE; g.set_assign(ref); yield "a new table with pending rows"
        ev = g.clean_server(11); ref[ref == 11] = NONE; assert ev > 0; yield "clean_server"
        idx = np.flatnonzero(ref == NONE)[:5000].astype(np.uint32)
        used = oracle.recompute_used(ref, load, m)
        req = rng.integers(0, m, idx.size).astype(np.uint32)
        node, flag = g.place_pending(idx, req)
        wnode, wflag = oracle.place_pending(ref, load, cap, alive, used, idx, req)
        assert np.array_equal(node, wnode) and np.array_equal(flag, wflag); yield "place_pending"

    quiet_ticks()
    for what in changes():
        g.tick_async()
        ref, used, ost = oracle.tick(ref, load, aff, cap, alive, 2)
        want.append(ost)
        assert np.array_equal(g.get_assign(), ref), what
        assert np.array_equal(g.get_nodes()[2], used), what
        quiet_ticks()
    got = g.tick_wait()
    assert len(got) == len(want)
    for k in range(len(want)):
        assert got[k] == want[k], (k, got[k], want[k])
    assert np.array_equal(g.get_assign(), ref) and np.array_equal(g.get_nodes()[2], used)
    if isinstance(overlap, str):   # the quiet ticks of this run were links of a chain — or none of them was
        assert (g.chained_scans() > 20) == overlap.startswith("chained"), g.chained_scans()
    g.close()


@pytest.mark.parametrize("lab", [False, True])
def test_chained_quiet_ticks_hand_over_what_the_tick_before_them_wrote(gp, oracle, lab, monkeypatch):
    """A table big enough for the product's own rule (2^22 rows and more): once a tick's verdict says "fast path, nothing
    changed since", the scans of the following ticks alternate between two streams and hand their rows over workgroup by
    workgroup.  What a broken hand-over would show: the SECOND link of a run reads the column the first link wrote while that
    launch is still running — a stale read there sees the table as it was before the run (rows on dead nodes, rows not yet
    placed) and counts evictions / claims where the oracle chain counts kept rows.  Every tick's counters, the table and
    `used` after every run; cold start, then two rounds of node deaths with runs of back-to-back quiet ticks behind them."""
    import time
    cfg = synth.config("c3", n_override=(1 << 22) + 12_345)
    n, m = cfg["n"], cfg["m"]
    load, aff, cap = cfg["load"], cfg["aff"], cfg["cap"]
    alive = np.ones(m, np.uint8)
    ref = np.full(n, NONE, np.uint32)
    if lab:   # the form tables of 5 M rows and more get (k_resolve on the side stream); the product library runs this one in line
        monkeypatch.setenv("RIO_GP_CHAIN_INLINE_BELOW", "0")
    g = _mk(gp, n, m, load, aff, cap, alive, ref, lab=lab)
    want = []

    def tick():
        nonlocal ref
        g.tick_async()
        ref, used, ost = oracle.tick(ref, load, aff, cap, alive, 2)
        want.append(ost)
        return used

    for rnd in range(3):
        if rnd:
            alive = synth.churn_mask(m, 40 + rnd)
            g.set_alive_all(alive)
        for _ in range(3):          # the tick that changes the table, and two more until a verdict has certainly landed
            used = tick()
            time.sleep(0.01)
        for _ in range(70):         # a run longer than the ring of ticks: harvested in the middle, started again
            used = tick()
        got = g.tick_wait()
        assert len(got) == len(want)
        for k in range(len(want)):
            assert got[k] == want[k], (rnd, k, got[k], want[k])
        want = []
        assert np.array_equal(g.get_assign(), ref), rnd
        assert np.array_equal(g.get_nodes()[2], used), rnd
    if lab:
        assert g.chained_scans() >= 3 * 60, g.chained_scans()
    g.close()


def test_a_chained_wait_that_never_ends_gives_up_and_fails_loudly(gp, oracle, monkeypatch):
    """The in-kernel wait of a chained scan is bounded: if the workgroup it waits for never raises its flag (here: the lab build
    makes every link wait for a sequence number nobody stores), every wave gives up after a second, a word in mapped host memory
    is raised and rio_gp_tick_wait fails with Upstream — no hang, no silent result — and the handle works again once the table
    has been loaded anew."""
    monkeypatch.setenv("RIO_GP_CHAIN_DIAG", "3")
    monkeypatch.setenv("RIO_GP_OVERLAP_MIN_ROWS", "1")
    cfg = synth.config("c3", n_override=300_000)
    n, m = cfg["n"], cfg["m"]
    alive = np.ones(m, np.uint8)
    ref = synth.warm_assign(n, m)
    g = _mk(gp, n, m, cfg["load"], cfg["aff"], cfg["cap"], alive, ref, lab=True)
    import time
    g.tick_async()
    g.tick_wait()                       # (nothing chained yet: a verdict has to land first)
    for _ in range(3):
        g.tick_async()                  # a run: its second link waits for what never comes
    t0 = time.time()
    with pytest.raises(gp.ObjectPlacementError) as e:
        g.tick_wait()
    assert e.value.kind == "Upstream" and "chained scan gave up" in str(e.value), str(e.value)
    assert time.time() - t0 < 20
    monkeypatch.delenv("RIO_GP_CHAIN_DIAG")
    g.close()
    g = _mk(gp, n, m, cfg["load"], cfg["aff"], cfg["cap"], alive, ref, lab=True)   # (a fresh handle: the process is fine)
    st = g.tick()
    want, used, ost = oracle.tick(ref, cfg["load"], cfg["aff"], cfg["cap"], alive, 2)
    assert st == ost and np.array_equal(g.get_assign(), want)
    g.close()


def test_two_handles_tick_side_by_side_one_chain_at_a_time(gp, oracle):
    """Two handles of one process, a thread each, both running quiet tick streams over tables big enough to chain: one handle
    per process chains at a time (the chain's progress argument counts the workgroup slots of ONE pair of launches), the
    other's quiet ticks overlap without the chain, and the turn changes hands whenever a run ends.  Both tables, `used` and
    every tick's counters against the oracle."""
    import threading
    import time
    cfgs = [synth.config("c3", n_override=(1 << 22) + 1000 * (k + 1)) for k in range(2)]
    out, errs = [None, None], []

    def run(k):
        try:
            cfg = cfgs[k]
            n, m = cfg["n"], cfg["m"]
            alive = np.ones(m, np.uint8) if k == 0 else synth.churn_mask(m, 77)
            ref = np.full(n, NONE, np.uint32)
            g = _mk(gp, n, m, cfg["load"], cfg["aff"], cfg["cap"], alive, ref, lab=True)
            want, got = [], []
            for rnd in range(4):
                for i in range(40):
                    g.tick_async()
                    ref, used, ost = oracle.tick(ref, cfg["load"], cfg["aff"], cfg["cap"], alive, 2)
                    want.append(ost)
                    if i < 3:
                        time.sleep(0.005)
                got += g.tick_wait()
            out[k] = (got == want, np.array_equal(g.get_assign(), ref), np.array_equal(g.get_nodes()[2], used), g.chained_scans())
            g.close()
        except Exception as e:   # (a thread's exception must fail the test, not vanish)
            errs.append(repr(e))

    ts = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for k in range(2):
        assert out[k][:3] == (True, True, True), (k, out[k])
    assert out[0][3] + out[1][3] > 0, out   # (somebody chained)


# ---- place_pending ---------------------------------------------------------------------------------

@pytest.mark.parametrize("seed,cap_inf", [(0, True), (1, False), (2, False), (3, True)])
def test_place_pending_random(gp, oracle, seed, cap_inf):
    rng = np.random.default_rng(3000 + seed)
    n, m = int(rng.integers(200, 50000)), int(rng.integers(2, 200))
    load = rng.integers(0, 30, n).astype(np.uint32)
    cap = np.full(m, INF, np.uint64) if cap_inf else rng.integers(0, int(load.sum() / m) + 5, m).astype(np.uint64)
    alive = np.ones(m, np.uint8)
    g = gp.GpuPlacement(n, m, spill_rounds=2)
    g.set_nodes(cap, alive)
    g.set_objects(n, load, None)
    ref = np.full(n, NONE, np.uint32)
    used = np.zeros(m, np.uint64)
    for step in range(10):
        if step % 3 == 1:
            j = int(rng.integers(m))
            alive[j] ^= 1
            g.set_alive(j, alive[j])
        k = int(rng.integers(1, 20000))
        idx = rng.integers(0, n, k).astype(np.uint32)
        req = rng.integers(0, m, k).astype(np.uint32)
        node, flag = g.place_pending(idx, req)
        wnode, wflag = oracle.place_pending(ref, load, cap, alive, used, idx, req)
        assert np.array_equal(node, wnode) and np.array_equal(flag, wflag), step
        assert np.array_equal(g.get_assign(), ref)
        assert np.array_equal(g.get_nodes()[2], used)
    g.close()


def test_place_pending_replaced_flag(gp, oracle):
    """RIO_GP_FLAG_REPLACED (service.rs:268-285): the known-answer case of the oracle test, and a
    large batch in which a tenth of the nodes has just died."""
    assign = np.array([0, 0, 1, NONE], np.uint32)
    g = gp.GpuPlacement(4, 3)
    g.set_nodes(np.full(3, INF, np.uint64), np.array([0, 1, 1], np.uint8))
    g.set_objects(4, np.ones(4, np.uint32), None)
    g.set_assign(assign)
    node, flag = g.place_pending(np.array([0, 0, 2, 1, 3], np.uint32), np.array([1, 2, 1, 2, 2], np.uint32))
    assert list(node) == [1, 1, 1, 2, 2]
    assert list(flag) == [gp.FLAG_PLACED | gp.FLAG_REPLACED, gp.FLAG_REDIRECT, gp.FLAG_LOCAL,
                          gp.FLAG_PLACED | gp.FLAG_REPLACED, gp.FLAG_PLACED]
    assert list(g.get_assign()) == [1, 2, 1, 2]
    g.close()
    rng = np.random.default_rng(8)
    n, m, k = 200_000, 300, 50_000
    load = rng.integers(1, 20, n).astype(np.uint32)
    ref = rng.integers(0, m, n).astype(np.uint32)
    ref[rng.random(n) < 0.2] = NONE
    alive = (rng.random(m) > 0.1).astype(np.uint8)
    cap = np.full(m, int(load.sum()), np.uint64)
    g = gp.GpuPlacement(n, m)
    g.set_nodes(cap, alive)
    g.set_objects(n, load, None)
    g.set_assign(ref)
    used = oracle.recompute_used(ref, load, m)
    idx = rng.integers(0, n, k).astype(np.uint32)
    req = np.flatnonzero(alive)[rng.integers(0, int(alive.sum()), k)].astype(np.uint32)
    start = ref.copy()
    node, flag = g.place_pending(idx, req)
    wnode, wflag = oracle.place_pending(ref, load, cap, alive, used, idx, req)
    assert np.array_equal(node, wnode) and np.array_equal(flag, wflag)
    rep = (flag & gp.FLAG_REPLACED) != 0
    assert rep.sum() > 1000
    on_dead = (start[idx] != NONE) & (alive[np.minimum(start[idx], m - 1)] == 0)
    assert not (rep & ~on_dead).any()                      # only requests that found their object on a dead node ...
    _, firsts = np.unique(idx, return_index=True)
    is_first = np.zeros(k, bool); is_first[firsts] = True
    assert np.array_equal(rep, on_dead & is_first)         # ... and only the first request of each such object
    assert np.array_equal(g.get_assign(), ref)
    g.close()


def test_place_pending_batch_sizes_around_every_path_boundary(gp, oracle):
    """place_pending: one workgroup up to 1 024 requests, three launches (requests and rows gathered by many workgroups, the
    decision in one, results and table stores by many) up to 4 096 — or the general path over mapped pinned memory when
    the batch needs it —, staging copies beyond — the sizes on both sides of every boundary, with
    nodes dying in between, every call against the sequential oracle (nodes, flags, table, `used`)."""
    rng = np.random.default_rng(21)
    n, m = 300_000, 50
    load = rng.integers(1, 30, n).astype(np.uint32)
    cap = np.full(m, int(load.sum()) // m + 5000, np.uint64)
    alive = np.ones(m, np.uint8)
    g = gp.GpuPlacement(n, m, spill_rounds=2)
    g.set_nodes(cap, alive)
    g.set_objects(n, load, None)
    ref = np.full(n, NONE, np.uint32)
    used = np.zeros(m, np.uint64)
    for step, k in enumerate((256, 257, 1000, 1024, 1025, 4095, 4096, 4097, 20_000, 300, 2048, 257, 65_535, 65_536, 65_537, 8192)):
        if step % 3 == 2:
            j = int(rng.integers(m))
            alive[j] ^= 1
            g.set_alive(j, alive[j])
        idx = rng.integers(0, n // 4, k).astype(np.uint32)
        req = np.flatnonzero(alive)[rng.integers(0, int(alive.sum()), k)].astype(np.uint32)
        node, flag = g.place_pending(idx, req)
        wnode, wflag = oracle.place_pending(ref, load, cap, alive, used, idx, req)
        assert np.array_equal(node, wnode) and np.array_equal(flag, wflag), (step, k)
        assert np.array_equal(g.get_assign(), ref), (step, k)
        assert np.array_equal(g.get_nodes()[2], used), (step, k)
    g.close()


@pytest.mark.parametrize("seed,cap_mode", [(0, "inf"), (1, "tight"), (2, "roomy"), (3, "tight")])
def test_place_pending_micro_batches(gp, oracle, seed, cap_mode):
    """Batches of 1..256 requests take the one-launch micro-batch kernel (k_pp_small) when nothing heavy is
    involved and the general path otherwise; the request stream below mixes both (duplicates inside a batch, dead
    current nodes, dead and full requesters, zero-load objects) and every call must equal the sequential oracle."""
    rng = np.random.default_rng(4000 + seed)
    n, m = 3000, int(rng.integers(2, 40))
    load = rng.integers(0, 9, n).astype(np.uint32)
    cap = {"inf": np.full(m, INF, np.uint64),
           "tight": rng.integers(0, 40, m).astype(np.uint64),
           "roomy": np.full(m, int(load.sum()), np.uint64)}[cap_mode]
    alive = np.ones(m, np.uint8)
    g = gp.GpuPlacement(n, m, spill_rounds=2)
    g.set_nodes(cap, alive)
    g.set_objects(n, load, None)
    ref = np.full(n, NONE, np.uint32)
    used = np.zeros(m, np.uint64)
    for step in range(120):
        if step % 17 == 5:
            j = int(rng.integers(m))
            alive[j] ^= 1
            g.set_alive(j, alive[j])
        if step % 29 == 7:   # raw CRUD in between: `used` has to be rebuilt before the next micro-batch
            ii = rng.integers(0, n, 20).astype(np.uint32)
            g.remove_batch(ii)
            oracle.remove_batch(ref, ii)
            used[:] = oracle.recompute_used(ref, load, m)
        k = int(rng.choice([1, 1, 2, 3, 4, 5, 7, 32, 64, 200, 256]))
        idx = rng.integers(0, min(n, 40 + 30 * step), k).astype(np.uint32)   # small id range: plenty of duplicates
        req = rng.integers(0, m, k).astype(np.uint32)
        node, flag = g.place_pending(idx, req)
        wnode, wflag = oracle.place_pending(ref, load, cap, alive, used, idx, req)
        assert np.array_equal(node, wnode) and np.array_equal(flag, wflag), (step, k)
        assert np.array_equal(g.lookup_batch(idx), ref[idx])
        if step % 10 == 0:
            assert np.array_equal(g.get_assign(), ref)
            assert np.array_equal(g.get_nodes()[2], used)
    assert np.array_equal(g.get_assign(), ref)
    assert np.array_equal(g.get_nodes()[2], used)
    g.close()


def test_place_pending_micro_batch_same_requester_capacity(gp, oracle):
    """Round-2 advisor finding: first-touch requests of ONE wave of the micro-batch kernel that share a requester must be
    admitted by the index-ordered prefix of their loads — a capacity that fits each row alone but not their sum.  This is
    the normal shape of the rio_op combiner (every caller passes the same self_address)."""
    g = gp.GpuPlacement(64, 3)
    cap = np.array([2, 100, 100], np.uint64)
    g.set_nodes(cap, np.ones(3, np.uint8))
    load = np.ones(64, np.uint32)
    g.set_objects(64, load, None)
    ref, used = np.full(64, NONE, np.uint32), np.zeros(3, np.uint64)
    idx, req = np.array([0, 1, 2], np.uint32), np.zeros(3, np.uint32)
    node, flag = g.place_pending(idx, req)
    wnode, wflag = oracle.place_pending(ref, load, cap, np.ones(3, np.uint8), used, idx, req)
    assert np.array_equal(node, wnode) and np.array_equal(flag, wflag), (node, wnode, flag, wflag)
    assert list(g.get_nodes()[2]) == list(used) and used[0] == 2
    g.close()
    rng = np.random.default_rng(91)
    for trial in range(30):
        m = int(rng.integers(1, 6))
        n = 2000
        load = rng.integers(0, 5, n).astype(np.uint32)
        k = int(rng.choice([3, 17, 64, 65, 130, 256]))
        alive = np.ones(m, np.uint8)
        g = gp.GpuPlacement(n, m)
        ref, used = np.full(n, NONE, np.uint32), np.zeros(m, np.uint64)
        # capacities around the total a batch asks of its requesters: some batches fit (micro path), some overflow by a row
        cap = rng.integers(max(1, k // m - 4), 2 * k // m + 6, m).astype(np.uint64)
        g.set_nodes(cap, alive)
        g.set_objects(n, load, None)
        for step in range(3):
            idx = rng.permutation(n)[:k].astype(np.uint32)   # distinct first-touch rows
            req = rng.integers(0, m, k).astype(np.uint32)
            node, flag = g.place_pending(idx, req)
            wnode, wflag = oracle.place_pending(ref, load, cap, alive, used, idx, req)
            assert np.array_equal(node, wnode) and np.array_equal(flag, wflag), (trial, step, k, m)
            assert np.array_equal(g.get_nodes()[2], used), (trial, step)
        assert np.array_equal(g.get_assign(), ref)
        g.close()


@pytest.mark.parametrize("seed,cap_mode", [(0, "inf"), (1, "tight"), (2, "roomy")])
def test_place_pending_big_batches_partitioned_by_row_window(gp, oracle, seed, cap_mode):
    """Batches of >= 2^18 requests are sorted by row window once, the row-side step works out of LDS (k_pp_win_gather) and the
    solve writes its decisions into the table itself (k_scan<COMPACT 3>, k_fill through the object column): duplicates inside a batch (the first request decides), objects on dead nodes (clean_server of the whole
    node, REPLACED), dead and full requesters (spill / unplaced), the row-lifecycle column — nodes, flags, table and `used`
    against the sequential oracle, from host buffers and from device buffers."""
    from hipbuf import DevBuf
    rng = np.random.default_rng(5200 + seed)
    n, m = 1_400_000, 300
    load = rng.integers(0, 30, n).astype(np.uint32)
    cap = {"inf": np.full(m, INF, np.uint64), "tight": rng.integers(0, int(load.sum() / m) + 5, m).astype(np.uint64),
           "roomy": np.full(m, int(load.sum()), np.uint64)}[cap_mode]
    alive = np.ones(m, np.uint8)
    g = gp.GpuPlacement(n, m, flags=gp.CFG_ROW_LIFECYCLE)
    plain = gp.LabPlacement(n, m, flags=gp.CFG_ROW_LIFECYCLE)          # the same calls through the plain per-request kernels
    plain.set_compact("auto", partitioned_crud=False)
    small = gp.LabPlacement(n, m, flags=gp.CFG_ROW_LIFECYCLE)          # ... and through 16 384-entry chunks (the form of >= 4 M requests)
    gp.lab_lib().rio_gp_debug_set_part_shift(14 | 0x80)
    for h in (g, plain, small):
        h.set_nodes(cap, alive)
        h.set_objects(n, load, None)
    ref = np.full(n, NONE, np.uint32)
    used = np.zeros(m, np.uint64)
    # (2.2 M requests: more than four steps of chunks — the window kernel's long form with 8 192-entry chunks; the others its short one)
    for step, k in enumerate((300_000, 1_000_000, 262_144, 700_001, 2_200_000)):
        if step >= 1:
            for j in rng.integers(0, m, 12):
                alive[j] ^= 1
            for h in (g, plain, small):
                h.set_alive_all(alive)
        idx = rng.integers(0, n if step % 2 else n // 3, k).astype(np.uint32)   # (a third of the table: heavy duplication)
        req = rng.integers(0, m, k).astype(np.uint32)
        if step % 2 == 0:
            node, flag = g.place_pending(idx, req)
        else:
            d_idx, d_req, d_node, d_flag = DevBuf(idx), DevBuf(req), DevBuf(nbytes=4 * k), DevBuf(nbytes=4 * k)
            g.place_pending_dev(k, d_idx.ptr, d_req.ptr, d_node.ptr, d_flag.ptr)
            node, flag = d_node.to_host(), d_flag.to_host()
            for x in (d_idx, d_req, d_node, d_flag):
                x.free()
        pnode, pflag = plain.place_pending(idx, req)
        snode, sflag = small.place_pending(idx, req)
        wnode, wflag = oracle.place_pending(ref, load, cap, alive, used, idx, req)
        assert np.array_equal(node, wnode), (step, np.flatnonzero(node != wnode)[:5])
        assert np.array_equal(flag, wflag), (step, np.flatnonzero(flag != wflag)[:5])
        assert np.array_equal(pnode, wnode) and np.array_equal(pflag, wflag), step
        assert np.array_equal(snode, wnode) and np.array_equal(sflag, wflag), step
        assert np.array_equal(g.get_assign(), ref) and np.array_equal(small.get_assign(), ref), step
        assert np.array_equal(g.get_nodes()[2], used) and np.array_equal(small.get_nodes()[2], used), step
        assert np.array_equal(g.get_objects()[1], plain.get_objects()[1]), step   # row lifecycle column: same as the plain kernels
        assert np.array_equal(small.get_objects()[1], plain.get_objects()[1]), step
    gp.lab_lib().rio_gp_debug_set_part_shift(14)
    plain.close()
    small.close()
    # one index out of range: the call fails and NOTHING has changed (the kernels enqueued behind the validating one see its
    # counter and do nothing) — then the same handle goes on working
    life = g.get_objects()[1].copy()
    g.set_assign(np.full(n, NONE, np.uint32)); ref[:] = NONE; used[:] = 0
    bad = DevBuf(np.concatenate([np.arange(300_000, dtype=np.uint32), np.array([n], np.uint32)]))
    rq, out = DevBuf(np.zeros(300_001, np.uint32)), DevBuf(nbytes=4 * 300_001)
    with pytest.raises(gp.ObjectPlacementError) as e:
        g.place_pending_dev(300_001, bad.ptr, rq.ptr, out.ptr)
    assert e.value.rc == gp.EINVAL and np.array_equal(g.get_assign(), ref)
    assert np.array_equal(g.get_nodes()[2], used) and np.array_equal(g.get_objects()[1], life)
    idx = rng.integers(0, n, 300_000).astype(np.uint32)
    req = rng.integers(0, m, 300_000).astype(np.uint32)
    node, flag = g.place_pending(idx, req)
    wnode, wflag = oracle.place_pending(ref, load, cap, alive, used, idx, req)
    assert np.array_equal(node, wnode) and np.array_equal(flag, wflag) and np.array_equal(g.get_assign(), ref)
    g.close()


def test_place_pending_dev_equals_host_call(gp, oracle):
    """rio_gp_place_pending_dev: request / result arrays resident in HBM (torch tensors), same answers as the oracle;
    a bad entry fails the call before anything changes."""
    from hipbuf import DevBuf   # device arrays through hipMalloc / hipMemcpy (tools/hipbuf.py): no torch in the loop
    rng = np.random.default_rng(77)
    n, m = 400_000, 300
    load = rng.integers(0, 40, n).astype(np.uint32)
    cap = rng.integers(0, int(load.sum() / m) + 5, m).astype(np.uint64)
    alive = np.ones(m, np.uint8)
    alive[[7, 100]] = 0
    g = gp.GpuPlacement(n, m)
    g.set_nodes(cap, alive)
    g.set_objects(n, load, None)
    ref = np.full(n, NONE, np.uint32)
    used = np.zeros(m, np.uint64)
    for step in range(4):
        k = int(rng.integers(1000, 300_000))
        idx = rng.integers(0, n, k).astype(np.uint32)
        req = rng.integers(0, m, k).astype(np.uint32)
        d_idx, d_req = DevBuf(idx), DevBuf(req)
        d_node, d_flag = DevBuf(nbytes=4 * k), DevBuf(nbytes=4 * k)
        g.place_pending_dev(k, d_idx.ptr, d_req.ptr, d_node.ptr, d_flag.ptr)
        wnode, wflag = oracle.place_pending(ref, load, cap, alive, used, idx, req)
        assert np.array_equal(d_node.to_host(), wnode), step
        assert np.array_equal(d_flag.to_host(), wflag), step
        assert np.array_equal(g.get_assign(), ref)
        assert np.array_equal(g.get_nodes()[2], used)
        for x in (d_idx, d_req, d_node, d_flag):
            x.free()
    bad = DevBuf(np.array([1, 2, n, 3], np.uint32))      # index n is out of range
    rq, out = DevBuf(np.zeros(4, np.uint32)), DevBuf(nbytes=16)
    with pytest.raises(gp.ObjectPlacementError) as e:
        g.place_pending_dev(4, bad.ptr, rq.ptr, out.ptr)
    assert e.value.rc == gp.EINVAL
    assert np.array_equal(g.get_assign(), ref)                                       # nothing was changed
    g.close()


@pytest.mark.parametrize("speculate", ["auto", "always", "never"])
def test_place_pending_general_path_no_host_round_trip(gp, oracle, speculate):
    """The general request path (k_ppm_first / k_ppm_gather / solve / k_ppm_output: 4 097 ... 2^18 requests, or anything the
    one-workgroup kernels hand over) enqueues everything and waits once: an invalid entry anywhere in a device-resident batch
    fails the call with NOTHING changed (table, `used`, the election scratch: the next call is right), a solve that needs the
    cut / water-fill is found out on the device (status 1 -> fix-up -> outputs) or has its fix-up enqueued ahead of the verdict,
    dead nodes in the way are cleaned (REPLACED), from device arrays and from host buffers — every call against the oracle."""
    from hipbuf import DevBuf
    rng = np.random.default_rng(333)
    n, m = 500_000, 120
    load = rng.integers(0, 25, n).astype(np.uint32)
    cap = np.full(m, int(load.sum() // m // 4), np.uint64)     # tight: batches run requesters full along the way
    alive = np.ones(m, np.uint8)
    g = gp.LabPlacement(n, m, spill_rounds=2)
    g.set_speculate(speculate)
    g.set_nodes(cap, alive)
    g.set_objects(n, load, None)
    ref = np.full(n, NONE, np.uint32)
    used = np.zeros(m, np.uint64)
    for step, k in enumerate((4097, 8192, 5000, 65_536, 16_384, 70_001, 131_072, 6001, 4100)):
        if step in (3, 6):
            for j in rng.integers(0, m, 6):
                alive[j] ^= 1
            g.set_alive_all(alive)
        idx = rng.integers(0, n if step % 2 else n // 8, k).astype(np.uint32)
        req = rng.integers(0, m, k).astype(np.uint32)                      # dead requesters included
        if step % 3 == 1:   # a batch with ONE invalid entry first: nothing may change
            bi, br = idx.copy(), req.copy()
            if step % 2:
                bi[int(rng.integers(k))] = n
            else:
                br[int(rng.integers(k))] = m
            d_idx, d_req, d_node = DevBuf(bi), DevBuf(br), DevBuf(nbytes=4 * k)
            with pytest.raises(gp.ObjectPlacementError) as e:
                g.place_pending_dev(k, d_idx.ptr, d_req.ptr, d_node.ptr)
            assert e.value.rc == gp.EINVAL and np.array_equal(g.get_assign(), ref) and np.array_equal(g.get_nodes()[2], used)
            for x in (d_idx, d_req, d_node):
                x.free()
        if step % 2 == 0:
            d_idx, d_req, d_node, d_flag = DevBuf(idx), DevBuf(req), DevBuf(nbytes=4 * k), DevBuf(nbytes=4 * k)
            g.place_pending_dev(k, d_idx.ptr, d_req.ptr, d_node.ptr, d_flag.ptr)
            node, flag = d_node.to_host(), d_flag.to_host()
            for x in (d_idx, d_req, d_node, d_flag):
                x.free()
        else:
            node, flag = g.place_pending(idx, req)
        wnode, wflag = oracle.place_pending(ref, load, cap, alive, used, idx, req)
        assert np.array_equal(node, wnode), (step, k, np.flatnonzero(node != wnode)[:5])
        assert np.array_equal(flag, wflag), (step, k, np.flatnonzero(flag != wflag)[:5])
        assert np.array_equal(g.get_assign(), ref), (step, k)
        assert np.array_equal(g.get_nodes()[2], used), (step, k)
    g.close()


def test_place_pending_dev_arrays_that_are_not_16_byte_aligned(gp, oracle):
    """A caller's device arrays may start anywhere (4-byte aligned): the one-workgroup kernel, the general path's dwordx4
    loops and the window-sorted form's whole-vector stores all step aside for arrays that are not 16-byte aligned — same
    answers, every batch-size regime."""
    from hipbuf import DevBuf
    rng = np.random.default_rng(4242)
    n, m = 400_000, 64
    load = rng.integers(0, 20, n).astype(np.uint32)
    cap = np.full(m, int(load.sum() // m // 2), np.uint64)
    alive = np.ones(m, np.uint8)
    alive[[3, 50]] = 0
    g = gp.GpuPlacement(n, m)
    g.set_nodes(cap, alive)
    g.set_objects(n, load, None)
    ref = np.full(n, NONE, np.uint32)
    used = np.zeros(m, np.uint64)
    for step, (k, off) in enumerate(((700, 1), (4096, 3), (5001, 1), (70_000, 2), (300_000, 1), (300_001, 3))):
        idx = rng.integers(0, n, k).astype(np.uint32)
        req = rng.integers(0, m, k).astype(np.uint32)
        pad = np.zeros(off, np.uint32)
        d_idx, d_req = DevBuf(np.concatenate([pad, idx])), DevBuf(np.concatenate([pad, req]))
        d_node, d_flag = DevBuf(nbytes=4 * (k + off)), DevBuf(nbytes=4 * (k + off))
        sh = 4 * off
        g.place_pending_dev(k, d_idx.ptr + sh, d_req.ptr + sh, d_node.ptr + sh, d_flag.ptr + sh)
        wnode, wflag = oracle.place_pending(ref, load, cap, alive, used, idx, req)
        assert np.array_equal(d_node.to_host()[off:], wnode), (step, k)
        assert np.array_equal(d_flag.to_host()[off:], wflag), (step, k)
        assert np.array_equal(g.get_assign(), ref) and np.array_equal(g.get_nodes()[2], used), (step, k)
        for x in (d_idx, d_req, d_node, d_flag):
            x.free()
    g.close()


def test_place_pending_dev_small_batches_one_launch(gp, oracle):
    """Device-resident batches of up to 4 096 requests go through the one-workgroup kernel first, reading the caller's
    arrays in place: sizes that are not multiples of four (no vector past the end of an exact-size array), duplicates, a
    flag array left out, entries out of range anywhere in the batch (status 3: EINVAL, nothing changed), requesters that are
    dead or full (hand-over to the general path, same answers)."""
    from hipbuf import DevBuf
    rng = np.random.default_rng(81)
    n, m = 150_000, 96
    load = rng.integers(0, 40, n).astype(np.uint32)
    cap = np.full(m, int(load.sum() // m // 3), np.uint64)           # tight: some requesters run full along the way
    alive = np.ones(m, np.uint8)
    g = gp.GpuPlacement(n, m)
    g.set_nodes(cap, alive)
    g.set_objects(n, load, None)
    ref = np.full(n, NONE, np.uint32)
    used = np.zeros(m, np.uint64)
    for step, k in enumerate((1, 3, 7, 255, 256, 257, 1001, 1024, 1025, 4095, 4096, 2, 4093)):
        if step == 8:
            alive[[5, 40]] = 0
            g.set_alive_all(alive)
        idx = rng.integers(0, n if step % 2 else 3000, k).astype(np.uint32)      # (3 000 rows: duplicates, sticky hits)
        req = rng.integers(0, m, k).astype(np.uint32)
        d_idx, d_req, d_node, d_flag = DevBuf(idx), DevBuf(req), DevBuf(nbytes=4 * k), DevBuf(nbytes=4 * k)
        g.place_pending_dev(k, d_idx.ptr, d_req.ptr, d_node.ptr, d_flag.ptr if step % 3 else None)
        wnode, wflag = oracle.place_pending(ref, load, cap, alive, used, idx, req)
        assert np.array_equal(d_node.to_host(), wnode), (step, k)
        if step % 3:
            assert np.array_equal(d_flag.to_host(), wflag), (step, k)
        assert np.array_equal(g.get_assign(), ref), (step, k)
        assert np.array_equal(g.get_nodes()[2], used), (step, k)
        for x in (d_idx, d_req, d_node, d_flag):
            x.free()
    for k, where, what in ((9, 8, "idx"), (4096, 0, "req"), (1000, 999, "idx"), (257, 100, "req")):
        idx = rng.integers(0, n, k).astype(np.uint32)
        req = rng.integers(0, m, k).astype(np.uint32)
        if what == "idx":
            idx[where] = n
        else:
            req[where] = m
        d_idx, d_req, d_node = DevBuf(idx), DevBuf(req), DevBuf(nbytes=4 * k)
        with pytest.raises(gp.ObjectPlacementError) as e:
            g.place_pending_dev(k, d_idx.ptr, d_req.ptr, d_node.ptr)
        assert e.value.rc == gp.EINVAL and np.array_equal(g.get_assign(), ref) and np.array_equal(g.get_nodes()[2], used)
    g.close()


def test_config1_ping_pong_plumbing(gp, oracle):
    """BASELINE config 1: 1 000 objects x 4 nodes: 1 000 misses -> first touch -> 1 000 hits ->
    clean_server(node 2) -> re-place, against the string-level reference policy."""
    cfg = synth.config("c1")
    n, m = cfg["n"], cfg["m"]
    g = _mk(gp, n, m, cfg["load"], cfg["aff"], cfg["cap"], cfg["alive"])
    provider, storage = oracle.LocalObjectPlacement(), oracle.LocalStorage()
    for j in range(m):
        ip, port = synth.node_address(j).split(":")
        storage.push(ip, port, True)
    idx = np.arange(n, dtype=np.uint32)
    assert np.all(g.lookup_batch(idx) == NONE)
    node, flag = g.place_pending(idx, cfg["aff"])
    assert np.all(flag == gp.FLAG_PLACED) and np.array_equal(node, cfg["aff"])
    for i in range(n):
        assert oracle.get_or_create_placement(provider, storage, synth.node_address(int(cfg["aff"][i])), "Room",
                                              str(i)) == synth.node_address(int(node[i]))
    node2, flag2 = g.place_pending(idx, np.full(n, 1, np.uint32))        # every request lands on server 1
    assert np.array_equal(node2, node) and np.all(flag2 == np.where(node == 1, gp.FLAG_LOCAL, gp.FLAG_REDIRECT))
    ev = g.clean_server(2)
    provider.clean_server(synth.node_address(2))
    assert ev == int((cfg["aff"] == 2).sum())
    g.set_alive(2, 0)
    storage.set_is_active(*synth.node_address(2).split(":"), False)
    node3, flag3 = g.place_pending(idx, np.full(n, 3, np.uint32))
    for i in range(n):
        assert oracle.get_or_create_placement(provider, storage, synth.node_address(3), "Room", str(i)) == \
            synth.node_address(int(node3[i]))
    assert np.all(node3[cfg["aff"] == 2] == 3)
    g.close()


# ---- full-size properties (BASELINE sizes; size-independent checks, no oracle run) -------------------

def test_full_size_config3_properties(gp):
    cfg = synth.config("c3")
    n, m = cfg["n"], cfg["m"]
    g = _mk(gp, n, m, cfg["load"], cfg["aff"], cfg["cap"], cfg["alive"])
    st = g.tick()
    a = g.get_assign()
    assert st["slow_path"] == 0 and st["claimed"] == n and np.array_equal(a, cfg["aff"])  # no node overflows
    used = g.get_nodes()[2]
    assert int(used.sum()) == int(cfg["load"].astype(np.uint64).sum())                   # checksum of checksums
    assert np.array_equal(used, np.bincount(a, weights=cfg["load"].astype(np.float64), minlength=m).astype(np.uint64))
    assert np.all(used <= cfg["cap"])
    # churn: 10 % of the nodes die -> evict + re-place, nothing lands on a dead node, capacity holds
    alive = synth.churn_mask(m, 1)
    g.set_alive_all(alive)
    st2 = g.tick()
    b = g.get_assign()
    moved = a != b
    assert st2["evicted"] == int((alive[a] == 0).sum()) == int(moved.sum())
    assert np.all(alive[b[b != NONE]] == 1)
    used2 = g.get_nodes()[2]
    assert np.all(used2[alive == 1] <= cfg["cap"][alive == 1]) and np.all(used2[alive == 0] == 0)
    assert st2["kept"] + st2["claimed"] + st2["spilled"] + st2["unplaced"] == n
    st3 = g.tick()                                                                         # idempotent
    assert st3["evicted"] == 0 and np.array_equal(g.get_assign()[b != NONE], b[b != NONE])
    # clean_server == evict-by-scan
    ev = g.clean_server(int(b[0]))
    assert ev == int((b == b[0]).sum())
    g.close()


def test_soak_churn_stream_against_the_oracle():
    """120 ticks of membership churn with load / affinity edits, removals and request micro-batches in between, every
    tick and every batch compared with the oracle (tools/soak_churn.py; profiles/archive/r03_soak.json holds 4 900 such ticks)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_churn.py"), "120", "200000", "512", "5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["all_ticks_equal_oracle"] is True



def test_mixed_batch_is_the_four_calls_in_one_round_trip(gp, oracle):
    """rio_gp_mixed_batch (what the combiner sends when the callers of one generation asked for different things): update,
    remove, lookup and place_pending over up to 256 entries each, in that order, one enqueue and one wait — against the
    oracle's four calls, for every subset of kinds, entries riding in the kernel arguments (<= 4) and staged ones, duplicates
    across the kinds, a dead node in the way (the place_pending part hands over to the general path), tight capacities."""
    rng = np.random.default_rng(1234)
    n, m = 50_000, 48
    load = rng.integers(0, 30, n).astype(np.uint32)
    cap = np.full(m, int(load.sum() // m // 4), np.uint64)
    alive = np.ones(m, np.uint8)
    g = gp.GpuPlacement(n, m)
    g.set_nodes(cap, alive)
    g.set_objects(n, load, None)
    ref = np.full(n, NONE, np.uint32)
    for step in range(60):
        if step == 30:
            alive[[5, 17]] = 0
            g.set_alive_all(alive)
        mask = step % 15 + 1                                  # every non-empty subset of the four kinds
        def some(bit):
            if not mask >> bit & 1:
                return None
            k = int(rng.choice((1, 3, 4, 5, 60, 255, 256)))
            return rng.integers(0, 2000 if step % 2 else n, k).astype(np.uint32)   # narrow range: the kinds hit the same rows
        ui, ri, li, pi = some(0), some(1), some(2), some(3)
        un = None if ui is None else rng.integers(0, m, ui.size).astype(np.uint32)
        if un is not None:
            un[rng.random(un.size) < 0.1] = NONE
        pr = None if pi is None else rng.integers(0, m, pi.size).astype(np.uint32)
        rc, lo, pn, pf = g.mixed_batch(update=None if ui is None else (ui, un), remove=ri, lookup=li,
                                       place=None if pi is None else (pi, pr))
        assert rc == [0, 0, 0, 0], (step, rc)
        if ui is not None:
            oracle.update_batch(ref, m, ui, un)
        if ri is not None:
            oracle.remove_batch(ref, ri)
        if li is not None:
            assert np.array_equal(lo, oracle.lookup_batch(ref, li)), step
        if pi is not None:
            used = oracle.recompute_used(ref, load, m)
            wnode, wflag = oracle.place_pending(ref, load, cap, alive, used, pi, pr)
            assert np.array_equal(pn, wnode) and np.array_equal(pf, wflag), step
        assert np.array_equal(g.get_assign(), ref), step
        assert np.array_equal(g.get_nodes()[2], oracle.recompute_used(ref, load, m)), step
    g.close()


def test_mixed_batch_refuses_one_kind_and_runs_the_others(gp, oracle):
    """A kind with an out-of-range entry changes nothing and says so in rc[] (what its own call would have returned); the
    other kinds of the same call run.  More than 256 entries of a kind: the whole call is RIO_GP_EINVAL, nothing runs."""
    n, m = 1000, 8
    g = gp.GpuPlacement(n, m)
    g.set_nodes(np.full(m, INF, np.uint64), np.ones(m, np.uint8))
    g.set_objects(n, np.ones(n, np.uint32), None)
    rc, lo, pn, pf = g.mixed_batch(update=(np.array([1, 2, n], np.uint32), np.array([0, 1, 2], np.uint32)),   # row n: out of range
                                   remove=np.array([7], np.uint32), lookup=np.array([1, 2, 3], np.uint32),
                                   place=(np.array([3, 4], np.uint32), np.array([5, 6], np.uint32)))
    assert rc == [gp.EINVAL, 0, 0, 0]
    assert list(lo) == [NONE, NONE, NONE]                     # the refused update wrote nothing
    assert list(pn) == [5, 6] and list(pf) == [gp.FLAG_PLACED, gp.FLAG_PLACED]
    rc, lo, pn, pf = g.mixed_batch(update=(np.array([1, 2], np.uint32), np.array([0, 1], np.uint32)),
                                   remove=np.array([3], np.uint32), lookup=np.array([1, 2, 3, 4, n + 5], np.uint32),
                                   place=(np.array([9], np.uint32), np.array([m], np.uint32)))             # requester m: out of range
    assert rc == [0, 0, gp.EINVAL, gp.EINVAL]
    a = g.get_assign()
    assert a[1] == 0 and a[2] == 1 and a[3] == NONE and a[4] == 6 and a[9] == NONE
    with pytest.raises(gp.ObjectPlacementError):
        g.mixed_batch(lookup=np.zeros(257, np.uint32))
    rc, lo, pn, pf = g.mixed_batch()
    assert rc == [0, 0, 0, 0]
    g.close()
