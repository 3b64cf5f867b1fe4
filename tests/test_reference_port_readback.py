"""The two CPU oracles against each other at a size the GPU tests reuse: the string-level restatement of the reference
(LocalObjectPlacement + Service::get_or_create_placement, oracle/local_placement_oracle.cpp; local.rs:22-68,
service.rs:193-254) run request by request and READ BACK, against the dense array oracle's tick (orc_tick) with every
capacity unbounded — warm tables, dead nodes (their objects are cleaned and first-touched on the requester), objects never
seen before.  bench.py's `parity.against_reference_port` and tests/test_gpu_parity.py rest on this equality."""
import numpy as np
import pytest

NONE = 0xFFFFFFFF
INF = 0xFFFFFFFFFFFFFFFF


@pytest.mark.parametrize("seed,n,m", [(0, 1, 1), (1, 500, 3), (2, 20_000, 64), (3, 60_000, 1024)])
def test_port_readback_equals_dense_tick_with_unbounded_capacity(oracle, seed, n, m):
    rng = np.random.default_rng(4100 + seed)
    alive = (rng.random(m) < 0.85).astype(np.uint8)
    alive[rng.integers(m)] = 1
    live = np.flatnonzero(alive)
    aff = live[rng.integers(0, len(live), n)].astype(np.uint32)      # requesters are active members (service.rs:244: self.address)
    cur = rng.integers(0, m, n).astype(np.uint32)                     # warm: somewhere, dead nodes included
    cur[rng.random(n) < 0.3] = NONE                                   # ... or never placed
    load = np.ones(n, np.uint32)
    want, used, st = oracle.tick(cur, load, aff, np.full(m, INF, np.uint64), alive, 2)
    secs, got = oracle.policy_readback(n, m, aff, alive, cur)
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
    assert st["spilled"] == 0 and st["unplaced"] == 0 and st["cut_nodes"] == 0
    assert secs >= 0.0


def test_port_readback_cold_table_is_first_touch(oracle):
    n, m = 5000, 16
    aff = (np.arange(n) * 7 % m).astype(np.uint32)
    _, got = oracle.policy_readback(n, m, aff)
    assert np.array_equal(got, aff)


@pytest.mark.parametrize("seed,n,m", [(0, 300, 4), (1, 20_000, 64), (2, 50_000, 500)])
def test_port_readback_with_inactive_requesters_equals_the_self_assign_tick(oracle, seed, n, m):
    """RIO_GP_CFG_REF_SELF_ASSIGN: requesters that membership marks inactive still take their first touches, like
    service.rs:244-252 — the dense oracle with the flag equals the string restatement served request by request.
    One thing a batch cannot reproduce is an artefact of the reference's request ORDER: an object first-touched onto an
    inactive server is wiped again when a LATER request of the same batch finds its own object on that server and cleans it
    (service.rs:227-237 -> local.rs:51-58 removes every entry of the address, the fresh one included).  So the inactive
    requesters here hold no objects of their own when the batch starts — the state every such server is in one batch after
    it went down; the other inactive servers do, and are cleaned."""
    rng = np.random.default_rng(4300 + seed)
    alive = (rng.random(m) < 0.7).astype(np.uint8)
    alive[0] = 0
    dead = np.flatnonzero(alive == 0)
    dead_req = dead[::2]                                              # inactive members that keep sending requests
    senders = np.concatenate([np.flatnonzero(alive), dead_req])      # requesters: the active members and those inactive ones
    aff = senders[rng.integers(0, len(senders), n)].astype(np.uint32)
    cur = rng.integers(0, m, n).astype(np.uint32)
    cur[np.isin(cur, dead_req)] = NONE                                # ... which hold nothing at the start of the batch
    cur[rng.random(n) < 0.3] = NONE
    load = np.ones(n, np.uint32)
    want, used, st = oracle.tick(cur, load, aff, np.full(m, INF, np.uint64), alive, 2, flags=oracle.REF_SELF_ASSIGN)
    _, got = oracle.policy_readback(n, m, aff, alive, cur)
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
    assert st["spilled"] == 0 and st["unplaced"] == 0
    dead_targets = int((alive[want[want != NONE]] == 0).sum())
    assert dead_targets > 0          # objects really were first-touched onto inactive members
    # without the flag the same table sends those first touches to the water-fill instead (the documented divergence)
    want0, _, st0 = oracle.tick(cur, load, aff, np.full(m, INF, np.uint64), alive, 2)
    assert int((alive[want0[want0 != NONE]] == 0).sum()) == 0 and st0["spilled"] > 0
