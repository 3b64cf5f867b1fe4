"""The dense-index solver oracle (oracle/placement_oracle.c): pinned to the reference policy
where the reference defines behaviour (capacity = infinity), hand-computed known answers and
invariants where it does not (capacity / spill: "parity unpinned")."""
import numpy as np
import pytest

import synth

NONE = 0xFFFFFFFF
INF = 0xFFFFFFFFFFFFFFFF


def _addr(j):
    return synth.node_address(j)


def _mk_storage(oracle, alive):
    st = oracle.LocalStorage()
    for j, a in enumerate(alive):
        ip, port = _addr(j).split(":")
        st.push(ip, port, bool(a))
    return st


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_place_pending_equals_reference_policy_when_capacity_is_infinite(oracle, seed):
    """cap = inf, load = 1: orc_place_pending == service.rs:193-254 run request by request."""
    rng = np.random.default_rng(seed)
    n_obj, m = 300, 7
    alive = np.ones(m, np.uint8)
    assign = np.full(n_obj, NONE, np.uint32)
    load = np.ones(n_obj, np.uint32)
    cap = np.full(m, INF, np.uint64)
    used = np.zeros(m, np.uint64)
    provider = oracle.LocalObjectPlacement()
    storage = _mk_storage(oracle, alive)
    for step in range(30):
        if step % 7 == 3:  # a node dies / revives between batches
            j = int(rng.integers(m))
            alive[j] ^= 1
            ip, port = _addr(j).split(":")
            storage.set_is_active(ip, port, bool(alive[j]))
        live = np.flatnonzero(alive)
        if len(live) == 0:
            continue
        nreq = int(rng.integers(1, 60))
        idx = rng.integers(0, n_obj, nreq).astype(np.uint32)           # duplicates on purpose
        req = live[rng.integers(0, len(live), nreq)].astype(np.uint32)  # requesters are live servers
        start = assign.copy()
        out_node, out_flag = oracle.place_pending(assign, load, cap, alive, used, idx, req)
        seen = set()
        for k in range(nreq):
            before = provider.lookup("Obj", str(idx[k]))
            got = oracle.get_or_create_placement(provider, storage, _addr(int(req[k])), "Obj", str(idx[k]))
            assert got == _addr(int(out_node[k])), (step, k)
            verdict = oracle.check_address_mismatch(provider, storage, _addr(int(req[k])), got)
            base = int(out_flag[k]) & 0x0F
            assert verdict == ("ok" if base in (0, 2) else "redirect")
            if base == 2:
                assert before is None or before != got
            # REPLACED (service.rs:268-285): the first request of an object that sat on a dead server when the batch arrived
            first = int(idx[k]) not in seen
            seen.add(int(idx[k]))
            was_dead = start[idx[k]] != NONE and not alive[start[idx[k]]]
            assert bool(int(out_flag[k]) & 0x10) == bool(first and was_dead), (step, k)
        for i in range(n_obj):  # whole table identical, including bulk evictions
            v = provider.lookup("Obj", str(i))
            assert (v is None and assign[i] == NONE) or v == _addr(int(assign[i]))
        assert np.array_equal(used, oracle.recompute_used(assign, load, m))


def test_tick_equals_reference_policy_when_capacity_is_infinite(oracle):
    """A tick with cap = inf is every object being requested once on its affinity server."""
    rng = np.random.default_rng(7)
    n_obj, m = 500, 9
    alive = np.ones(m, np.uint8)
    alive[[2, 5]] = 0
    cur = rng.integers(0, m, n_obj).astype(np.uint32)
    cur[rng.random(n_obj) < 0.4] = NONE
    live = np.flatnonzero(alive)
    aff = live[rng.integers(0, len(live), n_obj)].astype(np.uint32)
    load = np.ones(n_obj, np.uint32)
    nxt, used, st = oracle.tick(cur, load, aff, np.full(m, INF, np.uint64), alive)
    provider = oracle.LocalObjectPlacement()
    storage = _mk_storage(oracle, alive)
    for i in range(n_obj):
        if cur[i] != NONE:
            provider.update("Obj", str(i), _addr(int(cur[i])))
    for i in range(n_obj):
        got = oracle.get_or_create_placement(provider, storage, _addr(int(aff[i])), "Obj", str(i))
        assert got == _addr(int(nxt[i]))
    assert st["unplaced"] == 0 and st["spilled"] == 0 and st["cut_nodes"] == 0
    assert st["kept"] + st["claimed"] == n_obj
    assert np.array_equal(used, oracle.recompute_used(nxt, load, m))


def test_tick_prefix_cut_known_answer(oracle):
    """free = 100, claimants 60,50,10,10 -> prefix 60 ok, 110 cut: the rest spill (strict prefix)."""
    cur = np.full(4, NONE, np.uint32)
    load = np.array([60, 50, 10, 10], np.uint32)
    aff = np.zeros(4, np.uint32)
    cap = np.array([100, 1000], np.uint64)
    alive = np.ones(2, np.uint8)
    nxt, used, st = oracle.tick(cur, load, aff, cap, alive, rounds=1)
    assert list(nxt) == [0, 1, 1, 1]
    assert list(used) == [60, 70]
    assert st["claimed"] == 1 and st["spilled"] == 3 and st["cut_nodes"] == 1 and st["slow_path"] == 1


def test_tick_waterfill_known_answer(oracle):
    """Spill order = nodes by (free desc, index asc); an object that straddles two nodes' free
    intervals is skipped in that round and retried in the next."""
    # node0 dead -> everything with aff 0 spills.  free: n1=5, n2=8, n3=8 -> order n2,n3,n1; C=[0,8,16,21]
    cur = np.full(5, NONE, np.uint32)
    load = np.array([6, 4, 3, 5, 9], np.uint32)  # Q = 0,6,10,13,18
    aff = np.zeros(5, np.uint32)
    cap = np.array([100, 5, 8, 8], np.uint64)
    alive = np.array([0, 1, 1, 1], np.uint8)
    nxt, used, st = oracle.tick(cur, load, aff, cap, alive, rounds=1)
    # obj0: Q=0 in [0,8) fits (6<=8) -> n2; obj1: Q=6 in [0,8), 10>8 straddles -> none;
    # obj2: Q=10 in [8,16) 13<=16 -> n3; obj3: Q=13 in [8,16) 18>16 -> none; obj4: Q=18 in [16,21) 27>21 none
    assert list(nxt) == [2, NONE, 3, NONE, NONE]
    assert st["rounds_run"] == 1 and st["unplaced"] == 3
    nxt2, used2, st2 = oracle.tick(cur, load, aff, cap, alive, rounds=2)
    # round 2: free n1=5,n2=2,n3=5 -> order n1,n3,n2 C=[0,5,10,12]; rem loads 4,5,9 Q=0,4,9
    # obj1: Q=0 fits n1 (4<=5); obj3: Q=4 in [0,5) 9>5 none; obj4: Q=9 in [5,10) 18>10 none
    assert list(nxt2) == [2, 1, 3, NONE, NONE]
    assert list(used2) == [0, 4, 6, 3]
    assert st2["rounds_run"] == 2 and st2["spilled"] == 3 and st2["unplaced"] == 2


def test_tick_sticky_over_capacity_and_dead_eviction(oracle):
    cur = np.array([0, 0, 1, NONE], np.uint32)
    load = np.array([10, 10, 5, 1], np.uint32)
    aff = np.array([1, 1, 0, 0], np.uint32)
    cap = np.array([5, 100], np.uint64)  # node0 already over capacity: kept objects stay
    alive = np.array([1, 0], np.uint8)   # node1 dead: its object is evicted and re-placed
    nxt, used, st = oracle.tick(cur, load, aff, cap, alive)
    assert list(nxt) == [0, 0, NONE, NONE]  # nothing fits: node0 full, node1 dead
    assert st["kept"] == 2 and st["evicted"] == 1 and st["unplaced"] == 2
    assert list(used) == [20, 0]


@pytest.mark.parametrize("seed", range(6))
def test_tick_invariants_random(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    n_obj, m = int(rng.integers(1, 4000)), int(rng.integers(1, 40))
    cur = rng.integers(0, m, n_obj).astype(np.uint32)
    cur[rng.random(n_obj) < 0.5] = NONE
    load = rng.integers(0, 50, n_obj).astype(np.uint32)
    aff = rng.integers(0, m, n_obj).astype(np.uint32)
    aff[rng.random(n_obj) < 0.05] = NONE
    alive = (rng.random(m) < 0.8).astype(np.uint8)
    cap = rng.integers(0, int(load.sum() * 1.5 / m) + 2, m).astype(np.uint64)
    nxt, used, st = oracle.tick(cur, load, aff, cap, alive, rounds=3)
    kept = (cur != NONE) & (alive[np.minimum(cur, m - 1)] == 1)
    assert np.array_equal(nxt[kept], cur[kept])                       # sticky
    placed = nxt != NONE
    assert np.all(alive[nxt[placed]] == 1)                             # never on a dead node
    assert np.array_equal(used, oracle.recompute_used(nxt, load, m))
    used_kept = oracle.recompute_used(np.where(kept, cur, NONE).astype(np.uint32), load, m)
    for j in range(m):                                                 # new load never exceeds free capacity
        assert used[j] - used_kept[j] <= max(0, int(cap[j]) - int(used_kept[j]))
    assert st["kept"] + st["claimed"] + st["spilled"] + st["unplaced"] == n_obj
    assert st["kept"] == int(kept.sum())
    # idempotence: a second tick on the result changes nothing that was placed
    nxt2, used2, st2 = oracle.tick(nxt, load, aff, cap, alive, rounds=3)
    assert np.array_equal(nxt2[placed], nxt[placed])


def test_place_pending_capacity_and_duplicates(oracle):
    n_obj, m = 6, 3
    assign = np.array([NONE, NONE, NONE, 2, NONE, NONE], np.uint32)
    load = np.array([5, 5, 5, 1, 2, 7], np.uint32)
    cap = np.array([10, 6, 100], np.uint64)
    alive = np.ones(m, np.uint8)
    used = oracle.recompute_used(assign, load, m)
    idx = np.array([0, 1, 0, 2, 3, 4], np.uint32)
    req = np.array([0, 0, 1, 0, 1, 1], np.uint32)
    node, flag = oracle.place_pending(assign, load, cap, alive, used, idx, req, rounds=1)
    # obj0 -> n0 (5<=10) PLACED; obj1 -> n0 (10<=10) PLACED; dup obj0 asked on n1 -> REDIRECT to n0;
    # obj2 on n0: prefix 15 > 10 -> spill: free n2=99,n1=6,n0=0 -> Q=0 -> n2 SPILLED;
    # obj3 sticky on n2, asked on n1 -> REDIRECT; obj4 -> n1 (2<=6) PLACED
    assert list(node) == [0, 0, 0, 2, 2, 1]
    assert list(flag) == [2, 2, 1, 3, 1, 2]
    assert list(assign) == [0, 0, 2, 2, 1, NONE]
    assert list(used) == [10, 2, 6]


def test_invalid_arguments(oracle):
    assign = np.full(4, NONE, np.uint32)
    assert oracle.update_batch(assign, 2, [4], [0]) == 1
    assert oracle.update_batch(assign, 2, [0], [2]) == 1
    assert oracle.remove_batch(assign, [9]) == 1
    with pytest.raises(ValueError):
        oracle.lookup_batch(assign, [4])
    assert np.all(assign == NONE)


def test_place_pending_replaced_flag_known_answer(oracle):
    """service.rs:268-285 folded into the flag: a request that finds its object on a server
    that is not alive gets REPLACED OR-ed onto its outcome — that server is cleaned, the object re-placed; later requests
    for the same object in the batch observe the new placement."""
    assign = np.array([0, 0, 1, NONE], np.uint32)
    load = np.ones(4, np.uint32)
    cap = np.full(3, INF, np.uint64)
    alive = np.array([0, 1, 1], np.uint8)
    used = oracle.recompute_used(assign, load, 3)
    idx = np.array([0, 0, 2, 1, 3], np.uint32)
    req = np.array([1, 2, 1, 2, 2], np.uint32)
    node, flag = oracle.place_pending(assign, load, cap, alive, used, idx, req)
    assert list(node) == [1, 1, 1, 2, 2]
    assert list(flag) == [0x12, 1, 0, 0x12, 2]   # PLACED|REPLACED, REDIRECT, LOCAL, PLACED|REPLACED, PLACED
    assert list(assign) == [1, 2, 1, 2] and list(used) == [0, 2, 2]
