"""The C oracle against an independent plain-Python restatement of DESIGN.md section 2 (tests/spec_tick.py)."""
import numpy as np
import pytest

import spec_tick

NONE = 0xFFFFFFFF


@pytest.fixture(scope="module")
def oracle():
    import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.mark.parametrize("seed", range(40))
def test_c_oracle_equals_the_written_specification(oracle, seed):
    rng = np.random.default_rng(9000 + seed)
    n, m = int(rng.integers(0, 400)), int(rng.integers(1, 24))
    cur = rng.integers(0, m + 2, n).astype(np.uint32)           # m, m+1: invalid node ids
    cur[rng.random(n) < 0.5] = NONE
    load = rng.integers(0, int(rng.choice([3, 50, 4000])), n).astype(np.uint32)   # zero loads included
    aff = rng.integers(0, m + 1, n).astype(np.uint32)
    aff[rng.random(n) < 0.05] = NONE
    if seed % 3 == 0:
        aff[rng.random(n) < 0.15] = spec_tick.INACTIVE            # rows that are not objects (row lifecycle)
    alive = (rng.random(m) < 0.8).astype(np.uint8)
    scale = float(rng.choice([0.0, 0.4, 1.0, 3.0]))
    cap = rng.integers(0, int(load.sum() * scale / m) + 2, m).astype(np.uint64)
    if seed % 7 == 0:
        cap[rng.integers(0, m)] = np.uint64(0xFFFFFFFFFFFFFFFF)   # an unbounded node: saturating cumulative capacity
    rounds = int(rng.integers(1, 4))
    want, used, st = oracle.tick(cur, load, aff, cap, alive, rounds)
    got, gused = spec_tick.tick(cur.tolist(), load.tolist(), aff.tolist(), cap.tolist(), alive.tolist(), rounds)
    assert got == want.tolist()
    assert gused == used.tolist()
    # ... and under RIO_GP_CFG_REF_SELF_ASSIGN (a claim does not need an active node: service.rs:244-252)
    want_sa, used_sa, _ = oracle.tick(cur, load, aff, cap, alive, rounds, flags=oracle.REF_SELF_ASSIGN)
    got_sa, gused_sa = spec_tick.tick(cur.tolist(), load.tolist(), aff.tolist(), cap.tolist(), alive.tolist(), rounds, self_assign=True)
    assert got_sa == want_sa.tolist() and gused_sa == used_sa.tolist()
    placed = [g != NONE for g in got]
    inactive = sum(1 for i in range(n) if aff[i] == spec_tick.INACTIVE and got[i] == NONE)
    assert st["n_objects"] == n - inactive
    assert st["kept"] + st["claimed"] + st["spilled"] == sum(placed) and st["unplaced"] == st["n_objects"] - sum(placed)


@pytest.mark.parametrize("seed", range(40))
def test_c_oracle_place_pending_equals_the_written_contract(oracle, seed):
    rng = np.random.default_rng(7000 + seed)
    n, m = int(rng.integers(1, 300)), int(rng.integers(1, 16))
    load = rng.integers(0, int(rng.choice([3, 60, 3000])), n).astype(np.uint32)
    alive = (rng.random(m) < 0.75).astype(np.uint8)
    if not alive.any():
        alive[0] = 1
    assign = rng.integers(0, m, n).astype(np.uint32)            # some rows sit on dead nodes
    assign[rng.random(n) < 0.5] = NONE
    scale = float(rng.choice([0.3, 1.0, 2.5]))
    cap = rng.integers(0, int(load.sum() * scale / m) + 2, m).astype(np.uint64)
    if seed % 5 == 0:
        cap[:] = np.uint64(0xFFFFFFFFFFFFFFFF)
    used = np.zeros(m, np.uint64)
    for i in range(n):
        if assign[i] != NONE:
            used[assign[i]] += np.uint64(load[i])
    q = int(rng.integers(1, 120))
    idx = rng.integers(0, n, q).astype(np.uint32)               # duplicates on purpose
    req = rng.integers(0, m, q).astype(np.uint32)               # requesters may be inactive members
    rounds = int(rng.integers(1, 4))
    a2, u2 = assign.tolist(), [int(x) for x in used]
    snode, sflag = spec_tick.place_pending(a2, load.tolist(), cap.tolist(), alive.tolist(), u2, idx.tolist(), req.tolist(), rounds)
    a1, u1 = assign.copy(), used.copy()
    onode, oflag = oracle.place_pending(a1, load, cap, alive, u1, idx, req, rounds)
    assert onode.tolist() == snode and oflag.tolist() == sflag
    a3, u3 = assign.tolist(), [int(x) for x in used]
    snode, sflag = spec_tick.place_pending(a3, load.tolist(), cap.tolist(), alive.tolist(), u3, idx.tolist(), req.tolist(), rounds,
                                           self_assign=True)
    a4, u4 = assign.copy(), used.copy()
    onode, oflag = oracle.place_pending(a4, load, cap, alive, u4, idx, req, rounds, flags=oracle.REF_SELF_ASSIGN)
    assert onode.tolist() == snode and oflag.tolist() == sflag and a4.tolist() == a3 and u4.tolist() == u3
    assert a1.tolist() == a2 and [int(x) for x in u1] == u2


@pytest.mark.parametrize("seed", range(20))
def test_capacity_class_order_stays_within_a_fifth_of_the_exact_order(oracle, seed):
    """The water-fill takes the nodes by capacity CLASS (free capacity rounded down to three significant bits), not by exact
    free capacity — the order exists to send the spill to the emptiest nodes first, and a class order can be counted instead
    of sorted (rule 3; frozen).  What that gives up, pinned here so that it cannot grow silently:
      * the class is monotone in the free capacity, and two nodes of one class differ by less than a factor 5/4 — so every
        node the water-fill takes has MORE THAN 4/5 of the free capacity of any node it takes later;
      * on a table whose rows all spill, both orders place every row when capacity suffices (the interval rule fills the
        nodes it takes first to the brim under either order: what differs is only WHICH of two nearly equally empty nodes
        that is).
    The exact-order water-fill is written out here, independently (rounds 1-2 of this repository had it in the oracle)."""
    rng = np.random.default_rng(12000 + seed)
    m = int(rng.integers(2, 200))
    free = [int(x) for x in rng.integers(1, 1 << int(rng.integers(3, 40)), m)]
    order = sorted(range(m), key=lambda j: (-spec_tick.capacity_class(free[j]), j))
    for a, b in zip(order, order[1:]):
        assert spec_tick.capacity_class(free[a]) >= spec_tick.capacity_class(free[b])
        assert 5 * free[a] > 4 * free[b], (free[a], free[b])              # never less than 4/5 of a later node's room
    for a in range(m):                                                        # monotone: more room is never a lower class
        for b in range(m):
            if free[a] >= free[b]:
                assert spec_tick.capacity_class(free[a]) >= spec_tick.capacity_class(free[b])
    # behaviour: all rows pending without a live affinity node (pure spill), unit loads, capacity for everybody
    n = int(sum(free) * 0.6) if sum(free) < 20000 else 12000
    cap = np.array(free, np.uint64)
    if n == 0 or n > 20000:
        return
    cur = np.full(n, NONE, np.uint32)
    aff = np.full(n, NONE, np.uint32)
    load = np.ones(n, np.uint32)
    alive = np.ones(m, np.uint8)
    got, used, st = oracle.tick(cur, load, aff, cap, alive, 8)
    assert st["unplaced"] == 0 and st["spilled"] == n                        # capacity suffices: everything is placed
    # exact order (free capacity descending, index ascending), same interval rule, same rounds
    used_x = [0] * m
    rest = list(range(n))
    for _ in range(8):
        if not rest:
            break
        fr = [free[j] - used_x[j] for j in range(m)]
        ordx = sorted((j for j in range(m) if fr[j] > 0), key=lambda j: (-fr[j], j))
        C = [0]
        for j in ordx:
            C.append(C[-1] + fr[j])
        q, left, k = 0, [], 0
        for i in rest:
            while k + 1 < len(C) - 1 and C[k + 1] <= q:
                k += 1
            if ordx and q < C[-1] and q + 1 <= C[k + 1]:
                used_x[ordx[k]] += 1
            else:
                left.append(i)
            q += 1
        rest = left
    assert not rest and sum(used_x) == n == int(used.sum())
    assert all(int(used[j]) <= free[j] and used_x[j] <= free[j] for j in range(m))
    # the nodes either order leaves untouched are emptier ones: whatever the class order skipped has less than 5/4 of the room
    # of the fullest-capacity node the exact order used, and vice versa
    took_c = [j for j in range(m) if used[j] > 0]
    took_x = [j for j in range(m) if used_x[j] > 0]
    if took_c and took_x and len(took_c) < m and len(took_x) < m:
        skipped_c = max(free[j] for j in range(m) if used[j] == 0)
        assert 5 * min(free[j] for j in took_c) > 4 * skipped_c or skipped_c <= min(free[j] for j in took_c)
