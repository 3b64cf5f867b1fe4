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
