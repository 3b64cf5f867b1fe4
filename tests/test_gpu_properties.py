"""Property-based parity on the GPU (hypothesis): generated tables and request batches through the C ABI
of the product library, every answer against the CPU oracle — the shapes fixed seeds do not think of (one node, no rows,
zero capacities, every node dead, loads of zero, invalid current nodes, rows that are not objects, the same object asked for
forty times in a batch).  Handles are reused across examples: the tables are tiny, a call is what an example costs."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

pytestmark = pytest.mark.gpu

NONE = 0xFFFFFFFF
INACTIVE = 0xFFFFFFFE
INF = 0xFFFFFFFFFFFFFFFF
MAX_N, MAX_M = 700, 48
COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


@pytest.fixture(scope="module")
def gp():
    import rio_gp
    rio_gp.build()
    return rio_gp


@pytest.fixture(scope="module")
def handles(gp):
    hs = {r: gp.GpuPlacement(MAX_N, MAX_M, spill_rounds=r) for r in (1, 2, 3)}
    yield hs
    for h in hs.values():
        h.close()


@st.composite
def tables(draw):
    m = draw(st.integers(1, MAX_M))
    n = draw(st.one_of(st.integers(0, 40), st.integers(0, MAX_N)))
    node_or_none = st.one_of(st.integers(0, m - 1), st.just(NONE))
    cur = draw(st.lists(node_or_none, min_size=n, max_size=n))
    aff = draw(st.lists(st.one_of(node_or_none, st.just(INACTIVE)), min_size=n, max_size=n))
    load = draw(st.lists(st.one_of(st.integers(0, 6), st.integers(0, 5000)), min_size=n, max_size=n))
    alive = draw(st.lists(st.integers(0, 1), min_size=m, max_size=m))
    cap = draw(st.lists(st.one_of(st.integers(0, 40), st.integers(0, 200000), st.just(INF)), min_size=m, max_size=m))
    return (np.array(cur, np.uint32), np.array(load, np.uint32), np.array(aff, np.uint32), np.array(cap, np.uint64),
            np.array(alive, np.uint8), draw(st.integers(1, 3)))


@settings(max_examples=250, **COMMON)
@given(tables())
def test_tick_equals_oracle_on_generated_tables(handles, oracle, t):
    cur, load, aff, cap, alive, rounds = t
    n, m = len(cur), len(cap)
    g = handles[rounds]
    g.set_nodes(cap, alive, m=m)
    g.set_objects(n, load, aff)
    if n:
        g.set_assign(cur)
    want, used, ost = oracle.tick(cur, load, aff, cap, alive, rounds)
    st1 = g.solve()
    assert np.array_equal(g.get_solved(), want) and st1 == ost
    g.commit()
    assert np.array_equal(g.get_assign(), want) and np.array_equal(g.get_nodes()[2], used)
    # a second tick consumes the first one's table: asynchronous form, counters read afterwards
    want2, used2, ost2 = oracle.tick(want, load, aff, cap, alive, rounds)
    g.tick_async()
    got = g.tick_wait()
    assert got == [ost2] and np.array_equal(g.get_assign(), want2) and np.array_equal(g.get_nodes()[2], used2)


@st.composite
def request_batches(draw):
    m = draw(st.integers(1, 12))
    n = draw(st.integers(1, 120))
    assign = draw(st.lists(st.one_of(st.integers(0, m - 1), st.just(NONE)), min_size=n, max_size=n))
    load = draw(st.lists(st.integers(0, 50), min_size=n, max_size=n))
    alive = draw(st.lists(st.integers(0, 1), min_size=m, max_size=m))
    cap = draw(st.lists(st.one_of(st.integers(0, 300), st.just(INF)), min_size=m, max_size=m))
    q = draw(st.one_of(st.integers(1, 6), st.integers(1, 300)))
    idx = draw(st.lists(st.integers(0, n - 1), min_size=q, max_size=q))           # duplicates on purpose
    req = draw(st.lists(st.integers(0, m - 1), min_size=q, max_size=q))
    return (np.array(assign, np.uint32), np.array(load, np.uint32), np.array(cap, np.uint64), np.array(alive, np.uint8),
            np.array(idx, np.uint32), np.array(req, np.uint32))


@settings(max_examples=250, **COMMON)
@given(request_batches(), request_batches())
def test_place_pending_equals_oracle_on_generated_batches(handles, oracle, b1, b2):
    """Two batches in a row on one table (the second one's shape, the first one's table where it fits): micro-batches through
    the one-workgroup kernel, its hand-over to the general path (dead nodes in the way, dead or full requesters), the
    REPLACED flag, `used` kept up to date across calls."""
    assign, load, cap, alive, idx, req = b1
    n, m = len(assign), len(cap)
    g = handles[2]
    g.set_nodes(cap, alive, m=m)
    g.set_objects(n, load, None)
    g.set_assign(assign)
    ref = assign.copy()
    used = oracle.recompute_used(ref, load, m)
    for k, (ii, rr) in enumerate(((idx, req), (b2[4] % np.uint32(n), b2[5] % np.uint32(m)))):
        node, flag = g.place_pending(ii, rr)
        wnode, wflag = oracle.place_pending(ref, load, cap, alive, used, ii, rr)
        assert np.array_equal(node, wnode) and np.array_equal(flag, wflag), k
        assert np.array_equal(g.get_assign(), ref) and np.array_equal(g.get_nodes()[2], used), k
