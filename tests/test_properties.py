"""Property-based tests (hypothesis), CPU only.

The load / capacity / spill behaviour of the solver is new (the reference has none): its definition is the C oracle, so the
oracle itself is held to (a) an independent plain-Python restatement of DESIGN.md section 2 on generated tables, (b) the
invariants the specification implies, and (c) the reference policy when every capacity is unbounded.  The string layer
(rio-rs_amd/csrc/gpu_object_placement.cpp) is driven with generated call sequences against the C++ restatement of
LocalObjectPlacement + Service::get_or_create_placement, over the host-memory stub of the dense ABI
(tests/stub_rio_gp.cpp — test infrastructure; the product has no CPU path)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import spec_tick

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NONE = 0xFFFFFFFF
INF = 0xFFFFFFFFFFFFFFFF
COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


@pytest.fixture(scope="module")
def oracle():
    import pyoracle
    pyoracle.build()
    return pyoracle


@st.composite
def tables(draw, max_n=60, max_m=9):
    m = draw(st.integers(1, max_m))
    n = draw(st.integers(0, max_n))
    node_or_none = st.one_of(st.integers(0, m + 1), st.just(NONE))       # m, m + 1: invalid node ids
    cur = draw(st.lists(node_or_none, min_size=n, max_size=n))
    aff = draw(st.lists(st.one_of(node_or_none, st.just(spec_tick.INACTIVE)), min_size=n, max_size=n))
    load = draw(st.lists(st.one_of(st.integers(0, 6), st.integers(0, 5000)), min_size=n, max_size=n))
    alive = draw(st.lists(st.integers(0, 1), min_size=m, max_size=m))
    cap = draw(st.lists(st.one_of(st.integers(0, 40), st.integers(0, 20000), st.just(INF)), min_size=m, max_size=m))
    rounds = draw(st.integers(1, 3))
    return cur, load, aff, cap, alive, rounds


def _np(cur, load, aff, cap, alive):
    return (np.array(cur, np.uint32), np.array(load, np.uint32), np.array(aff, np.uint32), np.array(cap, np.uint64),
            np.array(alive, np.uint8))


@settings(max_examples=300, **COMMON)
@given(tables())
def test_tick_oracle_equals_written_specification(oracle, t):
    cur, load, aff, cap, alive, rounds = t
    want, used, stats = oracle.tick(*_np(cur, load, aff, cap, alive), rounds)
    got, gused = spec_tick.tick(cur, load, aff, cap, alive, rounds)
    assert want.tolist() == got and used.tolist() == gused
    # invariants the specification implies
    n, m = len(cur), len(cap)
    placed = [i for i in range(n) if got[i] != NONE]
    assert all(got[i] < m and alive[got[i]] for i in placed)                       # only live nodes hold objects
    assert sum(gused) == sum(load[i] for i in placed)                              # `used` is the load of the placed rows
    for i in range(n):
        if cur[i] < m and alive[cur[i]]:
            assert got[i] == cur[i]                                                # sticky (service.rs:241-242)
        elif aff[i] == spec_tick.INACTIVE:
            assert got[i] == NONE                                                  # not an object: never placed
    kept = [0] * m
    for i in range(n):
        if cur[i] < m and alive[cur[i]]:
            kept[cur[i]] += load[i]
    for j in range(m):                                                             # capacity is only ever exceeded by kept rows
        assert gused[j] <= max(cap[j], kept[j])
    assert stats["kept"] + stats["claimed"] + stats["spilled"] + stats["unplaced"] == stats["n_objects"]


@settings(max_examples=150, **COMMON)
@given(tables(max_n=40, max_m=6))
def test_tick_with_unbounded_capacity_is_the_reference_policy(oracle, t):
    """cap = infinity: every pending object whose requester is up lands on it (first touch, service.rs:244-252), whatever
    its load — the reduction of the solver to the reference policy."""
    cur, load, aff, _, alive, rounds = t
    m = len(alive)
    cap = [INF] * m
    want, used, stats = oracle.tick(*_np(cur, load, aff, cap, alive), rounds)
    for i in range(len(cur)):
        if cur[i] < m and alive[cur[i]]:
            assert want[i] == cur[i]
        elif aff[i] < m and alive[aff[i]]:
            assert want[i] == aff[i]
    assert stats["cut_nodes"] == 0


@st.composite
def request_batches(draw):
    m = draw(st.integers(1, 6))
    n = draw(st.integers(1, 40))
    assign = draw(st.lists(st.one_of(st.integers(0, m - 1), st.just(NONE)), min_size=n, max_size=n))
    load = draw(st.lists(st.integers(0, 50), min_size=n, max_size=n))
    alive = draw(st.lists(st.integers(0, 1), min_size=m, max_size=m))
    cap = draw(st.lists(st.one_of(st.integers(0, 120), st.just(INF)), min_size=m, max_size=m))
    q = draw(st.integers(1, 50))
    idx = draw(st.lists(st.integers(0, n - 1), min_size=q, max_size=q))           # duplicates on purpose
    req = draw(st.lists(st.integers(0, m - 1), min_size=q, max_size=q))
    return assign, load, cap, alive, idx, req, draw(st.integers(1, 3))


@settings(max_examples=300, **COMMON)
@given(request_batches())
def test_place_pending_oracle_equals_written_contract(oracle, b):
    assign, load, cap, alive, idx, req, rounds = b
    m = len(cap)
    used = [0] * m
    for i, a in enumerate(assign):
        if a != NONE:
            used[a] += load[i]
    a2, u2 = list(assign), list(used)
    snode, sflag = spec_tick.place_pending(a2, load, cap, alive, u2, idx, req, rounds)
    a1, u1 = np.array(assign, np.uint32), np.array(used, np.uint64)
    onode, oflag = oracle.place_pending(a1, np.array(load, np.uint32), np.array(cap, np.uint64), np.array(alive, np.uint8), u1,
                                        np.array(idx, np.uint32), np.array(req, np.uint32), rounds)
    assert onode.tolist() == snode and oflag.tolist() == sflag
    assert a1.tolist() == a2 and [int(x) for x in u1] == u2
    # every request for one object reports the same node; later duplicates only observe
    first = {}
    for k, i in enumerate(idx):
        if i in first:
            assert snode[k] == snode[first[i]] and (sflag[k] & 0x0F) in (0, 1, 4)
        else:
            first[i] = k


# ---- string layer over the dense-ABI stub ---------------------------------------------------------------------------

@pytest.fixture(scope="module")
def oplib(tmp_path_factory):
    out = tmp_path_factory.mktemp("stub") / "libstub_op.so"
    srcs = [os.path.join(ROOT, "rio-rs_amd", "csrc", "gpu_object_placement.cpp"), os.path.join(ROOT, "tests", "stub_rio_gp.cpp")]
    subprocess.run(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-pthread", "-I", os.path.join(ROOT, "include")] + srcs +
                   ["-o", str(out)], check=True)
    L = C.CDLL(str(out))
    vp = C.c_void_p
    L.rio_op_create.argtypes = [vp, C.POINTER(vp)]
    L.rio_op_release.argtypes = [vp]
    L.rio_op_release.restype = None
    L.rio_op_update.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p]
    L.rio_op_lookup.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
    L.rio_op_last_address_len.argtypes = [vp]
    L.rio_op_last_address_len.restype = C.c_size_t
    L.rio_op_clean_server.argtypes = [vp, C.c_char_p]
    L.rio_op_remove.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.rio_op_set_member.argtypes = [vp, C.c_char_p, C.c_int, C.c_uint64]
    L.rio_op_get_or_create_placement.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32)]
    L.rio_op_len.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.rio_op_try_lookup_n.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
    L.rio_op_try_get_or_create_placement_n.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p,
                                                       C.c_size_t, C.POINTER(C.c_uint32)]
    return L


class _Cfg(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("max_objects", C.c_uint64), ("max_nodes", C.c_uint32),
                ("spill_rounds", C.c_uint32), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class _Op:
    def __init__(self, L, max_objects, max_nodes):
        self.L, self.h = L, C.c_void_p()
        cfg = _Cfg(C.sizeof(_Cfg), 0, max_objects, max_nodes, 0, 0, 0)
        assert L.rio_op_create(C.byref(cfg), C.byref(self.h)) == 0

    def lookup(self, ty, oid, cap=32):
        while True:
            buf, found = C.create_string_buffer(cap), C.c_int(0)
            rc = self.L.rio_op_lookup(self.h, ty.encode(), oid.encode(), buf, cap, C.byref(found))
            if rc == 5:  # RIO_GP_ERANGE: nothing copied, the length is reported
                assert buf.value == b"" and found.value == 1
                cap = self.L.rio_op_last_address_len(self.h) + 1
                continue
            assert rc == 0
            return buf.value.decode() if found.value else None

    def close(self):
        self.L.rio_op_release(self.h)


ADDRS = ["10.0.0.1:5000", "10.0.0.2:5000", "h3:1", "a-rather-long-host-name-that-does-not-fit-thirty-two-bytes.example:65535"]
KEYS = [("T", "1"), ("T", "2"), ("a.b", "c"), ("a", "b.c"), ("U", "")]   # ("a.b","c") and ("a","b.c") collide (local.rs:26-29)
OPS = st.lists(st.one_of(
    st.tuples(st.just("update"), st.integers(0, len(KEYS) - 1), st.one_of(st.none(), st.integers(0, len(ADDRS) - 1))),
    st.tuples(st.just("lookup"), st.integers(0, len(KEYS) - 1), st.none()),
    st.tuples(st.just("remove"), st.integers(0, len(KEYS) - 1), st.none()),
    st.tuples(st.just("clean"), st.integers(0, len(ADDRS) - 1), st.none()),
    st.tuples(st.just("active"), st.integers(0, len(ADDRS) - 1), st.integers(0, 1)),
    st.tuples(st.just("request"), st.integers(0, len(KEYS) - 1), st.integers(0, len(ADDRS) - 1)),
), min_size=1, max_size=40)


@settings(max_examples=200, **COMMON)
@given(OPS)
def test_string_layer_equals_the_reference_restatement(oracle, oplib, ops):
    """Any sequence of trait calls, membership changes and policy requests: the string layer over the stub answers exactly
    like LocalObjectPlacement + Service::get_or_create_placement restated (local.rs:22-68, service.rs:193-254)."""
    op = _Op(oplib, 16, 8)
    ref, members = oracle.LocalObjectPlacement(), oracle.LocalStorage()
    active = {}
    for a in ADDRS:                       # every address is a member; `active` flips below
        ip, port = a.rsplit(":", 1)
        members.push(ip, port, True)
        active[a] = True
        assert oplib.rio_op_set_member(op.h, a.encode(), 1, INF) == 0
    try:
        for kind, x, y in ops:
            if kind == "update":
                ty, oid = KEYS[x]
                addr = None if y is None else ADDRS[y]
                assert oplib.rio_op_update(op.h, ty.encode(), oid.encode(), None if addr is None else addr.encode()) == 0
                ref.update(ty, oid, addr)
            elif kind == "lookup":
                # the non-blocking twin first: RIO_GP_EAGAIN (6), or the answer — the reference's, like the blocking call's
                ty, oid = (v.encode() for v in KEYS[x])
                buf, found = C.create_string_buffer(128), C.c_int(0)
                rc = oplib.rio_op_try_lookup_n(op.h, ty, len(ty), oid, len(oid), buf, 128, C.byref(found))
                assert rc in (0, 6)
                if rc == 0:
                    assert (buf.value.decode() if found.value else None) == ref.lookup(*KEYS[x])
                assert op.lookup(*KEYS[x]) == ref.lookup(*KEYS[x])
            elif kind == "remove":
                assert oplib.rio_op_remove(op.h, KEYS[x][0].encode(), KEYS[x][1].encode()) == 0
                ref.remove(*KEYS[x])
            elif kind == "clean":
                assert oplib.rio_op_clean_server(op.h, ADDRS[x].encode()) == 0
                ref.clean_server(ADDRS[x])
            elif kind == "active":
                ip, port = ADDRS[x].rsplit(":", 1)
                members.set_is_active(ip, port, bool(y))
                active[ADDRS[x]] = bool(y)
                assert oplib.rio_op_set_member(op.h, ADDRS[x].encode(), y, INF) == 0
            else:  # a request arriving at ANY member: the reference first-touches self.address whatever membership says about it
                me = ADDRS[y]  # (service.rs:244-252) — and so does the string layer by default (round-4 verdict, item 6)
                ty, oid = KEYS[x]
                tbuf, tflag = C.create_string_buffer(128), C.c_uint32(0)
                trc = oplib.rio_op_try_get_or_create_placement_n(op.h, ty.encode(), len(ty.encode()), oid.encode(), len(oid.encode()),
                                                                 me.encode(), tbuf, 128, C.byref(tflag))
                assert trc in (0, 6)
                buf, flag = C.create_string_buffer(128), C.c_uint32(0)
                assert oplib.rio_op_get_or_create_placement(op.h, ty.encode(), oid.encode(), me.encode(), buf, 128, C.byref(flag)) == 0
                want = oracle.get_or_create_placement(ref, members, me, ty, oid)
                assert buf.value.decode() == want
                if trc == 0:   # (the sticky branch: nothing changes, so the blocking call behind it says the same)
                    assert tbuf.value.decode() == want and tflag.value == flag.value and tflag.value in (0, 1)
        for key in KEYS:
            assert op.lookup(*key) == ref.lookup(*key)
        n = C.c_uint64(0)
        assert oplib.rio_op_len(op.h, C.byref(n)) == 0 and n.value == len(ref)
    finally:
        op.close()
