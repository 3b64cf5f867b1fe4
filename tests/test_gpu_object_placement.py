"""The reference's own ObjectPlacement tests, re-run against the GPU-backed provider through the
string-level C ABI (include/rio_gpu_object_placement.h).  Each test names the reference test it
transcribes.  Run on the GPU box: pytest -m gpu."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sql_backend_golden.json")


@pytest.fixture(scope="module")
def gp():
    import rio_gp
    rio_gp.build()
    return rio_gp


# rio-rs/tests/object_placement_backend.rs:11-16  no_placement
def test_no_placement(gp):
    provider = gp.GpuObjectPlacement()
    provider.prepare()
    assert provider.lookup("obj", "1") is None


# rio-rs/tests/object_placement_backend.rs:18-34  save_and_load
def test_save_and_load(gp):
    provider = gp.GpuObjectPlacement()
    provider.prepare()
    provider.update("obj", "1", "0.0.0.0:8888")
    assert provider.lookup("obj", "1") == "0.0.0.0:8888"
    provider.clean_server("0.0.0.0:8888")
    assert provider.lookup("obj", "1") is None


# rio-rs/src/object_placement/local.rs:71-123  local_object_placement_provider_is_clonable
def test_provider_is_clonable(gp):
    provider = gp.GpuObjectPlacement()
    cloned = provider.clone()
    provider.update("test", "1", "0.0.0.0:80")
    assert provider.lookup("test", "1") is not None
    assert cloned.lookup("test", "1") is not None
    cloned.clean_server("0.0.0.0:80")
    assert provider.lookup("test", "1") is None
    assert cloned.lookup("test", "1") is None
    provider.close()                      # dropping one clone keeps the shared state alive
    cloned.update("test", "2", "0.0.0.0:80")
    assert cloned.lookup("test", "2") == "0.0.0.0:80"


# rio-rs/src/object_placement/sqlite.rs:149-193  test_sanity
def test_sanity_upsert_overwrites(gp):
    p = gp.GpuObjectPlacement()
    assert p.lookup("Test", "1") is None
    p.update("Test", "1", "0.0.0.0:5000")
    assert p.lookup("Test", "1") == "0.0.0.0:5000"
    p.update("Test", "1", "0.0.0.0:5001")
    assert p.lookup("Test", "1") == "0.0.0.0:5001"
    p.clean_server("0.0.0.0:5001")
    assert p.lookup("Test", "1") is None


# local.rs:36-37 update(None) deletes; local.rs:60-68 remove of an absent key; local.rs:26-29 key join quirk
def test_none_remove_and_key_quirk(gp):
    p = gp.GpuObjectPlacement()
    p.remove("a", "1")
    p.update("a", "1", None)
    p.update("a", "1", "h:1")
    assert len(p) == 1
    p.update("a", "1", None)
    assert p.lookup("a", "1") is None and len(p) == 0
    p.update("a.b", "c", "h:1")
    assert p.lookup("a", "b.c") == "h:1"
    p.clean_server("never-seen:1")      # retain() on an address nothing lives on


# the reference's SQL semantics (golden file) replayed through the string layer
def test_sql_golden_through_string_layer(gp):
    doc = json.load(open(GOLD))
    for case in doc["cases"][:3]:
        p = gp.GpuObjectPlacement(max_objects=4096, max_nodes=64)
        got = []
        for op in case["ops"]:
            if op[0] == "lookup":
                got.append(p.lookup(op[1], op[2]))
            else:
                getattr(p, op[0])(*op[1:])
        assert got == case["expected_lookups"]
        p.close()


# rio-rs/tests/object_allocation.rs:75-137  move_object_on_server_failure (policy service.rs:193-254)
def test_move_object_on_server_failure(gp):
    p = gp.GpuObjectPlacement()
    p.set_member("0.0.0.0:7001", True)
    p.set_member("0.0.0.0:7002", True)
    assert p.lookup("MockService", "1") is None                       # starts not allocated
    first, flag = p.get_or_create_placement("MockService", "1", "0.0.0.0:7001")
    assert first == "0.0.0.0:7001" and flag == gp.FLAG_PLACED          # first message allocates it
    assert p.lookup("MockService", "1") == first
    again, flag = p.get_or_create_placement("MockService", "1", "0.0.0.0:7002")
    assert again == first and flag == gp.FLAG_REDIRECT                 # ResponseError::Redirect
    p.set_member("0.0.0.0:7001", False)                                # the host dies
    second, flag = p.get_or_create_placement("MockService", "1", "0.0.0.0:7002")
    assert second == "0.0.0.0:7002" and flag == gp.FLAG_PLACED | gp.FLAG_REPLACED   # found on the dead host, re-placed
    assert first != second                                             # assert_ne!(first_server, second_server)


def test_keys_with_an_interior_nul_are_keys_of_their_own(gp):
    """ObjectId(String, String) holds any Rust string (service_object.rs:19-26), a NUL byte included; the length-carrying
    entry points (rio_op_*_n, what the Rust adapter and this binding use) keep such keys apart from their prefixes, through
    every trait method, the policy call and the snapshot — and the NUL-terminated entry points still see the prefix only."""
    import ctypes as C
    p = gp.GpuObjectPlacement(max_objects=64, max_nodes=8)
    p.set_member("h:1", True)
    p.update("T", "a\0b", "h:1")
    assert p.lookup("T", "a\0b") == "h:1" and p.lookup("T", "a") is None and p.lookup("T", "a\0c") is None
    p.update("T\0x", "a", "h:2")                                   # ... in the struct name as well
    assert p.lookup("T\0x", "a") == "h:2" and p.lookup("T", "a") is None
    got, flag = p.get_or_create_placement("T", "a\0c", "h:1")
    assert got == "h:1" and flag == gp.FLAG_PLACED and p.lookup("T", "a\0c") == "h:1"
    assert sorted(p.snapshot()) == [("T", "a\0b", "h:1"), ("T", "a\0c", "h:1"), ("T\0x", "a", "h:2")]
    p.remove("T", "a\0b")
    assert p.lookup("T", "a\0b") is None and p.lookup("T", "a\0c") == "h:1" and len(p) == 2
    # the NUL-terminated twin of the same call addresses the key that ends at the NUL
    found, buf = C.c_int(0), C.create_string_buffer(64)
    assert gp._oplib().rio_op_lookup(p._h, b"T", b"a\0c", buf, 64, C.byref(found)) == 0 and found.value == 0
    # ... and the BATCHED calls carry the lengths too (rio_op_*_batch_n; round-4 advisor finding: update_batch cut "a\0b" down to
    # "a", so a snapshot written with the key and loaded back through update_batch overwrote another object)
    p.set_member("h:2", True)
    p.update_batch([("T", "k\0one"), ("T", "k"), ("T\0y", "k\0one")], ["h:1", "h:2", None])
    assert p.lookup_batch([("T", "k\0one"), ("T", "k"), ("T", "k\0two"), ("T\0y", "k\0one")]) == ["h:1", "h:2", None, None]
    got, flags = p.get_or_create_placement_batch([("T", "k\0two"), ("T", "k\0one"), ("T", "k")], ["h:1", "h:1", "h:1"])
    assert got == ["h:1", "h:1", "h:2"] and list(flags) == [gp.FLAG_PLACED, gp.FLAG_LOCAL, gp.FLAG_REDIRECT]
    p.set_object_load("T", "w\0x", 7)
    p.update("T", "w", "h:1")                                       # (a key of its own: the load stays with "w\0x")
    idx = {k: v for v, k in enumerate([r[:2] for r in p.snapshot()])}
    assert ("T", "k\0one") in idx and ("T", "k\0two") in idx and ("T", "w") in idx and ("T", "w\0x") not in idx
    with pytest.raises(ValueError):
        p.update_batch([("T", "z")], ["h\0:1"])                     # an address cannot hold a NUL ("{ip}:{port}" of a Member)
    # round trip through the SQLite twin of the table: every key comes back as the key it was
    import os, tempfile, snapshot
    before = sorted(p.snapshot())
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "placement.sqlite3")
        snapshot.dump_sqlite(p, path)
        q = gp.GpuObjectPlacement(max_objects=64, max_nodes=8)
        snapshot.load_sqlite(q, path)
        assert sorted(q.snapshot()) == before
        q.close()
    p.close()


# service.rs:213-223: a malformed record is removed (only that record) and the object re-placed
def test_policy_bad_record_removed(gp, oracle):
    p, o = gp.GpuObjectPlacement(), oracle.LocalObjectPlacement()
    st = oracle.LocalStorage()
    st.push("h", 1)
    p.set_member("h:1", True)
    for prov in (p, o):
        prov.update("T", "x", "nocolon")
        prov.update("T", "y", "nocolon")
    got, _ = p.get_or_create_placement("T", "x", "h:1")
    assert got == oracle.get_or_create_placement(o, st, "h:1", "T", "x") == "h:1"
    assert p.lookup("T", "y") == o.lookup("T", "y") == "nocolon"       # the other bad record is untouched


@pytest.mark.parametrize("seed,self_assign", [(0, False), (1, False), (2, True), (3, True)] + [(s, s % 2 == 1) for s in range(4, 12)])
def test_random_differential_vs_reference_restatement(gp, oracle, seed, self_assign):
    """Random trait calls + policy requests: the GPU provider and the C++ restatement of
    LocalObjectPlacement + service.rs policy must agree on every observable.  self_assign: the provider is created with its
    DEFAULT flags (the reference's behaviour) and the requests come from ANY member, inactive ones included — the reference
    first-touches the requester without asking whether it is active (service.rs:244-252), and so does the provider; the
    other seeds opt out (RIO_OP_CFG_LIVE_FIRST_TOUCH) and keep to requesters that are up."""
    rng = np.random.default_rng(seed)
    addrs = ["10.0.0.%d:%d" % (k, 5000 + k) for k in range(6)]
    p = gp.GpuObjectPlacement(max_objects=2048, max_nodes=32, flags=0 if self_assign else gp.OP_CFG_LIVE_FIRST_TOUCH)
    o, st = oracle.LocalObjectPlacement(), oracle.LocalStorage()
    alive = [True] * len(addrs)
    for a in addrs:
        ip, port = a.split(":")
        st.push(ip, port, True)
        p.set_member(a, True)
    keys = [("T%d" % (k % 3), str(k)) for k in range(120)]
    for step in range(400):
        r = rng.random()
        ty, oid = keys[int(rng.integers(len(keys)))]
        if r < 0.15:
            a = addrs[int(rng.integers(len(addrs)))]
            p.update(ty, oid, a)
            o.update(ty, oid, a)
        elif r < 0.25:
            p.remove(ty, oid)
            o.remove(ty, oid)
        elif r < 0.30:
            a = addrs[int(rng.integers(len(addrs)))]
            p.clean_server(a)
            o.clean_server(a)
        elif r < 0.36:
            k = int(rng.integers(len(addrs)))
            alive[k] = not alive[k]
            if not any(alive):
                alive[k] = True
            ip, port = addrs[k].split(":")
            st.set_is_active(ip, port, alive[k])
            p.set_member(addrs[k], alive[k])
        elif r < 0.42 and not self_assign:
            # the batched form (rio_op_get_or_create_placement_batch): n requests "as if sequentially in array order" — the
            # restatement serves them one by one
            kk = int(rng.integers(2, 40))
            bkeys = [keys[int(rng.integers(len(keys)))] for _ in range(kk)]
            live = [k for k in range(len(addrs)) if alive[k]]
            mes = [addrs[int(rng.choice(live))] for _ in range(kk)]
            got, flags = p.get_or_create_placement_batch(bkeys, mes)
            for (bty, boid), me, g1, f1 in zip(bkeys, mes, got, flags):
                want = oracle.get_or_create_placement(o, st, me, bty, boid)
                assert g1 == want, (step, bty, boid, me)
                verdict = oracle.check_address_mismatch(o, st, me, want)
                assert verdict == ("ok" if int(f1) & gp.FLAG_MASK in (gp.FLAG_LOCAL, gp.FLAG_PLACED) else "redirect"), (step, bty, boid)
        elif r < 0.75:
            me = addrs[int(rng.choice([k for k in range(len(addrs)) if alive[k] or self_assign]))]
            got, flag = p.get_or_create_placement(ty, oid, me)
            want = oracle.get_or_create_placement(o, st, me, ty, oid)
            assert got == want, (step, ty, oid)
            verdict = oracle.check_address_mismatch(o, st, me, want)
            if alive[addrs.index(want)] or want == me:   # (placed on an inactive requester other than me: the reference would clean it)
                assert verdict == ("ok" if flag & gp.FLAG_MASK in (gp.FLAG_LOCAL, gp.FLAG_PLACED) else "redirect")
        else:
            assert p.lookup(ty, oid) == o.lookup(ty, oid), step
    for ty, oid in keys:
        assert p.lookup(ty, oid) == o.lookup(ty, oid)
    assert len(p) == len(o)
    assert p.lookup_batch(keys) == [o.lookup(*k) for k in keys]


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_host_shadow_answers_exactly_what_the_device_would(gp, seed):
    """The host shadow (rio_gpu_object_placement.h: RIO_OP_CFG_NO_HOST_SHADOW) is a cache of device answers, never a decision:
    two providers fed the same random stream — trait calls, batched calls, membership flips, clean_server, policy requests from
    servers that are up or not, whole-table ticks, key reclaim on a small table — one with the shadow, one without, must agree
    on every answer and every flag; and the one with the shadow must have gone to the device far less often."""
    rng = np.random.default_rng(900 + seed)
    addrs = ["10.1.0.%d:%d" % (k, 7000 + k) for k in range(7)] + ["nocolon"]
    a = gp.GpuObjectPlacement(max_objects=160, max_nodes=16)
    b = gp.GpuObjectPlacement(max_objects=160, max_nodes=16, flags=gp.OP_CFG_NO_HOST_SHADOW)
    alive = [True] * len(addrs)
    for x in addrs[:-1]:
        for p in (a, b):
            p.set_member(x, True, capacity=40 if seed % 2 else gp.CAP_INF)     # (odd seeds: capacities bind, rows spill)
    keys = [("T%d" % (k % 3), "k%d" % k) for k in range(140)]
    tries = {"lookup": 0, "request": 0}
    for step in range(700):
        r = rng.random()
        ty, oid = keys[int(rng.integers(len(keys)))]
        if r < 0.10:
            x = addrs[int(rng.integers(len(addrs)))]
            a.update(ty, oid, x); b.update(ty, oid, x)
        elif r < 0.16:
            a.remove(ty, oid); b.remove(ty, oid)
        elif r < 0.19:
            x = addrs[int(rng.integers(len(addrs)))]
            a.clean_server(x); b.clean_server(x)
        elif r < 0.24:
            k = int(rng.integers(len(addrs) - 1))
            alive[k] = not alive[k]
            a.set_member(addrs[k], alive[k]); b.set_member(addrs[k], alive[k])
        elif r < 0.27:
            assert a.tick() == b.tick(), step
        elif r < 0.31:
            kk = int(rng.integers(2, 30))
            bk = [keys[int(rng.integers(len(keys)))] for _ in range(kk)]
            me = [addrs[int(rng.integers(len(addrs) - 1))] for _ in range(kk)]
            ga, fa = a.get_or_create_placement_batch(bk, me)
            gb, fb = b.get_or_create_placement_batch(bk, me)
            assert ga == gb and list(fa) == list(fb), step
        elif r < 0.35:
            bk = [keys[int(rng.integers(len(keys)))] for _ in range(int(rng.integers(1, 50)))]
            assert a.lookup_batch(bk) == b.lookup_batch(bk), step
        elif r < 0.38:    # new keys until the table is full: removed / cleaned keys are reclaimed, their rows change hands
            nk = ("N", "n%d" % step)
            x = addrs[int(rng.integers(len(addrs) - 1))]
            ra = rb = None
            try:
                a.update(nk[0], nk[1], x)
            except gp.ObjectPlacementError as e:
                ra = e.rc
            try:
                b.update(nk[0], nk[1], x)
            except gp.ObjectPlacementError as e:
                rb = e.rc
            assert ra == rb, step
            if ra is None:
                keys.append(nk)
        elif r < 0.70:
            me = addrs[int(rng.integers(len(addrs) - 1))]
            # the non-blocking twin first (rio_op_try_*: the shadow or RIO_GP_EAGAIN, never the device): when it answers, it
            # answers what the provider WITHOUT a shadow gets from the device
            before = a.device_round_trips()
            ok, taddr, tflag = a.try_get_or_create_placement(ty, oid, me)
            assert a.device_round_trips() == before and b.try_get_or_create_placement(ty, oid, me)[0] is False
            want = b.get_or_create_placement(ty, oid, me)
            if ok:
                tries["request"] += 1
                assert (taddr, tflag) == want and tflag in (gp.FLAG_LOCAL, gp.FLAG_REDIRECT), (step, ty, oid, me)
            assert a.get_or_create_placement(ty, oid, me) == want, (step, ty, oid, me)
        else:
            before = a.device_round_trips()
            ok, taddr = a.try_lookup(ty, oid)
            assert a.device_round_trips() == before
            want = b.lookup(ty, oid)
            if ok:
                tries["lookup"] += 1
                assert taddr == want, (step, ty, oid)
            assert a.lookup(ty, oid) == want, (step, ty, oid)
    for k in keys:
        assert a.lookup(*k) == b.lookup(*k)
    assert len(a) == len(b) and sorted(a.snapshot()) == sorted(b.snapshot())
    (_, ra), (_, rb) = a.device_round_trips(), b.device_round_trips()
    assert ra < rb, (ra, rb)            # (a stream of mostly mutations: every one of them is the device's)
    # ... and a read phase: every key has been looked up once by now, so lookups and sticky requests stay on the host
    live = [addrs[k] for k in range(len(addrs) - 1) if alive[k]]
    calls = 0
    for _ in range(2):
        for k in keys:
            la, lb = a.lookup(*k), b.lookup(*k)
            assert la == lb
            calls += 1
            if live and la is not None:     # (a key that is placed: its request needs no new row of the small table)
                assert a.get_or_create_placement(k[0], k[1], live[0]) == b.get_or_create_placement(k[0], k[1], live[0])
                calls += 1
    (_, ra2), (_, rb2) = a.device_round_trips(), b.device_round_trips()
    assert rb2 - rb >= calls - 2 * len(keys) and ra2 - ra < 0.6 * (rb2 - rb), (ra2 - ra, rb2 - rb, calls)
    # the try calls did answer (a dead entry point would pass every comparison above), and now that every key has been looked
    # up they answer every lookup of a known key — without a round trip
    assert tries["lookup"] > 0 and tries["request"] > 0, tries
    before = a.device_round_trips()
    for k in keys:                       # (one more pass of blocking lookups: whatever a late clean invalidated is back)
        a.lookup(*k)
    before = a.device_round_trips()
    for k in keys:
        ok, addr = a.try_lookup(*k)
        assert ok and addr == b.lookup(*k), k
    assert a.try_lookup("Never", "seen") == (True, None)          # a key nobody has interned: Ok(None), no device work
    assert b.try_lookup("Never", "seen") == (True, None)
    placed = [k for k in keys if b.lookup(*k) is not None]
    assert placed and all(b.try_lookup(*k)[0] is False for k in placed)   # without a shadow every interned key is EAGAIN
    assert a.device_round_trips() == before
    a.close(); b.close()


def test_snapshot_round_trip_in_the_reference_schema(gp, tmp_path):
    """dump -> a SQLite file in the reference's layout -> readable by the REFERENCE's own SELECT
    (sqlite.rs:87-93, text carried by the golden fixture) -> load into a fresh table; and the other way round:
    a DB built with the reference's DDL + upsert text warm-starts the GPU table."""
    import json
    import re
    import sqlite3
    import snapshot
    sql = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sql_backend_golden.json")))["sql"]
    bind = lambda stmt: re.sub(r"\$(\d)", r"?\1", stmt)
    a = gp.GpuObjectPlacement(max_objects=4096, max_nodes=16)
    keys = [("Room", str(i)) for i in range(300)] + [("a.b", "c"), ("Metric", "x:y")]
    addrs = ["10.0.0.%d:5000" % (i % 7) for i in range(len(keys))]
    a.update_batch(keys, addrs)
    a.remove("Room", "7")
    a.clean_server("10.0.0.3:5000")
    want = {k: a.lookup(*k) for k in keys}
    path = str(tmp_path / "placement.sqlite3")
    n = snapshot.dump_sqlite(a, path)
    assert n == sum(v is not None for v in want.values()) == len(a)
    db = sqlite3.connect(path)
    for k, v in want.items():   # the reference's lookup statement against OUR file
        row = db.execute(bind(sql["select"]), k).fetchone()
        assert (row[0] if row else None) == v
    db.close()
    b = gp.GpuObjectPlacement(max_objects=4096, max_nodes=16)
    assert snapshot.load_sqlite(b, path) == n
    assert {k: b.lookup(*k) for k in keys} == want
    # a database written by the reference's own DDL + upsert text
    path2 = str(tmp_path / "reference.sqlite3")
    db = sqlite3.connect(path2)
    db.executescript(sql["ddl"])
    for k, v in zip(keys, addrs):
        db.execute(bind(sql["upsert"]), (k[0], k[1], v))
    db.execute(bind(sql["upsert"]), ("Room", "1", "10.9.9.9:1"))   # upsert overwrites (sqlite.rs:149-193)
    db.commit(); db.close()
    c = gp.GpuObjectPlacement(max_objects=4096, max_nodes=16)
    assert snapshot.load_sqlite(c, path2) == len(keys)
    assert c.lookup("Room", "1") == "10.9.9.9:1" and c.lookup("a.b", "c") == addrs[300] and len(c) == len(keys)
    # the Postgres twin (migrations/0001-postgres-init.sql): a psql script with one COPY block, and back
    script = str(tmp_path / "placement.pg.sql")
    assert snapshot.dump_postgres_script(a, script) == n
    d = gp.GpuObjectPlacement(max_objects=4096, max_nodes=16)
    assert snapshot.load_postgres_script(d, script) == n
    assert {k: d.lookup(*k) for k in keys} == want and len(d) == n
    for x in (a, b, c, d):
        x.close()


def test_concurrent_single_object_calls_share_round_trips(gp, tmp_path):
    """The trait is called from one task per connection (server.rs:292-304): T threads issue single-object lookups and
    get_or_create_placement calls against ONE provider through the C ABI (examples/c_host_threads.c, plain C + pthreads).
    Every answer must be right, and concurrent callers must not be slower than a lone one (the combining front-end lets
    them share a device round trip)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "c_host_threads"
    libdir = os.path.dirname(gp.LIB_PATH)
    subprocess.run(["gcc", "-O2", "-std=c99", "-pthread", "-I", os.path.join(root, "include"),
                    os.path.join(root, "examples", "c_host_threads.c"), "-o", str(exe), "-L", libdir, "-lrio_gp",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe), "5000", "400", "8"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    # 2 providers (host shadow | every call on the device) x 3 calls x threads {1, 4}; 8 is not in the ladder — and, with the
    # shadow, the two hand-off entries (every call through a pool of blocking threads | rio_op_try_* inline, the pool on EAGAIN)
    # x 3 calls at one thread
    assert len(rows) == 18 and all(x["wrong"] == 0 for x in rows)
    pick = lambda prov, call, t, entry="direct": [x for x in rows if x["provider"] == prov and x["call"] == call and
                                                  x["threads"] == t and x["entry"] == entry][0]
    for call in ("lookup", "get_or_create_placement"):   # known, placed keys: every try call is answered, none reaches the device
        x, y = pick("shadow", call, 1, "try"), pick("shadow", call, 1, "pool")
        assert x["try_calls"] == x["calls"] and x["try_answered"] == x["try_calls"] and x["requests_on_device"] == 0, x
        assert x["calls_per_s"] > 2 * y["calls_per_s"], (x, y)      # (the hand-off costs several times the hit it carries)
    x = pick("shadow", "churn", 1, "try")                # a tenth of the calls are first touches: those take the hand-off
    assert 0 < x["try_answered"] < x["try_calls"] and x["requests_on_device"] > 0, x
    for call in ("lookup", "get_or_create_placement", "churn"):
        one, four = pick("device", call, 1), pick("device", call, 4)
        assert four["calls_per_s"] > 0.5 * one["calls_per_s"], (call, one, four)   # shares round trips; the bound only guards against a convoy
        assert four["device_round_trips"] < four["requests_on_device"]          # ... and did share some
    for call in ("lookup", "get_or_create_placement"):   # known, placed keys: the shadow answers, the device is not asked
        for t in (1, 4):
            x = pick("shadow", call, t)
            assert x["requests_on_device"] == 0 and x["calls_per_s"] > 3 * pick("device", call, t)["calls_per_s"], x
    assert pick("shadow", "churn", 4)["requests_on_device"] > 0                 # first touches are the device's



# ---- rows that are not objects (rio_op_tick must not resurrect removed / never-inserted keys) -------

def test_tick_does_not_resurrect_removed_or_unknown_rows(gp):
    p = gp.GpuObjectPlacement(max_objects=1 << 12, max_nodes=8)
    p.set_member("10.0.0.1:1", True)
    p.set_member("10.0.0.2:1", True)
    for k in range(40):
        p.update("T", str(k), "10.0.0.%d:1" % (1 + k % 2))
    for k in range(0, 40, 4):
        p.remove("T", str(k))                                   # 10 removed
    p.update("T", "1", None)                                    # deleted through update(None) (local.rs:36-37)
    p.set_object_load("T", "load-only", 7)                      # a key with a load but never placed: not an object
    assert len(p) == 29
    st = p.tick()
    assert st["n_objects"] == 29 and st["kept"] == 29 and st["spilled"] == 0 and st["unplaced"] == 0
    assert len(p) == 29                                         # NOT max_objects
    for k in range(0, 40, 4):
        assert p.lookup("T", str(k)) is None                    # removed stays removed
    assert p.lookup("T", "1") is None and p.lookup("T", "load-only") is None
    assert sorted(x[1] for x in p.snapshot()) == sorted(str(k) for k in range(40) if k % 4 and k != 1)
    # a server dies: its objects are evicted AND re-placed by the tick (eager form of service.rs:227-252); the removed
    # rows still do not take part
    p.set_member("10.0.0.1:1", False)
    on1 = [k for k in range(40) if k % 4 and k != 1 and k % 2 == 0]
    st = p.tick()
    assert st["evicted"] == len(on1) and st["n_objects"] == 29 and st["unplaced"] == 0
    for k in on1:
        assert p.lookup("T", str(k)) == "10.0.0.2:1"
    assert len(p) == 29
    # explicit clean_server drops the entries (local.rs:51-58): they are not re-placed by a tick, they come back lazily
    p.clean_server("10.0.0.2:1")
    assert len(p) == 0
    st = p.tick()
    assert st["n_objects"] == 0 and len(p) == 0
    got, flag = p.get_or_create_placement("T", "2", "10.0.0.2:1")
    assert got == "10.0.0.2:1" and flag == gp.FLAG_PLACED and len(p) == 1
    p.close()


def test_row_lifecycle_column_on_the_dense_layer(gp, oracle):
    """RIO_GP_CFG_ROW_LIFECYCLE: the CRUD calls maintain which rows are objects (affinity column), and a tick over such a
    table equals the oracle's tick over the same columns — inactive rows included."""
    rng = np.random.default_rng(5)
    n, m = 50_000, 24
    g = gp.GpuPlacement(n, m, flags=gp.CFG_ROW_LIFECYCLE)
    cap = np.full(m, 0xFFFFFFFFFFFFFFFF, np.uint64)
    g.set_nodes(cap, np.ones(m, np.uint8))
    g.set_objects(n)                                             # load 1, every row a non-object
    load, aff = g.get_objects()
    assert np.all(aff == gp.AFF_INACTIVE) and np.all(load == 1)
    idx = rng.choice(n, 20_000, replace=False).astype(np.uint32)
    node = rng.integers(0, m, len(idx)).astype(np.uint32)
    g.update_batch(idx, node)                                    # objects now, affinity = their node
    g.update_batch(idx[:10], node[:10])                          # micro-batch path
    g.remove_batch(idx[:3000])                                   # not objects any more
    g.update_batch(idx[3000:4000], np.full(1000, gp.NONE, np.uint32))   # update(None) deletes
    ev = g.clean_servers([0, 1])                                 # dropped by clean_server
    req_rows = np.setdiff1d(np.arange(n, dtype=np.uint32), idx)[:5000]
    reqs = rng.integers(2, m, len(req_rows)).astype(np.uint32)
    g.place_pending(req_rows, reqs)                              # first touch: objects, home = the requester
    g.place_pending(req_rows[:100], reqs[:100])                  # micro-batch: sticky hits, nothing changes
    assign = g.get_assign()
    load, aff = g.get_objects()
    live = np.zeros(n, bool)
    live[idx[4000:]] = True
    live[idx[4000:][np.isin(node[4000:], [0, 1])]] = False
    live[req_rows] = True
    assert ev == int(np.isin(node[4000:], [0, 1]).sum())
    assert np.array_equal(aff != gp.AFF_INACTIVE, live)
    assert np.array_equal(assign != gp.NONE, live)
    assert np.array_equal(aff[req_rows], reqs) and np.array_equal(aff[idx[4000:]][live[idx[4000:]]], node[4000:][live[idx[4000:]]])
    alive = np.ones(m, np.uint8)
    alive[[5, 9]] = 0
    g.set_alive_all(alive)
    want, used, ost = oracle.tick(assign, load, aff, cap, alive, 2)
    st = g.tick()
    assert st == ost and st["n_objects"] == int(live.sum())
    assert np.array_equal(g.get_assign(), want) and np.array_equal(g.get_nodes()[2], used)
    g.close()


# ---- first requests after a server joins, from many threads at once ------------------------------

def test_concurrent_calls_introducing_new_addresses(gp):
    """16 threads, every call introduces or reuses a server address nobody has pushed to the device yet (ctypes releases
    the GIL in the C call, so the calls really overlap): zero non-OK returns, every answer right.  Before the fix the
    thread that interned a new address released the table lock before the node table went to the device, another
    thread's request carrying the new id could get there first, and the WHOLE combined batch failed with EINVAL."""
    import threading
    p = gp.GpuObjectPlacement(max_objects=1 << 16, max_nodes=4096)
    errors, wrong = [], []

    def worker(t):
        mine = p.clone()
        try:
            for k in range(120):
                a = "10.1.%d.%d:7000" % ((t + k) % 200, k % 7)
                b = "10.2.%d.1:7000" % ((t * 3 + k) % 250)
                key = "n%d_%d" % (t, k)
                got, flag = mine.get_or_create_placement("New", key, a)
                if got != a or flag != gp.FLAG_PLACED:
                    wrong.append((key, a, got, flag))
                mine.update("Upd", key, b)
                if mine.lookup("Upd", key) != b:
                    wrong.append((key, b))
        except Exception as e:  # any ObjectPlacementError is a failure of the test
            errors.append(repr(e))
        finally:
            mine.close()

    th = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:3]
    assert not wrong, wrong[:3]
    assert len(p) == 2 * 16 * 120
    p.close()


def test_keys_are_reclaimed_when_the_table_runs_full(gp):
    """interned rows were never reclaimed — under object-id churn a long-running server hit 'object table full'
    for good.  Now removed / deleted / cleaned keys give their rows back when the table runs full, and a table full of
    LIVE objects fails with its own message while everything else keeps working."""
    p = gp.GpuObjectPlacement(max_objects=256, max_nodes=8)
    p.set_member("h:1", True)
    for k in range(2000):                                      # 2 000 distinct keys through 256 rows
        key = "c%d" % k
        assert p.get_or_create_placement("Churn", key, "h:1")[0] == "h:1"
        if k % 3 == 0:
            p.remove("Churn", key)
        elif k % 3 == 1:
            p.update("Churn", key, None)
        else:
            p.clean_server("h:1")
        assert p.lookup("Churn", key) is None
    assert len(p) == 0
    for k in range(256):
        p.update("Live", str(k), "h:1")
    assert len(p) == 256
    with pytest.raises(gp.ObjectPlacementError) as e:
        p.update("Live", "one-too-many", "h:1")
    assert e.value.kind == "Unknown" and "object table full" in e.value.text
    assert p.lookup("Live", "17") == "h:1"                      # the table still answers
    p.remove("Live", "17")
    p.update("Live", "one-too-many", "h:1")                     # and a freed row is found again
    assert p.lookup("Live", "one-too-many") == "h:1" and len(p) == 256
    p.close()


def test_addresses_of_any_length_are_never_truncated(gp):
    """local.rs:42-49 returns Option<String> of any length (round-2 verdict: the C layer truncated to out_cap).  A 600-byte
    address comes back whole through the adapter's retry, and the raw call reports RIO_GP_ERANGE plus the needed length."""
    import ctypes as C
    p = gp.GpuObjectPlacement(max_objects=64, max_nodes=8)
    long_addr = "host-" + "x" * 590 + ":5000"
    assert len(long_addr) == 600
    p.update("Long", "1", long_addr)
    assert p.lookup("Long", "1") == long_addr                                   # 512-byte first try, then the exact size
    L = gp._oplib()
    buf, found = C.create_string_buffer(64), C.c_int(0)
    rc = L.rio_op_lookup(p._h, b"Long", b"1", buf, 64, C.byref(found))
    assert rc == gp.ERANGE and found.value == 1 and buf.value == b""
    assert L.rio_op_last_address_len(p._h) == 600
    buf = C.create_string_buffer(601)
    assert L.rio_op_lookup(p._h, b"Long", b"1", buf, 601, C.byref(found)) == gp.OK and buf.value.decode() == long_addr
    # exactly fitting / one short
    assert L.rio_op_lookup(p._h, b"Long", b"1", C.create_string_buffer(600), 600, C.byref(found)) == gp.ERANGE
    # the policy call: the decision is made and flagged even when the address does not fit the caller's buffer
    p.set_member(long_addr, True)
    addr, flag = p.get_or_create_placement("Long", "2", long_addr, _cap=32)
    assert addr == long_addr and flag == gp.FLAG_PLACED
    addr, flag = p.get_or_create_placement("Long", "2", long_addr)
    assert addr == long_addr and flag == gp.FLAG_LOCAL
    assert p.lookup("Long", "nobody") is None and L.rio_op_last_address_len(p._h) == 0
    p.close()


def test_a_load_set_ahead_of_first_use_survives_key_reclaim(gp):
    """Round-2 advisor finding: reclaim() recycled (and reset to load 1) the row of a key whose load had been set before its
    first update / request.  Such a key keeps its row until it is used; afterwards it is reclaimed like any other key."""
    p = gp.GpuObjectPlacement(max_objects=64, max_nodes=4)
    p.set_member("a:1", True, capacity=10)
    p.set_member("b:1", True, capacity=1000)
    p.set_object_load("Heavy", "h", 7)                      # not an object yet
    for k in range(300):                                    # key churn through the other 63 rows: several reclaims
        key = "c%d" % k
        assert p.get_or_create_placement("Churn", key, "b:1")[0] == "b:1"
        p.remove("Churn", key)
    assert p.get_or_create_placement("Heavy", "h", "a:1") == ("a:1", gp.FLAG_PLACED)
    st = p.tick()                                           # the row still carries load 7
    assert st["load_kept"] == 7
    # 'a:1' holds 7 of 10: a second object of load 7 does not fit and is water-filled onto b:1
    p.set_object_load("Heavy", "h2", 7)
    assert p.get_or_create_placement("Heavy", "h2", "a:1") == ("b:1", gp.FLAG_SPILLED)
    p.remove("Heavy", "h")                                  # used now: reclaimable like any other key
    for k in range(300):
        key = "d%d" % k
        assert p.get_or_create_placement("Churn", key, "b:1")[0] == "b:1"
        p.remove("Churn", key)
    assert p.lookup("Heavy", "h") is None and p.lookup("Heavy", "h2") == "b:1"
    p.close()
