"""The durable twins of the table on the host side, without a GPU: the Postgres transports of
rio-rs_amd/snapshot.py against the reference's own schema / statements (migrations/0001-postgres-init.sql:1-9,
postgres.rs:74-98), driven by a stand-in for the placement (the same two methods the GPU object has: snapshot(),
update_batch())."""
import os
import re
import sqlite3
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rio-rs_amd"))
import snapshot  # noqa: E402


class MapPlacement:
    """local.rs:12-40 as a dict: what snapshot() / update_batch() mean."""

    def __init__(self):
        self.m = {}

    def snapshot(self):
        return sorted((k[0], k[1], v) for k, v in self.m.items())

    def update_batch(self, keys, addrs):
        for k, a in zip(keys, addrs):
            if a is None:
                self.m.pop(tuple(k), None)
            else:
                self.m[tuple(k)] = a


class FormatCursor:
    """psycopg-style cursor (%s parameters) over SQLite: the statements are handed over unchanged but for the markers."""

    def __init__(self, db):
        self.c = db.cursor()

    def execute(self, stmt, params=()):
        self.c.execute(stmt.replace("%s", "?"), params)

    def executemany(self, stmt, rows):
        self.c.executemany(stmt.replace("%s", "?"), rows)

    def fetchmany(self, n):
        return self.c.fetchmany(n)


class FormatConn:
    def __init__(self, path):
        self.db = sqlite3.connect(path)

    def cursor(self):
        return FormatCursor(self.db)

    def commit(self):
        self.db.commit()


NASTY = [("Room", "1", "10.0.0.1:5000"), ("a.b", "c", "h:1"), ("tab\there", "nl\nthere", "h:2"), ("back\\slash", "\\N", "h:3"),
         ("cr\rx", "uni\u00e9\u4e16", "h:4"), ("", "", "h:5"), ("\\.", "x", "h:6"), ("bs\b", "ff\f\v", "h:7")]


def filled():
    p = MapPlacement()
    p.update_batch([(a, b) for a, b, _ in NASTY], [c for _, _, c in NASTY])
    return p


def test_copy_text_escaping_round_trips_every_character_class():
    for row in NASTY:
        for x in row + (None,):
            e = snapshot.copy_escape(x)
            assert "\t" not in e and "\n" not in e and "\r" not in e
            assert snapshot.copy_unescape(e) == x
    assert snapshot.copy_escape(None) == "\\N" and snapshot.copy_escape("\\N") == "\\\\N"
    assert snapshot.copy_unescape("a\\101\\x41\\q") == "aAAq"   # octal, hex, and "any other character stands for itself"
    # \ooo / \xhh are BYTES of the server encoding (round-4 advisor finding): two escaped bytes are one character
    assert snapshot.copy_unescape("caf\\303\\251") == "caf\u00e9" and snapshot.copy_unescape("\\xe2\\x82\\xac!") == "\u20ac!"
    with pytest.raises(UnicodeDecodeError):
        snapshot.copy_unescape("bad\\303x")                      # half a character: refused, never silently changed


def test_script_round_trip_and_pg_dump_shape(tmp_path):
    a = filled()
    path = str(tmp_path / "placement.sql")
    assert snapshot.dump_postgres_script(a, path) == len(NASTY)
    text = open(path, encoding="utf-8").read()
    assert text.startswith(snapshot.PG_SCHEMA)
    assert text.count("\n\\.\n") == 1
    b = MapPlacement()
    assert snapshot.load_postgres_script(b, path) == len(NASTY)
    assert b.m == a.m
    # what pg_dump --data-only writes: other statements around, the column list in another order, a NULL address
    dump = str(tmp_path / "pg_dump.sql")
    with open(dump, "w", encoding="utf-8", newline="\n") as f:
        f.write("SET client_encoding = 'UTF8';\nCOPY public.other (a) FROM stdin;\nx\n\\.\n\n"
                "COPY public.object_placement_history (struct_name, object_id, server_address) FROM stdin;\nOld\t1\th:0\n\\.\n\n"
                "COPY public.object_placement (server_address, struct_name, object_id) FROM stdin;\n"
                "h:9\tRoom\t77\n\\N\tRoom\t78\nh:1\\tx\tT\\\\\tq\n\\.\n\nSELECT 1;\n")
    c = MapPlacement()
    assert snapshot.load_postgres_script(c, dump) == 2
    assert c.m == {("Room", "77"): "h:9", ("T\\", "q"): "h:1\tx"}


def test_the_schema_is_the_reference_migration():
    """Where the reference is at hand (this container, not the GPU box): PG_SCHEMA is the migration's statements."""
    ref = "/root/reference/rio-rs/src/object_placement/migrations/0001-postgres-init.sql"
    if not os.path.exists(ref):
        pytest.skip("no reference tree here")
    norm = lambda t: re.sub(r"\s+", " ", t).strip().lower()
    assert norm(open(ref).read()) == norm(snapshot.PG_SCHEMA)


def test_dbapi_transport_with_the_reference_statements(tmp_path):
    a = filled()
    conn = FormatConn(str(tmp_path / "pg.sqlite3"))
    assert snapshot.dump_postgres(a, conn) == len(NASTY)
    # the reference's lookup statement (postgres.rs:89-98, $n markers) finds every row
    ref_select = "SELECT server_address FROM object_placement WHERE struct_name = $1 and object_id = $2"
    for ty, oid, addr in NASTY:
        row = conn.db.execute(re.sub(r"\$(\d)", r"?\1", ref_select), (ty, oid)).fetchone()
        assert row[0] == addr
    # rows the reference's upsert (postgres.rs:74-85) adds / overwrites arrive on load
    ref_upsert = ("INSERT INTO object_placement(struct_name, object_id, server_address) VALUES ($1, $2, $3) "
                  "ON CONFLICT(struct_name, object_id) DO UPDATE SET server_address=$3")
    conn.db.execute(re.sub(r"\$(\d)", r"?\1", ref_upsert), ("Room", "1", "moved:1"))
    conn.db.execute(re.sub(r"\$(\d)", r"?\1", ref_upsert), ("Room", "2", "new:1"))
    conn.db.commit()
    b = MapPlacement()
    assert snapshot.load_postgres(b, conn) == len(NASTY) + 1
    assert b.m[("Room", "1")] == "moved:1" and b.m[("Room", "2")] == "new:1" and len(b.m) == len(NASTY) + 1
    # replace=True rewrites the table
    assert snapshot.dump_postgres(a, conn) == len(NASTY)
    assert conn.db.execute("SELECT COUNT(*) FROM object_placement").fetchone()[0] == len(NASTY)


def test_a_key_with_a_nul_byte_is_refused(tmp_path):
    p = MapPlacement()
    p.update_batch([("T", "a\0b")], ["h:1"])
    with pytest.raises(ValueError):
        snapshot.dump_postgres_script(p, str(tmp_path / "x.sql"))
    with pytest.raises(ValueError):
        snapshot.dump_postgres(p, FormatConn(str(tmp_path / "x.sqlite3")))
