#!/usr/bin/env python3
"""Generate tests/golden/sql_backend_golden.json from the REFERENCE's own SQL.

Runs only in the build container (needs /root/reference); the JSON it writes is what
travels to the GPU box.  It extracts, verbatim, the SQL statements of the reference's
SqliteObjectPlacement (rio-rs/src/object_placement/sqlite.rs:72-78 upsert, :87-93 select,
:102-106 delete-by-server, :115-119 delete-by-key) and its DDL
(rio-rs/src/object_placement/migrations/0001-sqlite-init.sql), executes seeded random
sequences of trait calls against a real SQLite engine (python stdlib sqlite3) and records
every lookup result.  tests/test_oracle_golden.py replays the same sequences through the
oracle (and, on the GPU, through the HIP path) and demands identical answers.

Calls with server_address = None are excluded: there the backends genuinely diverge
(Local deletes, local.rs:36-37; SQL stores NULL and a later lookup fails to decode it,
sqlite.rs:81,99) and no in-tree caller passes None (service.rs:244).
"""
import json
import os
import random
import re
import sqlite3
import sys

REF = "/root/reference/rio-rs/src/object_placement"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sql_backend_golden.json")


def reference_sql():
    src = open(os.path.join(REF, "sqlite.rs")).read()
    stmts = re.findall(r'r#"(.*?)"#', src, flags=re.S)
    stmts = [" ".join(s.split()) for s in stmts]
    upsert = next(s for s in stmts if s.startswith("INSERT INTO"))
    select = next(s for s in stmts if s.startswith("SELECT server_address"))
    del_server = next(s for s in stmts if s.startswith("DELETE") and "server_address = $1" in s)
    del_key = next(s for s in stmts if s.startswith("DELETE") and "struct_name = $1" in s)
    mig_dir = os.path.join(REF, "migrations")
    ddl = ""
    for f in sorted(os.listdir(mig_dir)):
        if "sqlite" in f:
            ddl += open(os.path.join(mig_dir, f)).read() + "\n"
    return dict(upsert=upsert, select=select, del_server=del_server, del_key=del_key, ddl=ddl)


class RefSqlite:
    """The trait methods of sqlite.rs:68-127 executed with the reference's SQL text."""

    def __init__(self, sql):
        self.sql = sql
        self.db = sqlite3.connect(":memory:")
        self.db.executescript(sql["ddl"])  # prepare(), sqlite.rs:58-66

    @staticmethod
    def _bind(stmt, *args):  # sqlx binds $1,$2,$3 positionally
        return re.sub(r"\$(\d)", r"?\1", stmt), args

    def update(self, ty, oid, addr):
        self.db.execute(*self._bind(self.sql["upsert"], ty, oid, addr))

    def lookup(self, ty, oid):
        row = self.db.execute(*self._bind(self.sql["select"], ty, oid)).fetchone()
        return None if row is None else row[0]

    def clean_server(self, addr):
        self.db.execute(*self._bind(self.sql["del_server"], addr))

    def remove(self, ty, oid):
        self.db.execute(*self._bind(self.sql["del_key"], ty, oid))


def gen_sequence(rng, n_ops, n_types, n_ids, n_addrs):
    types = ["T%d" % t for t in range(n_types)]
    addrs = ["10.0.%d.%d:%d" % (a >> 8, a & 255, 5000 + (a % 3)) for a in range(n_addrs)]
    ops = []
    for _ in range(n_ops):
        p = rng.random()
        ty, oid = rng.choice(types), str(rng.randrange(n_ids))
        if p < 0.40:
            ops.append(["update", ty, oid, rng.choice(addrs)])
        elif p < 0.75:
            ops.append(["lookup", ty, oid])
        elif p < 0.88:
            ops.append(["remove", ty, oid])
        else:
            ops.append(["clean_server", rng.choice(addrs)])
    # finish with a full sweep of lookups so the final table is pinned entirely
    for ty in types:
        for i in range(n_ids):
            ops.append(["lookup", ty, str(i)])
    return ops


def main():
    sql = reference_sql()
    rng = random.Random(0x52494F)
    cases = []
    for (n_ops, n_types, n_ids, n_addrs) in [(40, 1, 4, 2), (200, 2, 12, 3), (600, 3, 40, 5), (1500, 2, 200, 8),
                                             (3000, 4, 64, 16)]:
        ops = gen_sequence(rng, n_ops, n_types, n_ids, n_addrs)
        be = RefSqlite(sql)
        expected = []
        for op in ops:
            if op[0] == "lookup":
                expected.append(be.lookup(op[1], op[2]))
            else:
                getattr(be, op[0])(*op[1:])
        cases.append(dict(ops=ops, expected_lookups=expected))
    doc = dict(
        source="reference SQL text of rio-rs/src/object_placement/sqlite.rs + migrations/0001-sqlite-init.sql "
               "executed by python sqlite3 %s" % sqlite3.sqlite_version,
        sql={k: v for k, v in sql.items()},
        cases=cases)
    with open(OUT, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", sum(len(c["ops"]) for c in cases), "ops")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not mounted; golden file is committed, nothing to do")
    main()
