"""Durable twin of the HBM placement table: dump to / load from a SQLite file that
uses the reference's on-disk layout, so a GPU-backed server can warm-start from — or write back to —
the database a `SqliteObjectPlacement` deployment already has.

Layout (rio-rs/src/object_placement/migrations/0001-sqlite-init.sql:1-9, written by the upsert of
rio-rs/src/object_placement/sqlite.rs:68-85): table `object_placement(struct_name TEXT NOT NULL,
object_id TEXT NOT NULL, server_address TEXT NULL)`, primary key `(struct_name, object_id)`, an index on
`server_address`.  Rows whose address is NULL are skipped on load: the reference's own `lookup` cannot
decode them either (sqlite.rs:99) and no in-tree caller writes them (service.rs:244).

The Postgres twin (migrations/0001-postgres-init.sql) is at the end of the file.

Not on the hot path: one D2H copy of the assignment column and a host loop over the interned keys.
"""
import re
import sqlite3

SCHEMA = (
    "CREATE TABLE IF NOT EXISTS object_placement ("
    " struct_name TEXT NOT NULL, object_id TEXT NOT NULL, server_address TEXT NULL,"
    " PRIMARY KEY (struct_name, object_id));"
    "CREATE INDEX IF NOT EXISTS idx_object_placement_server_address ON object_placement(server_address);"
)


def dump_sqlite(placement, path, replace=True):
    """Write every placed entry of a rio_gp.GpuObjectPlacement into `path`; returns the row count."""
    rows = placement.snapshot()
    db = sqlite3.connect(path)
    try:
        db.executescript(SCHEMA)
        if replace:
            db.execute("DELETE FROM object_placement")
        db.executemany(
            "INSERT INTO object_placement(struct_name, object_id, server_address) VALUES (?, ?, ?) "
            "ON CONFLICT(struct_name, object_id) DO UPDATE SET server_address = excluded.server_address", rows)
        db.commit()
    finally:
        db.close()
    return len(rows)


def load_sqlite(placement, path, batch=65536):
    """Upsert every row of the file's object_placement table into the placement (one kernel launch per batch)."""
    db = sqlite3.connect(path)
    n = 0
    try:
        cur = db.execute("SELECT struct_name, object_id, server_address FROM object_placement "
                         "WHERE server_address IS NOT NULL ORDER BY struct_name, object_id")
        while True:
            rows = cur.fetchmany(batch)
            if not rows:
                break
            placement.update_batch([(r[0], r[1]) for r in rows], [r[2] for r in rows])
            n += len(rows)
    finally:
        db.close()
    return n


# ------------------------------------------------------------------------------------------------
# The Postgres twin (rio-rs/src/object_placement/migrations/0001-postgres-init.sql:1-9; upsert of
# rio-rs/src/object_placement/postgres.rs:74-85, lookup :89-98).  Same three columns, same key, same index.  Two transports:
#   * a DB-API 2 connection with the `format` paramstyle (psycopg2 / psycopg 3 — neither ships in this image, so the tests drive
#     these functions through a shim that hands the statements to SQLite; the statements are the reference's text with
#     $n turned into %s — the upsert's `DO UPDATE SET server_address = excluded.server_address` stands for postgres.rs:79's
#     `= $3`, which it equals: one bound value less);
#   * a script `psql -f` loads: the DDL + one `COPY object_placement (...) FROM stdin` block in COPY's text format, which is
#     also what `pg_dump --data-only --table object_placement` writes — load_postgres_script reads either.
# Postgres TEXT cannot hold a NUL byte: a key with one (ObjectId holds any Rust string) is refused here with ValueError, as
# the reference's own PostgresObjectPlacement would fail on it at bind time.
# ------------------------------------------------------------------------------------------------
PG_SCHEMA = (
    "CREATE TABLE IF NOT EXISTS object_placement\n"
    "(\n"
    "    struct_name     TEXT                NOT NULL,\n"
    "    object_id       TEXT                NOT NULL,\n"
    "    server_address  TEXT                NULL,\n"
    "\n"
    "    PRIMARY KEY (struct_name, object_id)\n"
    ");\n"
    "CREATE INDEX IF NOT EXISTS idx_object_placement_server_address on object_placement(server_address);\n"
)
PG_UPSERT = ("INSERT INTO object_placement(struct_name, object_id, server_address) VALUES (%s, %s, %s) "
             "ON CONFLICT(struct_name, object_id) DO UPDATE SET server_address=excluded.server_address")
PG_SELECT_ALL = ("SELECT struct_name, object_id, server_address FROM object_placement "
                 "WHERE server_address IS NOT NULL ORDER BY struct_name, object_id")

_COPY_HEAD = re.compile(r'copy\s+(?:"?\w+"?\.)?"?object_placement"?\s*(\(|from\b)')   # (not object_placement_anything_else)
_COPY_ESC = {"\\": "\\\\", "\t": "\\t", "\n": "\\n", "\r": "\\r", "\b": "\\b", "\f": "\\f", "\v": "\\v"}
_COPY_UNESC = {"\\": "\\", "t": "\t", "n": "\n", "r": "\r", "b": "\b", "f": "\f", "v": "\v"}


def _no_nul(rows):
    for r in rows:
        if any(x is not None and "\0" in x for x in r):
            raise ValueError("a Postgres TEXT value cannot hold a NUL byte: %r" % (r,))
    return rows


def copy_escape(value):
    """One field of COPY's text format (NULL is \\N; backslash, tab, newline, ... are backslash sequences)."""
    if value is None:
        return "\\N"
    return "".join(_COPY_ESC.get(ch, ch) for ch in value)


def copy_unescape(field):
    """Inverse of copy_escape for one field of COPY's text format.  \\ooo and \\xhh stand for BYTES of the server encoding
    (UTF-8 here), not code points: consecutive escaped bytes are collected and decoded together, so a character somebody
    wrote as \\303\\251 comes back as the one character it is (pg_dump itself never emits these escapes)."""
    if field == "\\N":
        return None
    out, raw, i, n = [], bytearray(), 0, len(field)

    def flush():
        if raw:
            out.append(bytes(raw).decode("utf-8"))   # an invalid sequence raises: a key is never silently changed
            raw.clear()
    while i < n:
        ch = field[i]
        if ch != "\\" or i + 1 == n:
            flush()
            out.append(ch)
            i += 1
            continue
        nx = field[i + 1]
        if nx in _COPY_UNESC:
            flush()
            out.append(_COPY_UNESC[nx])
            i += 2
        elif nx in "01234567":            # \ooo: one byte
            j = i + 1
            while j < n and j < i + 4 and field[j] in "01234567":
                j += 1
            raw.append(int(field[i + 1:j], 8) & 0xFF)
            i = j
        elif nx == "x" and i + 2 < n and field[i + 2] in "0123456789abcdefABCDEF":   # \xh, \xhh: one byte
            j = i + 2
            while j < n and j < i + 4 and field[j] in "0123456789abcdefABCDEF":
                j += 1
            raw.append(int(field[i + 2:j], 16))
            i = j
        else:                              # any other backslashed character stands for itself
            flush()
            out.append(nx)
            i += 2
    flush()
    return "".join(out)


def dump_postgres(placement, conn, replace=True, batch=65536):
    """Write every placed entry through a DB-API connection (psycopg-style %s parameters); returns the row count."""
    rows = _no_nul(placement.snapshot())
    cur = conn.cursor()
    for stmt in PG_SCHEMA.split(";\n"):
        if stmt.strip():
            cur.execute(stmt)
    if replace:
        cur.execute("DELETE FROM object_placement")
    for i in range(0, len(rows), batch):
        cur.executemany(PG_UPSERT, rows[i:i + batch])
    conn.commit()
    return len(rows)


def load_postgres(placement, conn, batch=65536):
    """Upsert every row of the connection's object_placement table into the placement (one kernel launch per batch)."""
    cur = conn.cursor()
    cur.execute(PG_SELECT_ALL)
    n = 0
    while True:
        rows = cur.fetchmany(batch)
        if not rows:
            break
        placement.update_batch([(r[0], r[1]) for r in rows], [r[2] for r in rows])
        n += len(rows)
    return n


def dump_postgres_script(placement, path, replace=True):
    """A script for `psql -f`: the reference's DDL, then the table's rows as one COPY block; returns the row count."""
    rows = _no_nul(placement.snapshot())
    with open(path, "w", encoding="utf-8", newline="\n") as f:
        f.write(PG_SCHEMA)
        f.write("BEGIN;\n")
        if replace:
            f.write("DELETE FROM object_placement;\n")
        f.write("COPY object_placement (struct_name, object_id, server_address) FROM stdin;\n")
        for r in rows:
            f.write("\t".join(copy_escape(x) for x in r))
            f.write("\n")
        f.write("\\.\n")
        f.write("COMMIT;\n")
    return len(rows)


def load_postgres_script(placement, path, batch=65536):
    """Read the COPY block(s) of object_placement out of a dump_postgres_script / pg_dump text file and upsert them."""
    n, cols, keys, addrs = 0, None, [], []

    def flush():
        nonlocal keys, addrs
        if keys:
            placement.update_batch(keys, addrs)
        keys, addrs = [], []

    with open(path, "r", encoding="utf-8", newline="\n") as f:
        for line in f:
            line = line.rstrip("\n")
            if cols is None:
                s = line.strip()
                low = s.lower()
                if _COPY_HEAD.match(low) and low.rstrip(";").rstrip().endswith("from stdin"):
                    inside = s[s.index("(") + 1:s.index(")")] if "(" in s else "struct_name, object_id, server_address"
                    cols = [c.strip().strip('"').lower() for c in inside.split(",")]
                    want = ("struct_name", "object_id", "server_address")
                    if sorted(cols) != sorted(want):
                        raise ValueError("unexpected columns in %r" % s)
                    cols = [cols.index(c) for c in want]
                continue
            if line == "\\.":
                cols = None
                continue
            fields = line.split("\t")
            if len(fields) != 3:
                raise ValueError("a COPY row of object_placement has three fields: %r" % line)
            ty, oid, addr = (copy_unescape(fields[c]) for c in cols)
            if addr is None:        # (see the module docstring: the reference's lookup cannot decode a NULL address either)
                continue
            keys.append((ty, oid)); addrs.append(addr)
            n += 1
            if len(keys) >= batch:
                flush()
    flush()
    return n
