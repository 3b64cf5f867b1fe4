"""Durable twin of the HBM placement table: dump to / load from a SQLite file that
uses the reference's on-disk layout, so a GPU-backed server can warm-start from — or write back to —
the database a `SqliteObjectPlacement` deployment already has.

Layout (rio-rs/src/object_placement/migrations/0001-sqlite-init.sql:1-9, written by the upsert of
rio-rs/src/object_placement/sqlite.rs:68-85): table `object_placement(struct_name TEXT NOT NULL,
object_id TEXT NOT NULL, server_address TEXT NULL)`, primary key `(struct_name, object_id)`, an index on
`server_address`.  Rows whose address is NULL are skipped on load: the reference's own `lookup` cannot
decode them either (sqlite.rs:99) and no in-tree caller writes them (service.rs:244).

Not on the hot path: one D2H copy of the assignment column and a host loop over the interned keys.
"""
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
