#!/usr/bin/env python3
"""Soak of the chained quiet ticks under an uneven foreign load: handle A (config 3, 10 M rows) runs quiet committed ticks for
`seconds`; a second thread drives handle B (1 M rows) with update / remove / place_pending batches and whole-table solves the
whole time (other kernels taking CUs, LDS and wave slots at random moments).  Checked: no call fails (a chained wait that gave up
would fail rio_gp_tick_wait), every tick of A keeps every row, A's table and `used` are what they were, B's table equals the
oracle's after its last batch.  Usage: soak_chain.py [seconds=20]"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth, pyoracle
pyoracle.build()
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
a = rio_gp.LabPlacement(n, m)
a.set_nodes(cfg["cap"], cfg["alive"]); a.set_objects(n, cfg["load"], cfg["aff"]); a.set_assign(cfg["cur"])
st0 = a.tick()
table0, used0 = a.get_assign(), a.get_nodes()[2]
stop, errs, out = threading.Event(), [], {}

def foreign():
    try:
        cb = synth.config("c3", n_override=1_000_000)
        nb, mb = cb["n"], cb["m"]
        b = rio_gp.GpuPlacement(nb, mb)
        b.set_nodes(cb["cap"], cb["alive"]); b.set_objects(nb, cb["load"], cb["aff"])
        ref = np.full(nb, 0xFFFFFFFF, np.uint32); b.set_assign(ref)
        rng = np.random.default_rng(5); k = 0
        while not stop.is_set():
            idx = rng.choice(nb, int(rng.choice([300, 5000, 300_000])), replace=False).astype(np.uint32)
            op = k % 4
            if op == 0:
                node = rng.integers(0, mb, idx.size).astype(np.uint32); b.update_batch(idx, node); ref[idx] = node
            elif op == 1:
                b.remove_batch(idx); ref[idx] = 0xFFFFFFFF
            elif op == 2:
                b.lookup_batch(idx)
            else:
                b.solve()
            k += 1
        out["foreign_calls"] = k
        out["foreign_table_equal"] = bool(np.array_equal(b.get_assign(), ref))
        b.close()
    except Exception as e:
        errs.append("foreign: " + repr(e))

t = threading.Thread(target=foreign); t.start()
t0 = time.time(); ticks = 0; bad = 0
try:
    while time.time() - t0 < seconds:
        for _ in range(64):
            a.tick_async()
        for s in a.tick_wait():
            ticks += 1
            bad += int(s["kept"] != st0["kept"] + st0["claimed"] + st0["spilled"] or s["slow_path"] != 0)
except Exception as e:
    errs.append("ticks: " + repr(e))
stop.set(); t.join()
out.update({"seconds": round(time.time() - t0, 1), "ticks": ticks, "ticks_not_all_kept": bad, "chained_scans": a.chained_scans(),
            "us_per_tick_with_the_foreign_load": (time.time() - t0) / max(ticks, 1) * 1e6,
            "table_equal": bool(np.array_equal(a.get_assign(), table0)), "used_equal": bool(np.array_equal(a.get_nodes()[2], used0)),
            "errors": errs})
print(json.dumps(out))
a.close()
sys.exit(1 if errs or bad or not out["table_equal"] else 0)
