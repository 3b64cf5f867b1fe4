#!/usr/bin/env python3
"""Same-run A/B of the whole-table fix-up's routes on the two tables in which capacity binds (config 3 contended / skew):
   find+fill (k_cut_find, k_fill<APPLY,FILL>, k_fill<FILL>) | apply (k_cut_apply without packing, k_cut_settle, k_fill<FILL> x rounds
   over the table) | apply+pack (the rounds over the packed rows) | find+fill with packing in round 0.  Lab build.
   Usage: binding_route_ab.py [reps]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
routes = {
    "library default": dict(cut_pack="auto", cut_apply="auto"),
    "find + fill": dict(cut_pack="never", cut_apply="never"),
    "find + fill, round 0 packs": dict(cut_pack="always", cut_apply="never"),
    "apply (no packing) + rounds over the table": dict(cut_pack="never", cut_apply="always"),
    "apply + pack, rounds over the packed rows": dict(cut_pack="always", cut_apply="always"),
}
out = {"reps": reps, "n": n, "m": m, "tables": {}}
for which in ("contended", "skew"):
    if which == "contended":
        cap = (cfg["cap"].astype(np.float64) * 0.72).astype(np.uint64)
        aff = cfg["aff"]
    else:
        cap = cfg["cap"]
        aff = synth.skew_affinity(n, m)
    rec = {}
    ref = None
    for rnd in range(2):  # two alternations
        for name, kw in routes.items():
            g = rio_gp.LabPlacement(n, m)
            g.set_compact("never", **kw)
            g.set_nodes(cap, cfg["alive"])
            g.set_objects(n, cfg["load"], aff)
            for _ in range(3):
                st = g.solve()
            g.sync(); ts = []
            for _ in range(reps):
                t0 = time.perf_counter(); st = g.solve(); ts.append(time.perf_counter() - t0)
            nxt = g.read_next() if hasattr(g, "read_next") else None
            key = (st["claimed"], st["spilled"], st["unplaced"], st["load_spilled"], st["load_unplaced"])
            if ref is None: ref = key
            rec.setdefault(name, []).append({"us_per_solve_median": float(np.median(ts) * 1e6), "us_p10": float(np.percentile(ts, 10) * 1e6),
                                             "equal_counters": key == ref})
            g.close()
    out["tables"][which] = rec
print(json.dumps(out, indent=1))
