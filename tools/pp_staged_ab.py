#!/usr/bin/env python3
"""place_pending from host buffers, 300 .. 1 024 requests: the one-workgroup kernel against the three-launch form (lab knob:
the batch size from which the staged form is used), same run, alternating.  us per call, first touches and sticky hits."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
rng = np.random.default_rng(3)
out = {}
calls = 300
for rep in range(2):
    for staged_from in (1024, 256):
        rio_gp.lab_lib().rio_gp_debug_set_part_shift(14 | (staged_from << 8))
        for k in (256, 257, 300, 512, 1000, 1024):
            for what in ("first_touch", "sticky"):
                g = rio_gp.LabPlacement(n, m)
                g.set_nodes(cfg["cap"], cfg["alive"])
                g.set_objects(n, cfg["load"], cfg["aff"])
                if what == "sticky":
                    g.set_assign(synth.warm_assign(n, m)); g.tick()
                batches = [(rng.choice(n, k, replace=False).astype(np.uint32), rng.integers(0, m, k).astype(np.uint32)) for _ in range(calls + 4)]
                for b in batches[:4]:
                    g.place_pending(*b)
                t0 = time.perf_counter()
                for b in batches[4:]:
                    g.place_pending(*b)
                out.setdefault("staged_from_%d" % staged_from, {}).setdefault("%d %s" % (k, what), []).append(round((time.perf_counter() - t0) / calls * 1e6, 1))
                g.close()
rio_gp.lab_lib().rio_gp_debug_set_part_shift(14 | (1024 << 8))
print(json.dumps(out))
