import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
g = rio_gp.LabPlacement(n, m)
g.set_nodes(cfg["cap"], cfg["alive"])
g.set_objects(n, cfg["load"], cfg["aff"])
g.set_assign(synth.warm_assign(n, m))
which = sys.argv[1]
if which == "aa_compact":      # all alive, packing forced: k_scan<false, true, 1, 2>
    g.set_compact("always")
    for _ in range(12): g.tick()
elif which == "dead_nocompact":  # 10 % dead, no packing: k_scan<false, false, 2, 0>
    g.set_compact("never", cut_pack="never")
    for k in range(12):
        g.set_alive_all(synth.churn_mask(m, 2 + k)); g.tick()
elif which == "dead_compact":
    for k in range(12):
        g.set_alive_all(synth.churn_mask(m, 2 + k)); g.tick()
g.close()
