#!/usr/bin/env python3
"""Launch shapes of the plain 3-read/1-write streaming probe next to k_scan (config 3): wave-contiguous and block-tiled with
one and two 1 024-thread workgroups per CU, grid-stride with 2 048 / 8 192 workgroups.  Measured: two workgroups per CU buy
nothing (24.7 vs 24.4 us); what separates k_scan's shape from the grid-stride probe (22.4-22.7 us) is the access pattern
(4 096 separate streams instead of one moving window), which the ordered-prefix design needs."""
import sys
sys.path[:0]=["rio-rs_amd","oracle"]
import rio_gp, synth, numpy as np
cfg=synth.config("c3"); n=cfg["n"]
g=rio_gp.LabPlacement(n,cfg["m"]); g.set_nodes(cfg["cap"],cfg["alive"]); g.set_objects(n,cfg["load"],cfg["aff"])
for rnd in range(3):
    for mode,name in ((2,"wavecontig 256x1024"),(10,"wavecontig 512x1024"),(1,"blocktile 256x1024"),(11,"blocktile 512x1024"),(0,"gridstride 2048x256"),(5,"gridstride 8192x256")):
        ms=g.stream_probe(mode,30); print(rnd,name,"%.2f us %.0f GB/s"%(ms*1e3,16*n/ms/1e6))
sc=[g.solve_profiled()[0] for _ in range(60)][10:]; print("k_scan %.2f us"%(np.median(sc)*1e3))
