#!/usr/bin/env python3
"""How clustered are the rows a config-5 churn tick evicts?  Replays the stream on the CPU oracle (no GPU needed) and, for
a few ticks, reports the share of 4-row lane groups, 64-byte segments (16 rows), 128-byte lines (32 rows) and 256-row tiles of
a column that hold at least one pending row (unplaced, or on a node that is not alive), and the run lengths of consecutive
pending rows.  The answer decides whether the load / affinity columns of the pending rows are worth gathering (they are not:
DESIGN.md section 5).  Usage: cluster_stats.py [ticks=70]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import pyoracle, synth
pyoracle.build()
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 70
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
ref = synth.warm_assign(n, m)
ref, used, st = pyoracle.tick(ref, cfg["load"], cfg["aff"], cfg["cap"], np.ones(m, np.uint8), 2)
for k in range(ticks):
    alive = synth.churn_mask(m, 2 + k)
    if k in (0, 1, 5, 20, 60, ticks - 1):
        pend = (ref == 0xFFFFFFFF) | (alive[np.minimum(ref, m - 1)] == 0)
        P = int(pend.sum())
        frac = lambda g: float(pend[:n // g * g].reshape(-1, g).any(1).mean())
        d = np.diff(np.concatenate([[0], pend.view(np.int8), [0]]))
        rl = np.flatnonzero(d == -1) - np.flatnonzero(d == 1)
        per_wg = pend[:n // 256 * 256].reshape(256, -1).sum(1)
        print("tick %2d: pending %d | groups holding one: 4 rows %.3f, 16 rows (64 B) %.3f, 32 rows (128 B) %.3f, 256 rows %.3f | "
              "runs %d, mean length %.2f | per 1/256 of the table: mean %d, max %d" % (
                  k, P, frac(4), frac(16), frac(32), frac(256), len(rl), rl.mean(), per_wg.mean(), per_wg.max()), flush=True)
    ref, used, st = pyoracle.tick(ref, cfg["load"], cfg["aff"], cfg["cap"], alive, 2)
