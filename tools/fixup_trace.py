#!/usr/bin/env python3
"""Where the fix-up kernels of a config-5 churn tick spend their time: phase traces written by the kernels themselves
(wall_clock64, 100 MHz) for k_cut_find (+ the block search inside it), k_cut_apply_rank and both k_spill_apply rounds.
Usage: fixup_trace.py [ticks] [auto|never] [churn|contended|skew]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
g = rio_gp.GpuPlacement(n, m)
if len(sys.argv) > 2:
    g.set_compact(sys.argv[2])
workload = sys.argv[3] if len(sys.argv) > 3 else "churn"   # churn | contended (cold solves, capacity 0.9 x load)
if workload in ("contended", "skew"):
    if workload == "contended":
        g.set_nodes((cfg["cap"].astype(np.float64) * 0.72).astype(np.uint64), cfg["alive"])
        g.set_objects(n, cfg["load"], cfg["aff"])
    else:   # Pareto-skewed affinity: a few hot nodes oversubscribed, 9.4 M rows go to the water-fill
        g.set_nodes(cfg["cap"], cfg["alive"])
        g.set_objects(n, cfg["load"], np.minimum((np.random.default_rng(1).pareto(1.1, n)).astype(np.int64), m - 1).astype(np.uint32))
    g.solve()
    g.solve()
    g.cut_trace(True)
    for k in range(ticks):
        st = g.solve()
else:
    g.set_nodes(cfg["cap"], cfg["alive"])
    g.set_objects(n, cfg["load"], cfg["aff"])
    g.set_assign(synth.warm_assign(n, m))
    g.tick()
    g.cut_trace(True)
    for k in range(ticks):
        g.set_alive_all(synth.churn_mask(m, 2 + k))
        st = g.tick()
us = lambda x: float(x) / 100.0
def table(t, names, first_tile=False):
    tr = g.ktrace(t).astype(np.int64)
    live = tr[:, 0] > 0
    if not live.any():
        return None
    t0 = tr[live, 0].min()
    rec = {"workgroups": int(live.sum())}
    for i, nm in enumerate(names):
        if i == 0:
            rec["start_spread_us"] = us(tr[live, 0].max() - t0)
            continue
        d = tr[live, i] - tr[live, i - 1]
        rec[nm + "_us_median_max"] = [us(np.median(d)), us(d.max())]
    rec["end_minus_first_start_us_max"] = us(tr[live, len(names) - 1].max() - t0)
    if first_tile:   # where along the index-ordered prefix are the slow workgroups?  mean of 16 consecutive workgroups each
        idx = np.flatnonzero(live)
        rows_d = (tr[:, 3] - tr[:, 2])[live]
        sync_d = (tr[:, 4] - tr[:, 3])[live]
        nb = max(1, len(idx) // 16)
        rec["rows_us_mean_by_16th_of_the_grid"] = [round(us(rows_d[k * nb:(k + 1) * nb].mean()), 2) for k in range(16) if len(rows_d[k * nb:(k + 1) * nb])]
        rec["rows+sync_us_mean_by_16th_of_the_grid"] = [round(us((rows_d + sync_d)[k * nb:(k + 1) * nb].mean()), 2) for k in range(16) if len(rows_d[k * nb:(k + 1) * nb])]
        rec["rows_us_deciles"] = [round(us(x), 2) for x in np.percentile(rows_d, [10, 30, 50, 70, 90, 100])]
    if first_tile:   # slots 6 / 7: wave 0's first tile — scan + bracket done, rows done (0 when the wave had no tile)
        has = live & (tr[:, 6] > tr[:, 2]) & (tr[:, 7] >= tr[:, 6])
        if has.any():
            rec["wave0_first_tile_us_median"] = {"lo_run search + load wait + scan + bracket": us(np.median(tr[has, 6] - tr[has, 2])),
                                                 "its rows": us(np.median(tr[has, 7] - tr[has, 6])),
                                                 "rest of the wave's tiles": us(np.median(tr[has, 3] - tr[has, 7])),
                                                 "waves_with_a_tile": int(has.sum())}
    return rec
out = {"last_tick": st,
       "k_spill_apply_round0": table(0, ["start", "prefix+pending", "C[] build", "rows", "block sync", "used_cur atomics"], True),
       "k_spill_apply_last": table(1, ["start", "prefix+pending", "C[] build", "rows", "block sync", "used_cur atomics"], True),
       "k_cut_apply_rank": table(2, ["start", "thr+alive load", "rows", "reduce"])}
tr = g.ktrace(3).astype(np.int64)
live = tr[:, 0] > 0
if live.any():
    out["k_cut_find"] = {"workgroups": int(live.sum()), "items": int(tr[live, 3].max()),
                         "item_table_us_median": us(np.median(tr[live, 1] - tr[live, 0])),
                         "items_us_median_max": [us(np.median(tr[live, 2] - tr[live, 1])), us((tr[live, 2] - tr[live, 1]).max())],
                         "end_minus_first_start_us_max": us(tr[live, 2].max() - tr[live, 0].min())}
ct = g.cut_trace(False, read=True).astype(np.int64)
busy = ct[:, 5] > 0
if busy.any():
    out["block_search_last_item_per_wg"] = {"workgroups_with_nodes": int(busy.sum()), "nodes_median_max": [float(np.median(ct[busy, 5])), int(ct[busy, 5].max())],
                                            "P0_us_median": us(np.median(ct[busy, 1] - ct[busy, 0])), "passes_us_median": us(np.median(ct[busy, 2])),
                                            "walks_us_median": us(np.median(ct[busy, 7])), "row_search_us_median": us(np.median(ct[busy, 3])), "S_median": float(np.median(ct[busy, 6]))}
if busy.any():   # the three slowest block searches, phase by phase
    tot = ct[:, 4] - ct[:, 0]
    order = [int(b) for b in np.argsort(-tot * busy)[:3]]
    out["slowest_block_searches"] = [{"workgroup": b, "nodes": int(ct[b, 5]), "S": int(ct[b, 6]), "total_us": us(tot[b]),
                                      "P0_us": us(ct[b, 1] - ct[b, 0]), "passes_us": us(ct[b, 2]), "walks_us": us(ct[b, 7]),
                                      "row_search_us": us(ct[b, 3])} for b in order]
print(json.dumps(out, indent=1))
g.close()
