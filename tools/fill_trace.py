#!/usr/bin/env python3
"""Phase traces of the fix-up kernels over config-5 churn ticks (lab build: rio_gp_debug_ktrace).  Per kernel the median /
max over the workgroups of every phase's duration (us), measured by the kernels themselves with wall_clock64 (100 MHz).
  table 0 k_resolve with the in-kernel cut search: start | column sums | cut blocks located + RP | ranges known | batch requested |
          first half searched | both halves | end
  table 1 k_fill round 0, table 2 later rounds: start | prologue + pending | order of the nodes | pass A | pass B set up |
          pass B rows | block sync | end
Usage: fill_trace.py [ticks] [ticks of churn before the traced ones]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import rio_gp, synth
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg = synth.config("c3")
n, m = cfg["n"], cfg["m"]
g = rio_gp.LabPlacement(n, m)
g.set_nodes(cfg["cap"], cfg["alive"])
g.set_objects(n, cfg["load"], cfg["aff"])
g.set_assign(synth.warm_assign(n, m))
g.tick()
warm_ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 60   # the stream's steady state (the first ticks are ~10 us cheaper)
for k in range(warm_ticks):
    g.set_alive_all(synth.churn_mask(m, 2 + k))
    g.tick_async()
g.tick_wait()
g.ktrace(True)
import ctypes as C
for k in range(warm_ticks, warm_ticks + ticks):
    mask = synth.churn_mask(m, 2 + k)
    before = g.get_assign() if k == warm_ticks + ticks - 1 else None
    g.set_alive_all(mask)
    st = g.tick()
# pending rows per wave range of the last tick (the split the kernels use)
L = rio_gp.lab_lib()
nw = C.c_uint32(0)
L.rio_gp_debug_wave_row_lo(n, m, 0, C.byref(nw))
lo = np.array([L.rio_gp_debug_wave_row_lo(n, m, w, None) for w in range(nw.value + 1)], dtype=np.int64)
lo = np.minimum(lo, n)
pend = (before == 0xFFFFFFFF) | (mask[np.minimum(before, m - 1)] == 0)
cs = np.concatenate([[0], np.cumsum(pend)])
per_wave = cs[lo[1:]] - cs[lo[:-1]]
per_wg = per_wave.reshape(-1, 16)
out = {"last_tick": st, "pending_rows_per_wave": {"mean": float(per_wave.mean()), "max": int(per_wave.max()), "waves_over_256": int((per_wave > 256).sum()),
                                                  "waves_over_512": int((per_wave > 512).sum())},
       "pending_rows_per_wg": {"mean": float(per_wg.sum(1).mean()), "max": int(per_wg.sum(1).max()), "argmax": int(per_wg.sum(1).argmax()),
                               "wg_122_125": [int(x) for x in per_wg.sum(1)[122:126]], "max_wave_in_122_125": [int(x) for x in per_wg.max(1)[122:126]]}}
names = {0: "k_resolve<search>", 1: "k_fill round 0", 2: "k_fill round 1"}
names[4] = "order of the nodes (inside the last k_fill)"
for t in (0, 1, 2, 4):
    tr = g.ktrace(True, t).astype(np.int64)
    rows = tr[tr[:, 0] > 0]
    if not len(rows):
        continue
    t0 = rows[:, 0].min()
    d = np.diff(rows, axis=1) / 100.0
    d[rows[:, 1:] == 0] = 0
    out[names[t]] = {"workgroups": int(len(rows)), "first_start_to_last_end_us": float((rows[:, 7].max() - t0) / 100.0),
                     "start_skew_us": float((rows[:, 0].max() - t0) / 100.0),
                     "phase_median_us": [round(float(np.median(d[:, c])), 2) for c in range(7)],
                     "phase_max_us": [round(float(d[:, c].max()), 2) for c in range(7)],
                     "wg_total_median_us": round(float(np.median((rows[:, 7] - rows[:, 0]) / 100.0)), 2),
                     "wg_total_max_us": round(float(((rows[:, 7] - rows[:, 0]) / 100.0).max()), 2),
                     "slowest_wgs": [{"wg": int(b), "phases_us": [round(float(x), 1) for x in (np.diff(tr[b]) / 100.0)]}
                                     for b in np.argsort(-(tr[:, 7] - tr[:, 0]))[:6] if tr[b, 0] > 0]}
# the three kernels on one time axis (us from the first workgroup start of k_resolve): first start / last end of each
names[3] = "k_scan"
tabs = [g.ktrace(True, t).astype(np.int64) for t in (0, 1, 2, 3)]
base = min(int(t[t[:, 0] > 0][:, 0].min()) for t in tabs if (t[:, 0] > 0).any())
out["time_axis_us"] = {names[i]: [round((int(t[t[:, 0] > 0][:, 0].min()) - base) / 100.0, 2), round((int(t[:, 7].max()) - base) / 100.0, 2)]
                       for i, t in enumerate(tabs) if (t[:, 0] > 0).any()}
g.ktrace(False)
print(json.dumps(out))
g.close()
