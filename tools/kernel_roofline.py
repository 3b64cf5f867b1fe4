#!/usr/bin/env python3
"""rocprofv3 outputs of tools/kernel_roofline.sh -> one record per kernel: average duration, algorithmic bytes per launch,
HBM bytes per launch from the PMC counters (FETCH_SIZE + WRITE_SIZE, calibrated on the stream probes of the same session),
achieved algorithmic GB/s, fraction of the 8 TB/s roofline and traffic / algorithmic.  Usage:
  kernel_roofline.py <dir with <phase>_{trace,fetch,write} sub-directories> <out.json>"""
import csv, glob, json, os, re, sys

N, M = 10_000_000, 1024
PEAK = 8000.0


def trace_avg(d):
    """kernel name -> (avg duration us, launches, [durations])"""
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            acc.setdefault(k, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    return {k: (sum(v) / len(v), len(v), v) for k, v in acc.items()}


def pmc_avg(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("riogp::", "")


# (phase, kernel-name regex) -> (label, algorithmic bytes per launch, what the bytes are)
def specs(pending_rows):
    pk = pending_rows   # packed rows of a churn tick
    return [
        ("fast", r"k_scan<false, true, 2, 0", "k_scan (fast path, TPI 2)", 16 * N, "12 B read + 4 B written per row"),
        ("fast", r"k_resolve", "k_resolve", 2 * M * 8 * 256, "H: 2m u64 per block x 256 blocks"),
        ("churn", r"k_inc_scan<2, false>", "k_inc_scan (churn tick: in place, no histogram)", 16 * N, "SURVEY 8d's tick figure, 16 B/row; the kernel itself moves 12 B/row + 12 B per packed pending row + the changed vectors"),
        ("churn", r"k_rebal", "k_rebal (pending rows dealt out evenly + per-block histograms)", 28 * pk + 2 * M * 8 * 256, "12 B in + 16 B out per pending row, H"),
        ("churn_noinc", r"k_scan<false, false, 1, 2", "k_scan<COMPACT> (churn tick, in-place scan switched off)", 16 * N + 16 * pk, "16 B/row + 16 B per packed pending row"),
        ("churn_noinc", r"k_resolve<true>", "k_resolve<SEARCH> (row-range layout)", 2 * M * 8 * 256 + 8 * pk, "H + affinity and load of the packed pending rows"),
        ("churn_noinc", r"k_fill<true, true, true, false>", "k_fill round 0 (row-range layout)", 24 * pk, "as below"),
        ("churn_noinc", r"k_fill<false, false, true, false>", "k_fill round 1 (row-range layout)", 12 * pk, "as below"),
        ("churn", r"k_resolve<true>", "k_resolve<SEARCH> (column sums + exact cuts over the packed rows)", 2 * M * 8 * 256 + 8 * pk, "H + affinity and load of the packed pending rows"),
        ("churn", r"k_fill<true, true, true, false>", "k_fill round 0 (packed rows: re-mark, node order, water-fill)", 24 * pk, "pass A cur/aff/load + pass B next/load/idx of the packed rows (+4 B per placed row)"),
        ("churn", r"k_fill<false, false, true, false>", "k_fill round 1 (packed rows)", 12 * pk, "next + load + idx of the packed rows (upper bound: most are placed by round 0)"),
        ("churn_unpacked", r"k_cut_find<false>", "k_cut_find (whole table, churn)", 12 * N, "upper bound: every block owns a cut"),
        ("churn_unpacked", r"k_fill<false, true, true, false>", "k_fill round 0 (whole table, churn)", 20 * N, "pass A 12 B/row + pass B next/load 8 B/row"),
        ("churn_unpacked", r"k_fill<false, false, true, false>", "k_fill round 1 (whole table, churn)", 8 * N, "next + load per row"),
        ("contended", r"k_cut_find<false>", "k_cut_find (whole table, contended)", 12 * N, "one pass over the blocks that own cuts (all)"),
        ("contended", r"k_fill<false, true, true, false>", "k_fill round 0 (whole table, contended)", 20 * N, "pass A 12 B/row + pass B next/load 8 B/row"),
        ("contended", r"k_fill<false, false, true, false>", "k_fill round 1 (whole table, contended)", 8 * N, "next + load per row"),
        ("contended_packed", r"k_cut_apply<true", "k_cut_apply<PACK> (whole table, contended: exact cuts applied, water-fill rows packed, one pass)", 16 * N, "12 B read + 4 B written per row (+12 B per packed row)"),
        ("contended_packed", r"k_cut_settle", "k_cut_settle (the undecided rows of the cut blocks, final spill totals)", 2 * M * 16 * 8 + 16 * 50000, "Tg [m][16] u64 read + ~16 B per undecided row (a few dozen per cut node); latency-bound"),
        ("contended_packed", r"k_fill<false, false, true, false>", "k_fill rounds 0 and 1 (contended, rows packed by k_cut_apply; average of both launches)", 12 * N // 10, "next + load + idx of ~1 M packed rows"),
        ("crud", r"k_lookup4", "k_lookup4 (10 M random indices)", 12 * N, "idx + gather + out per lookup"),
        ("lookup_seq", r"k_lookup4", "k_lookup4 (10 M sequential indices)", 12 * N, "idx + gather + out per lookup"),
        ("crud", r"k_part_bin<true", "k_part_bin<update> (10 M random)", 8 * N, "idx + node per entry (update = bin + apply: 8 B/op over both)"),
        ("crud", r"k_part_update", "k_part_update (10 M random)", 8 * N, "see k_part_bin"),
        ("crud", r"k_part_bin<false", "k_part_bin<remove> (10 M random)", 8 * N, "idx + row per removal (remove = bin + apply)"),
        ("crud", r"k_part_remove", "k_part_remove (10 M random)", 8 * N, "see k_part_bin"),
        ("crud_plain", r"k_update_elect", "k_update_elect (plain kernels, 10 M random)", 8 * N, "idx + node per entry (update = elect + apply)"),
        ("crud_plain", r"k_update_apply", "k_update_apply (plain kernels, 10 M random)", 8 * N, "see k_update_elect"),
        ("crud_plain", r"k_remove", "k_remove (plain kernel, 10 M random)", 8 * N, "idx + row per removal"),
        ("crud", r"k_clean", "k_clean (10 % of the nodes)", 4 * N + 4 * N // 10, "4 B/row read + 4 B per evicted row"),
        ("clean1", r"k_clean", "k_clean (one node)", 4 * N, "4 B/row read (+4 B per evicted row: 0.1 %)"),
        ("pp", r"k_part_bin<true", "k_part_bin (place_pending, 1 M requests)", 16 * 1000000, "idx + requester in, one 8-byte record out per request"),
        ("pp", r"k_pp_win_gather", "k_pp_win_gather (1 M requests)", 20 * 1000000 + 8 * N, "record in + answer record out + the first touch's dense write per request; the windows' assignment and load columns once"),
        ("pp", r"k_pp_win_unsort", "k_pp_win_unsort (1 M requests)", 16 * 1000000, "sorted answer + batch position in, node + flag out in batch order"),
        ("pp10", r"k_part_bin<true", "k_part_bin (place_pending, 10 M requests)", 16 * 10000000, "idx + requester in, one 8-byte record out per request"),
        ("pp10", r"k_pp_win_gather", "k_pp_win_gather (10 M requests)", 20 * 10000000 + 8 * N, "record in + answer record out + the first touch's dense write per request; the windows' assignment and load columns once"),
        ("pp10", r"k_pp_win_verdict", "k_pp_win_verdict (10 M requests)", 24 * M, "claim, cap, used per requester"),
        ("pp10", r"k_pp_win_unsort", "k_pp_win_unsort (10 M requests)", 16 * 10000000, "sorted answer + batch position in, node + flag out in batch order"),
        ("pp_mid", r"k_ppm_first", "k_ppm_first (16 384 device-resident requests)", 12 * 16384, "idx + requester in, the election word"),
        ("pp_mid", r"k_ppm_gather", "k_ppm_gather (16 384 requests)", 24 * 16384, "idx in, election word + row + load gathered, three columns out"),
        ("pp_mid", r"k_scan<true, true, 1, 0", "k_scan<VIRT> (16 384 requests)", 16 * 16384, "three columns in, the decision out"),
        ("pp_mid", r"k_ppm_output", "k_ppm_output (16 384 requests)", 28 * 16384, "five columns in, node + flag out, the first touches into the table"),
        ("pp_small", r"k_pp_stage", "k_pp_stage (4 096 host-buffer requests: requests + rows -> staging table)", 12 * 4096, "idx + requester in over PCIe, the rows"),
        ("pp_small", r"k_pp_decide", "k_pp_decide (one workgroup: the decision over the staged records)", 40 * 4096, "32 B staged record in, 8 B result out"),
        ("pp_small", r"k_pp_apply", "k_pp_apply (results out over PCIe, first touches into the column)", 16 * 4096, "8 B result in, node + flag out"),
        ("pp_256", r"k_pp_one<256, 1>", "k_pp_one (256 host-buffer requests, one launch)", 28 * 256, "idx + requester in over PCIe, node + flag out, the rows"),
    ]


def main():
    base, out = sys.argv[1], sys.argv[2]
    # calibration from the probes phase
    pf, pw = pmc_avg(os.path.join(base, "probes_fetch"), "FETCH_SIZE"), pmc_avg(os.path.join(base, "probes_write"), "WRITE_SIZE")
    col = 4.0 * N
    def pick(d, needle):
        for k, v in d.items():
            if needle in k:
                return v
        return None
    fs = [x for x in (col / pick(pf, "k_probe_copy") / 1024 if pick(pf, "k_probe_copy") else None,
                      3 * col / pick(pf, "k_probe_readonly") / 1024 if pick(pf, "k_probe_readonly") else None,
                      3 * col / pick(pf, "k_probe_gridstride") / 1024 if pick(pf, "k_probe_gridstride") else None) if x]
    ws = [x for x in (col / pick(pw, "k_probe_copy") / 1024 if pick(pw, "k_probe_copy") else None,
                      col / pick(pw, "k_probe_gridstride") / 1024 if pick(pw, "k_probe_gridstride") else None) if x]
    fscale = sum(fs) / len(fs) if fs else 2.0
    wscale = sum(ws) / len(ws) if ws else 1.0
    pending = None
    try:
        pending = json.load(open(os.path.join(base, "churn_stats.json")))["pending_rows"]
    except Exception:
        pending = 1_000_000
    doc = {"n_rows": N, "n_nodes": M, "peak_GBps": PEAK,
           "calibration": {"fetch_bytes_per_reported_byte": fscale, "write_bytes_per_reported_byte": wscale,
                           "note": "FETCH_SIZE / WRITE_SIZE (KiB) scaled by known-traffic stream probes of the same session "
                                   "(MI355X_MICROARCH.md section HBM: gfx950 FETCH_SIZE reports half of a wide coalesced read); "
                                   "narrow / scattered accesses are NOT covered by that calibration: their counter bytes are a lower bound"},
           "churn_pending_rows": pending, "kernels": []}
    cache = {}
    for phase, pat, label, alg, what in specs(pending):
        if phase not in cache:
            cache[phase] = (trace_avg(os.path.join(base, phase + "_trace")), pmc_avg(os.path.join(base, phase + "_fetch"), "FETCH_SIZE"),
                            pmc_avg(os.path.join(base, phase + "_write"), "WRITE_SIZE"))
        tr, pf_, pw_ = cache[phase]
        names = [k for k in tr if re.search(pat, short(k))]
        if not names:
            doc["kernels"].append({"kernel": label, "phase": phase, "missing": True})
            continue
        durs = [d for k in names for d in tr[k][2]]
        dur = sum(durs) / len(durs)
        f = [pf_[k] for k in names if k in pf_]
        w = [pw_[k] for k in names if k in pw_]
        traffic = None
        if f and w:
            traffic = (sum(f) / len(f)) * 1024 * fscale + (sum(w) / len(w)) * 1024 * wscale
        rec = {"kernel": label, "phase": phase, "launches": len(durs), "avg_us": dur, "algorithmic_bytes": alg, "bytes_are": what,
               "achieved_GBps": alg / dur / 1e3, "frac_of_8TBps": alg / dur / 1e3 / PEAK,
               "counter_bytes": traffic, "traffic_over_algorithmic": (traffic / alg) if traffic else None}
        doc["kernels"].append(rec)
    json.dump(doc, open(out, "w"), indent=1)
    for r in doc["kernels"]:
        if r.get("missing"):
            print("%-46s MISSING" % r["kernel"])
        else:
            print("%-46s %9.1f us  %8.1f GB/s  %5.1f %%  traffic/alg %s" % (r["kernel"], r["avg_us"], r["achieved_GBps"], 100 * r["frac_of_8TBps"],
                  ("%.2f" % r["traffic_over_algorithmic"]) if r["traffic_over_algorithmic"] else "-"))


if __name__ == "__main__":
    main()
