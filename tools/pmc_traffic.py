#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per k_scan launch.

Method (MI355X_MICROARCH.md §HBM): FETCH_SIZE/WRITE_SIZE are in KiB of fabric requests; on gfx950
FETCH_SIZE under-reports wide coalesced reads by 2x and WRITE_SIZE is uncalibrated, so both are
calibrated on kernels of KNOWN traffic run in the same pass (the stream probes: 1:1 copy = n*4 B read
+ n*4 B written; read-only = 3*n*4 B read), then applied to k_scan.
Usage: pmc_traffic.py <fetch_dir> <write_dir> <n_rows> <out.json>"""
import csv, glob, json, os, sys


def per_kernel(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"]
            acc.setdefault(k, []).append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def pick(d, needle):
    for k, v in d.items():
        if needle in k:
            return v[0]
    return None


def compute(fdir, wdir, n):
    """Counters of the two passes -> dict (see the module docstring); bench.py calls this for its in-run passes."""
    F, W = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    col = 4.0 * n  # bytes of one u32 column
    f_copy, f_ro, f_gs = pick(F, "k_probe_copy"), pick(F, "k_probe_readonly"), pick(F, "k_probe_gridstride")
    w_copy, w_gs = pick(W, "k_probe_copy"), pick(W, "k_probe_gridstride")
    f_scan, w_scan = pick(F, "k_scan"), pick(W, "k_scan")
    f_res, w_res = pick(F, "k_resolve"), pick(W, "k_resolve")
    # bytes per reported KiB unit, from the known-traffic kernels
    f_scale = [x for x in (col / f_copy / 1024 if f_copy else None, 3 * col / f_ro / 1024 if f_ro else None,
                           3 * col / f_gs / 1024 if f_gs else None) if x]
    w_scale = [x for x in (col / w_copy / 1024 if w_copy else None, col / w_gs / 1024 if w_gs else None) if x]
    fs = sum(f_scale) / len(f_scale) if f_scale else None
    ws = sum(w_scale) / len(w_scale) if w_scale else None
    doc = {
        "n_rows": n, "raw_KiB": {"FETCH_SIZE": {k: v[0] for k, v in F.items()}, "WRITE_SIZE": {k: v[0] for k, v in W.items()}},
        "calibration": {"fetch_bytes_per_reported_byte": fs, "write_bytes_per_reported_byte": ws,
                        "fetch_scales": f_scale, "write_scales": w_scale,
                        "note": "scale = known bytes / (counter * 1024) on the stream probes of the same pass; "
                                "the guide's gfx950 correction for wide coalesced reads is 2.0"},
    }
    if f_scan and w_scan and fs and ws:
        rd, wr = f_scan * 1024 * fs, w_scan * 1024 * ws
        doc["k_scan"] = {"read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                         "algorithmic_bytes_per_launch": 16.0 * n, "traffic_over_algorithmic": (rd + wr) / (16.0 * n)}
        doc["hbm_bytes_per_launch"] = rd + wr
    if f_res and w_res and fs and ws:
        doc["k_resolve"] = {"hbm_bytes_per_launch": f_res * 1024 * fs + w_res * 1024 * ws}
    return doc


def main():
    fdir, wdir, n, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    doc = compute(fdir, wdir, n)
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: doc[k] for k in doc if k not in ("raw_KiB",)}))


if __name__ == "__main__":
    main()
