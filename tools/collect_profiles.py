#!/usr/bin/env python3
"""Copy the summaries of one evidence pass (tools/round6_pass.sh <tag> a|b, merged back under gpurun_out/) into profiles/ under
the round prefix: gpurun_out/ is scratch, profiles/ is tracked.  Usage: collect_profiles.py <tag> [round-prefix]"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
pre = sys.argv[2] if len(sys.argv) > 2 else tag
src, dst = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
def cp(a, b):
    a = os.path.join(src, a)
    if os.path.exists(a):
        shutil.copyfile(a, os.path.join(dst, b))
        print("profiles/" + b)
for name in ("bench_n1.json", "bench_c5.json", "kernel_roofline.json", "kernel_roofline.txt", "churn_timeline.txt",
             "ops.json", "pytest_gpu.log", "smoke.log", "slowpath_churn.json", "slowpath_contended.json", "slowpath_skew.json",
             "latency.txt", "place_pending.json", "place_pending_timeline.txt", "clean.json",
             "bench_sharded_2ranks_one_gpu.json", "bench_sharded_8ranks_one_gpu.json", "c5_variants.json", "c4_variants.json", "fill_trace.json",
             "pp_host_batches.txt", "c4_tick.json", "fuzz.json", "pp_sizes.json", "pp_sizes.txt", "pp_mid_timeline.txt",
             "c_host.json", "c_host_threads.json", "soak_sharded.json",
             # round 6: per-kernel durations of the window-partitioned batches and of the solves in which capacity binds, A/Bs, traces
             "crud_ab.json", "crud_ab.log", "quiet_overlap_ab.json", "prof_crud.txt", "prof_binding.txt", "clean_ab.json", "pp_gather_trace.json",
             "binding_trace_contended.json", "c5_scan_variants.json", "binding_contended_kernel_stats.csv", "binding_skew_kernel_stats.csv",
             "crud_kernel_stats.csv", "pp_1000000_kernel_stats.csv", "pp_10000000_kernel_stats.csv", "pp_small_kernel_stats.csv"):
    cp("%s_%s" % (tag, name), "%s_%s" % (pre, name))
if not os.path.exists(os.path.join(src, tag + "_crud_ab.json")):
    cp("crud_ab.json", pre + "_crud_ab.json")
for f in glob.glob(os.path.join(src, tag + "_prof", "*kernel_stats.csv")):
    shutil.copyfile(f, os.path.join(dst, pre + "_kernel_stats.csv")); print("profiles/%s_kernel_stats.csv" % pre)
for f in glob.glob(os.path.join(src, tag + "_prof_churn", "*kernel_stats.csv")):
    shutil.copyfile(f, os.path.join(dst, pre + "_slowpath_churn_kernel_stats.csv")); print("profiles/%s_slowpath_churn_kernel_stats.csv" % pre)
b = os.path.join(src, tag + "_bench_n1.json")
if os.path.exists(b):  # the in-run PMC passes of the bench line, as the stand-alone record bench.py falls back to
    d = json.loads(open(b).read())
    r = d["roofline"]
    if r.get("traffic"):
        doc = {"n_rows": d["config"]["objects_per_gpu"], "hbm_bytes_per_launch": r["traffic"], "source": r["traffic_source"]}
        doc.update(r.get("traffic_detail") or {})
        json.dump(doc, open(os.path.join(dst, pre + "_traffic.json"), "w"), indent=1)
        print("profiles/%s_traffic.json" % pre)
