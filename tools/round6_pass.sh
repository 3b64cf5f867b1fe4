#!/bin/bash
# The evidence pass of build round 6, in two gpurun calls (each half stays under 20 GPU-minutes).
#   tools/round6_pass.sh <tag> a   tests (every GPU test, no -x), smoke, bench N=1 (parity + in-run PMC + config 2 / config 4 on one GPU /
#                                  config 5), bench config 5 as its own line, rocprofv3 stats of the bench command, the request path
#                                  over its whole size range (tools/pp_sizes.py, pp_probe.py + timelines), the trait layer under
#                                  1..256 callers and the two C hosts (tools/c_hosts.sh), churn timeline
#   tools/round6_pass.sh <tag> b   per-kernel roofline records (tools/kernel_roofline.sh), per-operation rates, clean_server,
#                                  latencies, churn variants (config 3 / config 4 tables), fuzz campaign, fix-up traces, contended /
#                                  skewed cold solves, the sharded bench with 2 and 8 ranks on the one GPU, the sharded soak
TAG=${1:-round6}
HALF=${2:-a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
if [ "$HALF" = "a" ]; then
  ( timeout 1700 python -m pytest tests -m gpu -q -rf --timeout=900 2>&1 | tail -60 ) > $OUT/${TAG}_pytest_gpu.log
  ( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 ) > $OUT/${TAG}_smoke.log
  timeout 1200 python bench.py --steps 200 --warmup 20 > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err
  timeout 600 python bench.py --workload c5 --steps 100 --warmup 10 > $OUT/${TAG}_bench_c5.json 2> $OUT/${TAG}_bench_c5.err
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o ${TAG} -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-cold --no-c4 --no-c2 --no-c5 --no-pmc --no-parity > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
  cd $ROOT
  bash tools/prof_churn.sh ${TAG} > /dev/null 2>&1
  timeout 600 python tools/pp_sizes.py 60 > $OUT/${TAG}_pp_sizes.json 2> $OUT/${TAG}_pp_sizes.txt
  timeout 300 python tools/pp_probe.py > $OUT/${TAG}_place_pending.json 2> $OUT/${TAG}_place_pending.err
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp_tl -o pp -- python $ROOT/tools/pp_probe.py > /dev/null 2>&1; python $ROOT/tools/pp_timeline.py /tmp/pp_tl/pp_kernel_trace.csv 2>/dev/null | head -40 ) > $OUT/${TAG}_place_pending_timeline.txt 2>&1
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ppm_tl -o ppm -- python $ROOT/tools/pp_mid_timeline.py > /dev/null 2>&1; python $ROOT/tools/pp_mid_timeline.py /tmp/ppm_tl/ppm_kernel_trace.csv ) > $OUT/${TAG}_pp_mid_timeline.txt 2>&1
  bash tools/c_hosts.sh ${TAG} > /dev/null 2>&1
  echo "---- pytest"; tail -25 $OUT/${TAG}_pytest_gpu.log
  echo "---- smoke"; cat $OUT/${TAG}_smoke.log
  echo "---- bench"; cut -c1-2500 $OUT/${TAG}_bench_n1.json; tail -3 $OUT/${TAG}_bench_n1.err
  echo "---- bench c5"; cut -c1-1500 $OUT/${TAG}_bench_c5.json; tail -3 $OUT/${TAG}_bench_c5.err
  echo "---- kernel stats"; find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs cut -c1-60,180-300 | head -8
  echo "---- request sizes"; cat $OUT/${TAG}_pp_sizes.txt
  echo "---- place_pending"; cut -c1-700 $OUT/${TAG}_place_pending.json; cat $OUT/${TAG}_place_pending_timeline.txt $OUT/${TAG}_pp_mid_timeline.txt
  echo "---- hosts"; cut -c1-300 $OUT/${TAG}_c_host.json; grep -v '"threads": 4,' $OUT/${TAG}_c_host_threads.json | cut -c1-240
  tail -14 $OUT/${TAG}_churn_timeline.txt
else
  bash tools/kernel_roofline.sh ${TAG} > $OUT/${TAG}_kroof.log 2>&1
  timeout 600 python tools/measure_ops.py ${TAG} > $OUT/${TAG}_ops.log 2>&1
  timeout 200 python tools/clean_probe.py > $OUT/${TAG}_clean.json 2> $OUT/${TAG}_clean.err
  timeout 300 python tools/crud_ab.py 5 > $OUT/${TAG}_crud_ab.log 2>&1
  cp $OUT/crud_ab.json $OUT/${TAG}_crud_ab.json
  bash tools/prof_crud.sh ${TAG} > $OUT/${TAG}_prof_crud.txt 2>&1        # per-kernel durations: CRUD 10 M, requests 1 M / 10 M, both chunk forms
  bash tools/prof_binding.sh ${TAG} > $OUT/${TAG}_prof_binding.txt 2>&1  # ... of the solves in which capacity binds (contended / skew)
  bash tools/clean_ab.sh ${TAG} > /dev/null 2>&1
  timeout 200 python tools/quiet_overlap_ab.py 200 > $OUT/${TAG}_quiet_overlap_ab.json 2>&1   # quiet ticks: k_resolve beside the next scan | on the main stream
  timeout 150 python tools/pp_gather_trace.py 10000000 > $OUT/${TAG}_pp_gather_trace.json 2>&1
  timeout 200 python tools/binding_trace.py contended 6 > $OUT/${TAG}_binding_trace_contended.json 2>&1
  ( timeout 200 python tools/sync_probe.py; timeout 200 python tools/latency_probe.py | tail -1; timeout 200 python tools/latency_small_ops.py | tail -1; timeout 200 python tools/tick_rate_probe.py ) > $OUT/${TAG}_latency.txt 2>&1
  timeout 300 python tools/c5_variants.py 100 60 c3 > $OUT/${TAG}_c5_scan_variants.json 2> $OUT/${TAG}_c5_scan_variants.err
  timeout 400 python tools/c5_variants.py 60 12 c4 "auto,never,auto#2,never#2" > $OUT/${TAG}_c4_variants.json 2> $OUT/${TAG}_c4_variants.err
  timeout 500 python tests/test_gpu_fuzz.py 200 1000 > $OUT/${TAG}_fuzz.json 2> $OUT/${TAG}_fuzz.err
  timeout 300 python tools/fill_trace.py 6 60 > $OUT/${TAG}_fill_trace.json 2> $OUT/${TAG}_fill_trace.err
  for w in churn contended skew; do
    timeout 300 python tools/slowpath_workload.py $w 40 > $OUT/${TAG}_slowpath_$w.json 2> $OUT/${TAG}_slowpath_$w.err
  done
  for nr in 2 8; do   # (started plainly: bench.py starts its own ranks)
    timeout 900 python bench.py --gpus $nr --steps 10 --warmup 3 --total-objects 8000000 --objects 1000000 --backend gloo --same-device > $OUT/${TAG}_bench_sharded_${nr}ranks_one_gpu.json 2> $OUT/${TAG}_bench_sharded_${nr}ranks_one_gpu.err
  done
  timeout 600 python tools/soak_sharded.py 3 150 150000 96 5 > $OUT/${TAG}_soak_sharded.json 2> $OUT/${TAG}_soak_sharded.err
  echo "---- kernel roofline"; cat $OUT/${TAG}_kernel_roofline.txt
  echo "---- ops"; tail -25 $OUT/${TAG}_ops.log
  echo "---- clean"; cat $OUT/${TAG}_clean.json
  echo "---- crud"; tail -9 $OUT/${TAG}_crud_ab.log; cat $OUT/${TAG}_prof_crud.txt; cat $OUT/${TAG}_prof_binding.txt | cut -c1-200
  echo "---- quiet ticks"; grep -E "k_resolve|us_per_tick|equal" $OUT/${TAG}_quiet_overlap_ab.json
  echo "---- gather phases"; head -12 $OUT/${TAG}_pp_gather_trace.json
  echo "---- slow path"; for w in churn contended skew; do cut -c1-260 $OUT/${TAG}_slowpath_$w.json; done
  echo "---- churn variants"; cut -c1-1200 $OUT/${TAG}_c5_scan_variants.json
  echo "---- fuzz"; cat $OUT/${TAG}_fuzz.json; tail -5 $OUT/${TAG}_fuzz.err
  echo "---- churn variants, config 4"; cut -c1-1200 $OUT/${TAG}_c4_variants.json
  echo "---- latencies"; cat $OUT/${TAG}_latency.txt
  echo "---- sharded, ranks on one GPU"; for nr in 2 8; do cut -c1-900 $OUT/${TAG}_bench_sharded_${nr}ranks_one_gpu.json; tail -2 $OUT/${TAG}_bench_sharded_${nr}ranks_one_gpu.err; done
  echo "---- soak"; tail -3 $OUT/${TAG}_soak_sharded.json; tail -3 $OUT/${TAG}_soak_sharded.err
fi
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +4M -delete
