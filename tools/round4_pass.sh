#!/bin/bash
# One gpurun call = one evidence pass of build round 3.  Usage: tools/round4_pass.sh <tag> [quick]
#   tests (every GPU test, no -x), smoke, bench N=1 (parity + in-run PMC + config 2 / config 4 on one GPU / config 5),
#   bench config 5 as its own line, rocprofv3 stats of the bench command, churn timeline, per-kernel roofline records
#   (tools/kernel_roofline.sh), per-operation rates (tools/measure_ops.py), request path (tools/pp_probe.py + timeline),
#   clean_server, latencies, contended / skewed cold solves, the sharded bench with 2 and 8 ranks on the one GPU
TAG=${1:-round4}
QUICK=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 1700 python -m pytest tests -m gpu -q -rf --timeout=900 2>&1 | tail -60 ) > $OUT/${TAG}_pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 ) > $OUT/${TAG}_smoke.log
timeout 1200 python bench.py --steps 200 --warmup 20 > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err
timeout 600 python bench.py --workload c5 --steps 100 --warmup 10 > $OUT/${TAG}_bench_c5.json 2> $OUT/${TAG}_bench_c5.err
if [ -z "$QUICK" ]; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o ${TAG} -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-cold --no-c4 --no-c2 --no-c5 --no-pmc --no-parity > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
  cd $ROOT
  bash tools/prof_churn.sh ${TAG} > /dev/null 2>&1
  bash tools/kernel_roofline.sh ${TAG} > $OUT/${TAG}_kroof.log 2>&1
  timeout 600 python tools/measure_ops.py ${TAG} > $OUT/${TAG}_ops.log 2>&1
  timeout 300 python tools/pp_probe.py > $OUT/${TAG}_place_pending.json 2> $OUT/${TAG}_place_pending.err
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp_tl -o pp -- python $ROOT/tools/pp_probe.py > /dev/null 2>&1; python $ROOT/tools/pp_timeline.py /tmp/pp_tl/pp_kernel_trace.csv ) > $OUT/${TAG}_place_pending_timeline.txt 2>&1
  timeout 200 python tools/clean_probe.py > $OUT/${TAG}_clean.json 2> $OUT/${TAG}_clean.err
  timeout 300 python tools/crud_ab.py 5 > $OUT/${TAG}_crud_ab.log 2>&1
  ( timeout 200 python tools/sync_probe.py; timeout 200 python tools/latency_probe.py | tail -1; timeout 200 python tools/latency_small_ops.py | tail -1; timeout 200 python tools/tick_rate_probe.py ) > $OUT/${TAG}_latency.txt 2>&1
  timeout 300 python tools/c5_variants.py 100 60 c3 > $OUT/${TAG}_c5_variants.json 2> $OUT/${TAG}_c5_variants.err
  timeout 400 python tools/c5_variants.py 60 12 c4 "auto,never,auto#2,never#2" > $OUT/${TAG}_c4_variants.json 2> $OUT/${TAG}_c4_variants.err
  timeout 500 python tests/test_gpu_fuzz.py 200 1000 > $OUT/${TAG}_fuzz.json 2> $OUT/${TAG}_fuzz.err
  timeout 300 python tools/fill_trace.py 6 60 > $OUT/${TAG}_fill_trace.json 2> $OUT/${TAG}_fill_trace.err
  ( for k in 4096 2000 1000; do timeout 200 python tools/pp_one_trace.py $k 200; done ) > $OUT/${TAG}_pp_host_batches.txt 2> $OUT/${TAG}_pp_host_batches.err
  ( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_c4 -o c4 -- python $ROOT/tools/c4_tick_probe.py 8 ) > $OUT/${TAG}_c4_tick.json 2> $OUT/${TAG}_c4_tick.err
  for w in churn contended skew; do
    timeout 300 python tools/slowpath_workload.py $w 40 > $OUT/${TAG}_slowpath_$w.json 2> $OUT/${TAG}_slowpath_$w.err
  done
  for nr in 2 8; do
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $nr --master-addr 127.0.0.1 --master-port $((29600 + nr)) bench.py --gpus $nr --steps 10 --warmup 3 --total-objects 8000000 --objects 1000000 --backend gloo --same-device > $OUT/${TAG}_bench_sharded_${nr}ranks_one_gpu.json 2> $OUT/${TAG}_bench_sharded_${nr}ranks_one_gpu.err
  done
fi
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +4M -delete
echo "---- pytest"; tail -25 $OUT/${TAG}_pytest_gpu.log
echo "---- smoke"; cat $OUT/${TAG}_smoke.log
echo "---- bench"; cut -c1-2500 $OUT/${TAG}_bench_n1.json; tail -3 $OUT/${TAG}_bench_n1.err
echo "---- bench c5"; cut -c1-1500 $OUT/${TAG}_bench_c5.json; tail -3 $OUT/${TAG}_bench_c5.err
if [ -z "$QUICK" ]; then
  echo "---- kernel stats"; find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs cut -c1-60,180-300 | head -8
  echo "---- kernel roofline"; cat $OUT/${TAG}_kernel_roofline.txt
  echo "---- ops"; tail -25 $OUT/${TAG}_ops.log
  echo "---- place_pending"; cut -c1-1200 $OUT/${TAG}_place_pending.json; cat $OUT/${TAG}_place_pending_timeline.txt
  echo "---- clean"; cat $OUT/${TAG}_clean.json
  echo "---- slow path"; for w in churn contended skew; do cut -c1-260 $OUT/${TAG}_slowpath_$w.json; done
  tail -14 $OUT/${TAG}_churn_timeline.txt
  echo "---- churn variants"; cut -c1-3000 $OUT/${TAG}_c5_variants.json
  echo "---- fuzz"; cat $OUT/${TAG}_fuzz.json; tail -5 $OUT/${TAG}_fuzz.err
  echo "---- churn variants, config 4"; cut -c1-3000 $OUT/${TAG}_c4_variants.json
  echo "---- host batches"; cat $OUT/${TAG}_pp_host_batches.txt
  echo "---- c4 churn tick"; cat $OUT/${TAG}_c4_tick.json; find $OUT/${TAG}_prof_c4 -name "*kernel_stats.csv" | head -1 | xargs cut -c1-60,180-300 | head -12
  echo "---- latencies"; cat $OUT/${TAG}_latency.txt
  echo "---- sharded, ranks on one GPU"; for nr in 2 8; do cut -c1-900 $OUT/${TAG}_bench_sharded_${nr}ranks_one_gpu.json; tail -2 $OUT/${TAG}_bench_sharded_${nr}ranks_one_gpu.err; done
fi
