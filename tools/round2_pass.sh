#!/bin/bash
# One gpurun call = one evidence pass of build round 2.  Usage: tools/round2_pass.sh <tag> [quick]
#   tests (every GPU test, no -x: all failures in one pass), smoke, bench N=1 (parity + in-run PMC + config 4 on one GPU),
#   bench config 5 (synchronous and pipelined ticks), rocprofv3 stats of the bench command, churn timeline,
#   per-kernel roofline records (tools/kernel_roofline.sh), per-operation rates (tools/measure_ops.py)
TAG=${1:-round2}
QUICK=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 1700 python -m pytest tests -m gpu -q -rf --timeout=900 2>&1 | tail -60 ) > $OUT/${TAG}_pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 ) > $OUT/${TAG}_smoke.log
timeout 900 python bench.py --steps 200 --warmup 20 > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err
timeout 600 python bench.py --workload c5 --steps 100 --warmup 10 > $OUT/${TAG}_bench_c5.json 2> $OUT/${TAG}_bench_c5.err
if [ -z "$QUICK" ]; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o ${TAG} -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-cold --no-c4 --no-pmc --no-parity > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
  cd $ROOT
  bash tools/prof_churn.sh ${TAG} > /dev/null 2>&1
  bash tools/kernel_roofline.sh ${TAG} > $OUT/${TAG}_kroof.log 2>&1
  timeout 600 python tools/measure_ops.py ${TAG} > $OUT/${TAG}_ops.log 2>&1
  timeout 300 python tools/fixup_trace.py 6 > $OUT/${TAG}_fixup_trace.json 2> $OUT/${TAG}_fixup_trace.err
  timeout 300 python tools/crud_ab.py 5 > $OUT/${TAG}_crud_ab.log 2>&1
  ( timeout 200 python tools/sync_probe.py; timeout 200 python tools/latency_probe.py | tail -1; timeout 200 python tools/latency_small_ops.py | tail -1 ) > $OUT/${TAG}_latency.txt 2>&1
  for w in churn contended skew; do
    timeout 300 python tools/slowpath_workload.py $w 40 > $OUT/${TAG}_slowpath_$w.json 2> $OUT/${TAG}_slowpath_$w.err
  done
  timeout 300 python tools/slowpath_workload.py churn 40 auto fusedk > $OUT/${TAG}_slowpath_churn_fusedk.json 2>> $OUT/${TAG}_slowpath_churn.err
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
  echo "---- slow path"; for w in churn contended skew churn_fusedk; do cut -c1-260 $OUT/${TAG}_slowpath_$w.json; done
  tail -14 $OUT/${TAG}_churn_timeline.txt
  echo "---- latencies"; cat $OUT/${TAG}_latency.txt
  echo "---- fix-up phase traces"; cat $OUT/${TAG}_fixup_trace.json; tail -3 $OUT/${TAG}_fixup_trace.err
fi
