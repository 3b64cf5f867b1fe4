#!/bin/bash
# One gpurun call: parity suite, fix-up A/B (legacy / fused / speculative), bench sanity, kernel stats of the churn tick.
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > $OUT/${TAG}_pytest.log
echo "---- pytest"; tail -8 $OUT/${TAG}_pytest.log
for fx in legacy fused spec default; do
  timeout 300 python tools/slowpath_workload.py churn 40 auto $fx > $OUT/${TAG}_churn_$fx.json 2> $OUT/${TAG}_churn_$fx.err
  echo "churn $fx: $(cut -c1-200 $OUT/${TAG}_churn_$fx.json)"; tail -2 $OUT/${TAG}_churn_$fx.err
done
for w in contended skew; do for fx in legacy default; do
  timeout 300 python tools/slowpath_workload.py $w 20 auto $fx > $OUT/${TAG}_${w}_$fx.json 2> $OUT/${TAG}_${w}_$fx.err
  echo "$w $fx: $(cut -c1-200 $OUT/${TAG}_${w}_$fx.json)"
done; done
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_bench_quick.json 2> $OUT/${TAG}_bench_quick.err
echo "---- bench"; cut -c1-400 $OUT/${TAG}_bench_quick.json; tail -3 $OUT/${TAG}_bench_quick.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_churn -o churn -- python $ROOT/tools/slowpath_workload.py churn 20 > $OUT/${TAG}_prof_churn.json 2> $OUT/${TAG}_prof_churn.err
cd $ROOT
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT/${TAG}_prof_churn -name "*kernel_stats.csv" | head -1 | xargs -r cat | cut -c1-60,200- | head -30
