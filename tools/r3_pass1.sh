#!/bin/bash
# round-3 pass 1: every GPU test after the advisor fixes, the config-5 line and the per-kernel churn timeline (baseline of this pool)
TAG=${1:-r3a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -rf --timeout=900 2>&1 | tail -40 ) > $OUT/${TAG}_pytest_gpu.log
timeout 600 python bench.py --workload c5 --steps 100 --warmup 10 > $OUT/${TAG}_bench_c5.json 2> $OUT/${TAG}_bench_c5.err
bash tools/prof_churn.sh ${TAG} > /dev/null 2>&1
find $OUT -name "*kernel_trace.csv" -size +4M -delete
echo "---- pytest"; tail -25 $OUT/${TAG}_pytest_gpu.log
echo "---- bench c5"; cut -c1-1600 $OUT/${TAG}_bench_c5.json; tail -3 $OUT/${TAG}_bench_c5.err
echo "---- timeline"; tail -14 $OUT/${TAG}_churn_timeline.txt
cat $OUT/${TAG}_prof_churn.json
