#!/bin/bash
# round-3 iteration pass: GPU tests (all, or the files given in $2), config-5 bench line, per-kernel churn timeline.
# Usage: tools/r3_pass.sh <tag> ["pytest args"]
TAG=${1:-r3}
ARGS=${2:-tests}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 1500 python -m pytest $ARGS -m gpu -q -rf --timeout=900 -x 2>&1 | tail -60 ) > $OUT/${TAG}_pytest_gpu.log
timeout 600 python bench.py --workload c5 --steps 100 --warmup 10 > $OUT/${TAG}_bench_c5.json 2> $OUT/${TAG}_bench_c5.err
bash tools/prof_churn.sh ${TAG} > /dev/null 2>&1
for w in contended skew; do
  timeout 300 python tools/slowpath_workload.py $w 40 > $OUT/${TAG}_slowpath_$w.json 2> $OUT/${TAG}_slowpath_$w.err
done
find $OUT -name "*kernel_trace.csv" -size +4M -delete
echo "---- pytest"; tail -45 $OUT/${TAG}_pytest_gpu.log
echo "---- bench c5"; cut -c1-1700 $OUT/${TAG}_bench_c5.json; tail -3 $OUT/${TAG}_bench_c5.err
echo "---- timeline"; tail -12 $OUT/${TAG}_churn_timeline.txt
cut -c1-400 $OUT/${TAG}_prof_churn.json; tail -2 $OUT/${TAG}_prof_churn.err
echo "---- slow path"; for w in contended skew; do cut -c1-300 $OUT/${TAG}_slowpath_$w.json; tail -2 $OUT/${TAG}_slowpath_$w.err; done
