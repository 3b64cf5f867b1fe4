#!/bin/bash
# One gpurun call for a full evidence pass: parity suite, smoke, bench (with cpu_baseline), rocprofv3 kernel stats of the
# same bench command, the two PMC passes (separate runs, --kernel-trace only), churn timeline.  Usage: tools/gpu_round.sh <tag>
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > $OUT/${TAG}_pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 ) > $OUT/${TAG}_smoke.log
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o ${TAG} -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-cold > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o f -- python $ROOT/tools/pmc_workload.py > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o w -- python $ROOT/tools/pmc_workload.py > $OUT/${TAG}_pmc_write.log 2>&1
cd $ROOT
python tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write 10000000 $OUT/${TAG}_traffic.json
bash tools/prof_churn.sh ${TAG} > /dev/null 2>&1
for w in churn contended skew; do
  timeout 300 python tools/slowpath_workload.py $w 40 > $OUT/${TAG}_slowpath_$w.json 2> $OUT/${TAG}_slowpath_$w.err
done
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +4M -delete
echo "---- pytest"; tail -4 $OUT/${TAG}_pytest_gpu.log
echo "---- smoke"; cat $OUT/${TAG}_smoke.log
echo "---- bench"; cut -c1-1800 $OUT/${TAG}_bench_n1.json; tail -3 $OUT/${TAG}_bench_n1.err
echo "---- kernel stats"; find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs cut -c1-50,180-300 | head -8
echo "---- traffic"; cat $OUT/${TAG}_traffic.json
echo "---- slow path"; for w in churn contended skew; do cut -c1-260 $OUT/${TAG}_slowpath_$w.json; done
tail -12 $OUT/${TAG}_churn_timeline.txt
