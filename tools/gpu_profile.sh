#!/bin/bash
# On the GPU box (via gpurun): rocprofv3 kernel stats of bench.py + the two PMC passes -> gpurun_out/<tag>_*
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o ${TAG} -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o f -- python $ROOT/tools/pmc_workload.py > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o w -- python $ROOT/tools/pmc_workload.py > $OUT/${TAG}_pmc_write.log 2>&1
cd $ROOT
python tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write 10000000 $OUT/${TAG}_traffic.json
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +4M -delete
echo "---- kernel stats"; find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs head -8
ls -R $OUT/${TAG}_pmc_fetch | head; tail -3 $OUT/${TAG}_pmc_fetch.log
