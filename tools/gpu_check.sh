#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, bench, rocprofv3 kernel stats.
# Usage: tools/gpu_check.sh <tag> [quick]
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > $OUT/${TAG}_pytest.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 ) > $OUT/${TAG}_smoke.log
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o ${TAG} -- python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
cd $ROOT
find $OUT/${TAG}_prof -name "*kernel_stats*" | head -3 | while read f; do echo "== $f"; head -20 "$f"; done > $OUT/${TAG}_kernel_stats.txt; ls -R $OUT/${TAG}_prof | head -20 >> $OUT/${TAG}_kernel_stats.txt
# keep only the small summaries
find $OUT/${TAG}_prof -name "*kernel_trace*" -size +8M -delete
echo "---- pytest"; cat $OUT/${TAG}_pytest.log | tail -15
echo "---- smoke"; cat $OUT/${TAG}_smoke.log
echo "---- bench"; cat $OUT/${TAG}_bench.json; tail -3 $OUT/${TAG}_bench.err
echo "---- kernel stats"; cat $OUT/${TAG}_kernel_stats.txt
