#!/bin/bash
# per-kernel durations of the window-partitioned batches (random CRUD of 10 M entries, place_pending of 1 M / 10 M requests) in
# both chunk forms (8 192 / 16 384 entries): rocprofv3 --kernel-trace --stats over tools/crud_ab.py and tools/pp_probe.py.
# Usage: tools/prof_crud.sh <tag>
TAG=${1:-round6}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
show() {
  python - "$1" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if float(r["AverageNs"]) > 3000 and not r["Name"].startswith("__amd"):
        print("%-70s calls %5s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
}
rm -rf /tmp/pc_crud
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_crud -o pc -- python $ROOT/tools/crud_ab.py 6 part_w16384 > $OUT/${TAG}_prof_crud.log 2>&1
f=$(find /tmp/pc_crud -name "*kernel_stats.csv" | head -1); cp $f $OUT/${TAG}_crud_kernel_stats.csv; echo "---- crud, 10 M random entries (the product's geometry)"; show $f
for sz in 1000000 10000000; do
  rm -rf /tmp/pc_pp_$sz
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_pp_$sz -o pc -- python $ROOT/tools/pp_probe.py $sz > $OUT/${TAG}_prof_pp_$sz.log 2>&1
  f=$(find /tmp/pc_pp_$sz -name "*kernel_stats.csv" | head -1); cp $f $OUT/${TAG}_pp_${sz}_kernel_stats.csv; echo "---- place_pending, $sz requests (the product's chunk form)"; show $f
done
rm -rf /tmp/pc_pps
RIO_PART_SHIFT=78 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_pps -o pc -- python $ROOT/tools/pp_probe.py 10000000 > $OUT/${TAG}_prof_pp_small.log 2>&1
f=$(find /tmp/pc_pps -name "*kernel_stats.csv" | head -1); cp $f $OUT/${TAG}_pp_small_kernel_stats.csv; echo "---- place_pending, 10 M requests, 8 192-entry chunks"; show $f
