#!/bin/bash
# per-kernel durations of the solves in which capacity binds (contended / skew cold table): rocprofv3 --kernel-trace --stats over
# tools/slowpath_workload.py.  Usage: tools/prof_binding.sh <tag> [extra args of slowpath_workload.py after the reps]
TAG=${1:-round6}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for w in contended skew; do
  rm -rf /tmp/pb_$w
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$w -o pb -- python $ROOT/tools/slowpath_workload.py $w 40 "$@" > $OUT/${TAG}_binding_$w.json 2> $OUT/${TAG}_binding_$w.err
  f=$(find /tmp/pb_$w -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/${TAG}_binding_${w}_kernel_stats.csv
  echo "---- $w"; cut -c1-200 $OUT/${TAG}_binding_$w.json
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-90s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
