#!/bin/bash
# Same-run A/B of two builds of the product library on one workload of bench.py: tools/ab_lib.sh <other.so> <tag> [bench args...]
# alternates <other.so> (RIO_GP_LIB) and the tree's librio_gp.so three times each and keeps every JSON line.
OTHER=$1; TAG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
: > $OUT/${TAG}_ab.jsonl
for r in 1 2 3; do
  for which in other tree; do
    if [ $which = other ]; then export RIO_GP_LIB=$ROOT/$OTHER; else unset RIO_GP_LIB; fi
    timeout 300 python bench.py "$@" 2>/dev/null | tail -1 | sed "s/^{/{\"ab\": \"$which\", \"round\": $r, /" >> $OUT/${TAG}_ab.jsonl
  done
done
python - <<PY
import json
for l in open("$OUT/${TAG}_ab.jsonl"):
    d = json.loads(l)
    c5 = d.get("config5_churn") or {}
    print(d["ab"], d["round"], "ms_per_step", d.get("ms_per_step"), "c5 pipelined/sync ms",
          (c5.get("pipelined") or {}).get("ms_per_tick"), (c5.get("synchronous") or {}).get("ms_per_tick"))
PY
