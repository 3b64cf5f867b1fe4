#!/bin/bash
# A short gpurun pass while iterating on the fix-up kernels: the parity tests that cover them, the config-5 bench line and
# the phase traces.  Usage: tools/quick_pass.sh <tag>
TAG=${1:-quick}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full_size.py tests/test_gpu_sharded.py -m gpu -q -x --timeout=600 2>&1 | tail -15 ) > $OUT/${TAG}_pytest.log
timeout 600 python bench.py --workload c5 --steps 100 --warmup 10 > $OUT/${TAG}_bench_c5.json 2> $OUT/${TAG}_bench_c5.err
timeout 300 python tools/fixup_trace.py 6 > $OUT/${TAG}_fixup_trace.json 2> $OUT/${TAG}_fixup_trace.err
timeout 120 python tools/sync_probe.py > $OUT/${TAG}_sync_probe.json 2>&1
bash tools/prof_churn.sh ${TAG} > /dev/null 2>&1
echo "---- pytest"; cat $OUT/${TAG}_pytest.log
echo "---- bench c5"; cut -c1-1800 $OUT/${TAG}_bench_c5.json; tail -3 $OUT/${TAG}_bench_c5.err
echo "---- timeline"; tail -14 $OUT/${TAG}_churn_timeline.txt
echo "---- sync probe"; cat $OUT/${TAG}_sync_probe.json
echo "---- traces"; python - <<PY
import json
d=json.load(open("$OUT/${TAG}_fixup_trace.json"))
for k,v in d.items():
    if k!="last_tick": print(k, json.dumps(v))
PY
