#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x --timeout=600 --tb=short -k "place_pending or fuzz" 2>&1 | tail -8 ) > $OUT/r5h_pytest.log
timeout 300 python tools/pp_probe.py > $OUT/r5h_place_pending.json 2> $OUT/r5h_place_pending.err
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp_tl -o pp -- python $ROOT/tools/pp_probe.py > /dev/null 2>&1; python $ROOT/tools/pp_timeline.py /tmp/pp_tl/pp_kernel_trace.csv 2>/dev/null | head -28 ) > $OUT/r5h_place_pending_timeline.txt 2>&1
cat $OUT/r5h_pytest.log; cut -c1-700 $OUT/r5h_place_pending.json; cat $OUT/r5h_place_pending_timeline.txt
