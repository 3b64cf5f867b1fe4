#!/bin/bash
# round 5, GPU call 3: string layer after the contention fix + window-sorted request path with in-kernel answers
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_object_placement.py -m gpu -q --timeout=600 --tb=short 2>&1 | tail -60 ) > $OUT/r5c_pytest_op.log
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_properties.py -m gpu -q -x --timeout=900 --tb=short 2>&1 | tail -30 ) > $OUT/r5c_pytest_parity.log
timeout 300 python tools/pp_probe.py > $OUT/r5c_place_pending.json 2> $OUT/r5c_place_pending.err
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp_tl -o pp -- python $ROOT/tools/pp_probe.py > /dev/null 2>&1; python $ROOT/tools/pp_timeline.py /tmp/pp_tl/pp_kernel_trace.csv ) > $OUT/r5c_place_pending_timeline.txt 2>&1
bash tools/c_hosts.sh r5c > /dev/null 2>&1
echo "--- pytest op"; cat $OUT/r5c_pytest_op.log
echo "--- pytest parity"; cat $OUT/r5c_pytest_parity.log
echo "--- pp probe"; cut -c1-1500 $OUT/r5c_place_pending.json; tail -3 $OUT/r5c_place_pending.err
echo "--- timeline"; cat $OUT/r5c_place_pending_timeline.txt
echo "--- c hosts"; grep -v '"threads": 4,' $OUT/r5c_c_host_threads.json | cut -c1-235; tail -3 $OUT/r5c_c_host_threads.err
