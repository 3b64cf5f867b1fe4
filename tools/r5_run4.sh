#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
export RIO_OP_DBG=1
g++ -O2 -std=c++17 -pthread -DSTUB_LATENCY_US=10 -I include -c rio-rs_amd/csrc/gpu_object_placement.cpp -o /tmp/hl_gop.o &&
g++ -O2 -std=c++17 -pthread -DSTUB_LATENCY_US=10 -I include -c tests/stub_rio_gp.cpp -o /tmp/hl_stub.o &&
gcc -O2 -std=c99 -pthread -I include -c examples/c_host_threads.c -o /tmp/hl_main.o &&
g++ -pthread /tmp/hl_main.o /tmp/hl_gop.o /tmp/hl_stub.o -o /tmp/hl_threads
for c in 0 1; do /tmp/hl_threads 20000 2000 256 $c 2>&1 | grep -A1 '"device"' | grep -v '^--' | grep 'lookup\|combiner' | cut -c1-200; done > $OUT/r5d_stub.txt 2>&1
timeout 300 python tools/pp_probe.py > $OUT/r5d_place_pending.json 2> $OUT/r5d_place_pending.err
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp_tl -o pp -- python $ROOT/tools/pp_probe.py > /dev/null 2>&1; python $ROOT/tools/pp_timeline.py /tmp/pp_tl/pp_kernel_trace.csv | head -30 ) > $OUT/r5d_place_pending_timeline.txt 2>&1
( timeout 600 python -m pytest tests/test_gpu_object_placement.py -m gpu -q --timeout=600 --tb=short 2>&1 | tail -15 ) > $OUT/r5d_pytest_op.log
cat $OUT/r5d_stub.txt; cut -c1-600 $OUT/r5d_place_pending.json; cat $OUT/r5d_place_pending_timeline.txt; cat $OUT/r5d_pytest_op.log
