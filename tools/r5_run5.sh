#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT

g++ -O2 -std=c++17 -pthread -DSTUB_LATENCY_US=10 -I include -c rio-rs_amd/csrc/gpu_object_placement.cpp -o /tmp/hl_gop.o &&
g++ -O2 -std=c++17 -pthread -DSTUB_LATENCY_US=10 -I include -c tests/stub_rio_gp.cpp -o /tmp/hl_stub.o &&
gcc -O2 -std=c99 -pthread -I include -c examples/c_host_threads.c -o /tmp/hl_main.o &&
g++ -pthread /tmp/hl_main.o /tmp/hl_gop.o /tmp/hl_stub.o -o /tmp/hl_threads
for c in 0 1 4000; do /tmp/hl_threads 20000 2000 256 $c 2>&1 | grep -A1 '"device"' | grep -v '^--' | cut -c1-215; done > $OUT/r5e_stub.txt 2>&1
bash tools/c_hosts.sh r5e > /dev/null 2>&1
( timeout 600 python -m pytest tests/test_gpu_object_placement.py -m gpu -q --timeout=600 --tb=short 2>&1 | grep -v "^combiner" | tail -15 ) > $OUT/r5e_pytest_op.log
cat $OUT/r5e_stub.txt; grep -v '"threads": 4,' $OUT/r5e_c_host_threads.json | cut -c1-235; cat $OUT/r5e_pytest_op.log
