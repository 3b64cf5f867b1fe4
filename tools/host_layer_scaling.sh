#!/bin/bash
# The string layer's concurrency (interning under the shared table lock, host shadow, flat combining) WITHOUT a GPU: the trait-
# level example linked against the host-memory stub of the dense ABI (tests/stub_rio_gp.cpp) whose batched calls hold the
# "device" for 10 us.  A development aid for machines without a GPU; the numbers that count are tools/c_hosts.sh's on the box.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
g++ -O2 -std=c++17 -pthread -DSTUB_LATENCY_US=${2:-10} -I $ROOT/include -c $ROOT/rio-rs_amd/csrc/gpu_object_placement.cpp -o /tmp/hl_gop.o &&
g++ -O2 -std=c++17 -pthread -DSTUB_LATENCY_US=${2:-10} -I $ROOT/include -c $ROOT/tests/stub_rio_gp.cpp -o /tmp/hl_stub.o &&
gcc -O2 -std=c99 -pthread -I $ROOT/include -c $ROOT/examples/c_host_threads.c -o /tmp/hl_main.o &&
g++ -pthread /tmp/hl_main.o /tmp/hl_gop.o /tmp/hl_stub.o -o /tmp/hl_threads && /tmp/hl_threads 20000 ${3:-2000} ${1:-16}
