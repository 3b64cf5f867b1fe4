#!/bin/bash
# The two C hosts of examples/ against the library of this tree: pipelined solves + churn ticks from C99 (c_host), and the
# trait-shaped single-object calls from 1 / 4 / 16 / 64 / 256 pthreads sharing one provider (c_host_threads: with the host
# shadow and with every call on the device, default collect window and none; with the shadow also through the two hand-off
# entries — every call behind a pool of blocking threads | rio_op_try_* inline, the pool on EAGAIN).  Usage: tools/c_hosts.sh <tag>
TAG=${1:-chost}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
gcc -O2 -std=c99 -I include examples/c_host.c -o /tmp/c_host -L rio-rs_amd -lrio_gp -Wl,-rpath,$ROOT/rio-rs_amd -Wl,-rpath,/opt/rocm/lib -lm || exit 1
gcc -O2 -std=c99 -pthread -I include examples/c_host_threads.c -o /tmp/c_host_threads -L rio-rs_amd -lrio_gp -Wl,-rpath,$ROOT/rio-rs_amd -Wl,-rpath,/opt/rocm/lib -lm || exit 1
timeout 300 /tmp/c_host 10000000 1024 100 200 > $OUT/${TAG}_c_host.json 2> $OUT/${TAG}_c_host.err
# the box's CPU allowance: 256 hardware threads are visible, the container's cgroup grants a quota (cpu.max "<quota us> <period us>",
# 16 CPUs on this pool) — thread counts above it run throttled, and cpu.stat before / after says by how much
throttle() { grep -E 'nr_periods|nr_throttled|throttled_usec' /sys/fs/cgroup/cpu.stat 2>/dev/null | awk '{printf "\"%s\": %s, ", $1, $2}'; }
( echo "{\"host\": \"$(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2 | sed 's/^ //')\", \"hardware_threads\": $(nproc), \"cgroup_cpu_max\": \"$(cat /sys/fs/cgroup/cpu.max 2>/dev/null)\"}"
  echo "{$(throttle)\"when\": \"before\"}"
  timeout 600 /tmp/c_host_threads 20000 2000 256 0
  echo "{$(throttle)\"when\": \"after the default run\"}"
  timeout 600 /tmp/c_host_threads 20000 2000 256 1 | grep '"device"' ) > $OUT/${TAG}_c_host_threads.json 2> $OUT/${TAG}_c_host_threads.err
cat $OUT/${TAG}_c_host.json $OUT/${TAG}_c_host_threads.json
