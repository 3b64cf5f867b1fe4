#!/bin/bash
# round 5, GPU call 1: the new general request path (tests + sizes) and the trait layer under 1..256 threads (baseline)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
nproc > $OUT/r5a_nproc.txt; grep -c ^processor /proc/cpuinfo >> $OUT/r5a_nproc.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=600 -k "place_pending or reference_port or sql_golden" 2>&1 | tail -15 ) > $OUT/r5a_pytest_pp.log
( timeout 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_properties.py tests/test_gpu_object_placement.py -m gpu -q -x --timeout=600 2>&1 | tail -15 ) > $OUT/r5a_pytest_fuzz.log
timeout 600 python tools/pp_sizes.py 60 > $OUT/r5a_pp_sizes.json 2> $OUT/r5a_pp_sizes.err
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ppm_tl -o ppm -- python $ROOT/tools/pp_mid_timeline.py > /dev/null 2>&1; python $ROOT/tools/pp_mid_timeline.py /tmp/ppm_tl/ppm_kernel_trace.csv ) > $OUT/r5a_pp_mid_timeline.txt 2>&1
gcc -O2 -std=c99 -pthread -I include examples/c_host_threads.c -o /tmp/c_host_threads -L rio-rs_amd -lrio_gp -Wl,-rpath,$ROOT/rio-rs_amd -Wl,-rpath,/opt/rocm/lib -lm && timeout 300 /tmp/c_host_threads 20000 2000 256 > $OUT/r5a_c_host_threads.json 2> $OUT/r5a_c_host_threads.err
echo "--- nproc"; cat $OUT/r5a_nproc.txt
echo "--- pytest pp"; cat $OUT/r5a_pytest_pp.log
echo "--- pytest fuzz"; cat $OUT/r5a_pytest_fuzz.log
echo "--- pp sizes"; cat $OUT/r5a_pp_sizes.err
echo "--- mid timeline"; cat $OUT/r5a_pp_mid_timeline.txt
echo "--- host threads"; cat $OUT/r5a_c_host_threads.json; tail -3 $OUT/r5a_c_host_threads.err
