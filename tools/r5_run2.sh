#!/bin/bash
# round 5, GPU call 2: the string layer (shadow, flat combining, batch_n, default flags) + bench self-spawn
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_object_placement.py tests/test_snapshot_formats.py -m gpu -q --timeout=600 2>&1 | tail -25 ) > $OUT/r5b_pytest_op.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=600 -k "place_pending or clean" 2>&1 | tail -8 ) > $OUT/r5b_pytest_pp.log
( timeout 900 python -m pytest tests/test_gpu_bench_flow.py -m gpu -q -x --timeout=600 -k "without_a_launcher or two_ranks_as" 2>&1 | tail -8 ) > $OUT/r5b_pytest_flow.log
bash tools/c_hosts.sh r5b > /dev/null 2>&1
timeout 300 bash tools/host_layer_scaling.sh 256 10 2000 > $OUT/r5b_host_layer_stub.json 2> $OUT/r5b_host_layer_stub.err
echo "--- pytest op"; cat $OUT/r5b_pytest_op.log
echo "--- pytest pp"; cat $OUT/r5b_pytest_pp.log
echo "--- pytest flow"; cat $OUT/r5b_pytest_flow.log
echo "--- c hosts"; cut -c1-400 $OUT/r5b_c_host.json; cat $OUT/r5b_c_host_threads.json; tail -3 $OUT/r5b_c_host_threads.err
echo "--- stub"; cat $OUT/r5b_host_layer_stub.json | cut -c1-260; tail -3 $OUT/r5b_host_layer_stub.err
