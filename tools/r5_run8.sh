#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
( timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x --timeout=600 --tb=short 2>&1 | tail -12 ) > $OUT/r5j_pytest.log
timeout 500 python tests/test_gpu_fuzz.py 360 500000 > $OUT/r5j_fuzz_long.json 2> $OUT/r5j_fuzz_long.err
cat $OUT/r5j_pytest.log; cat $OUT/r5j_fuzz_long.json; tail -5 $OUT/r5j_fuzz_long.err
