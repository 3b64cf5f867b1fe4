#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x --timeout=600 --tb=short -k "place_pending or fuzz" 2>&1 | tail -12 ) > $OUT/r5i_pytest.log
timeout 300 python tools/pp_sizes.py 40 4097,8192,65536 > $OUT/r5i_pp_sizes.json 2> $OUT/r5i_pp_sizes.txt
cat $OUT/r5i_pytest.log $OUT/r5i_pp_sizes.txt
