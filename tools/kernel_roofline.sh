#!/bin/bash
# Per-kernel roofline records: for every phase of tools/roofline_workload.py one rocprofv3
# kernel-trace pass and the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, --kernel-trace only), then
# tools/kernel_roofline.py -> gpurun_out/<tag>_kernel_roofline.json.  Usage: tools/kernel_roofline.sh <tag>
TAG=${1:-round5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_kroof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RIO_KROOF_DIR=$OUT
for ph in probes fast churn churn_noinc churn_unpacked contended contended_packed crud crud_plain lookup_seq clean1 pp pp10 pp_mid pp_small pp_256; do
  timeout 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/${ph}_trace -o t -- python $ROOT/tools/roofline_workload.py $ph > $OUT/${ph}_trace.log 2>&1
  timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${ph}_fetch -o f -- python $ROOT/tools/roofline_workload.py $ph 4 > $OUT/${ph}_fetch.log 2>&1
  timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${ph}_write -o w -- python $ROOT/tools/roofline_workload.py $ph 4 > $OUT/${ph}_write.log 2>&1
done
cd $ROOT
python tools/kernel_roofline.py $OUT $ROOT/gpurun_out/${TAG}_kernel_roofline.json > $ROOT/gpurun_out/${TAG}_kernel_roofline.txt 2>&1
cat $ROOT/gpurun_out/${TAG}_kernel_roofline.txt
# keep the evidence small: per-kernel stats survive in the JSON, the raw traces do not travel back
find $OUT -name "*.csv" -size +2M -delete
