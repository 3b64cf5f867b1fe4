#!/bin/bash
# rocprofv3 kernel trace of the config-5 churn stream -> per-kernel stats + per-tick timeline (tools/tick_timeline.py)
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_prof_churn
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_churn -o churn -- python $ROOT/tools/slowpath_workload.py churn 20 > $OUT/${TAG}_prof_churn.json 2> $OUT/${TAG}_prof_churn.err
cd $ROOT
python tools/tick_timeline.py $OUT/${TAG}_prof_churn/churn_kernel_trace.csv > $OUT/${TAG}_churn_timeline.txt
cat $OUT/${TAG}_churn_timeline.txt
