#!/bin/bash
W=${1:-churn}
for v in old new; do
  if [ $v = old ]; then export RIO_GP_LIB=$PWD/ab/librio_gp_old.so; unset RIO_TRACE_ADD_WALKS; else unset RIO_GP_LIB; export RIO_TRACE_ADD_WALKS=1; fi
  echo "== $v"; timeout 100 python tools/cut_trace.py 3 $W 2>&1 | tail -7
done
