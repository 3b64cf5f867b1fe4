#!/bin/bash
# A/B of two builds on ONE box, interleaved: ab/librio_gp_old.so (RIO_GP_LIB) against the in-tree library
for i in 1 2 3 4; do for v in old new; do
  if [ $v = old ]; then export RIO_GP_LIB=$PWD/ab/librio_gp_old.so; else unset RIO_GP_LIB; fi
  echo -n "$v churn "; timeout 100 python tools/slowpath_workload.py churn 60 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_tick']*1e3,1))"
done; done
for v in old new; do
  if [ $v = old ]; then export RIO_GP_LIB=$PWD/ab/librio_gp_old.so; else unset RIO_GP_LIB; fi
  for w in contended skew; do echo -n "$v $w "; timeout 100 python tools/slowpath_workload.py $w 40 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_solve']*1e3,1))"; done
done
for v in old new; do
  if [ $v = old ]; then export RIO_GP_LIB=$PWD/ab/librio_gp_old.so; else unset RIO_GP_LIB; fi
  echo "== $v trace"; timeout 100 python tools/cut_trace.py 6 2>&1 | tail -6
done
