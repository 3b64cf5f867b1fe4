#!/bin/bash
# A/B of two builds on ONE box: synchronous call latencies (fast-path tick, churn tick from the C host)
gcc -O2 -std=c99 -I include examples/c_host.c -o examples/c_host -L rio-rs_amd -lrio_gp -Wl,-rpath,/opt/rocm/lib -lm
for i in 1 2; do for v in old new; do
  if [ $v = old ]; then mkdir -p /tmp/lo && cp ab/librio_gp_old.so /tmp/lo/librio_gp.so && export LD_LIBRARY_PATH=/tmp/lo; else export LD_LIBRARY_PATH=$PWD/rio-rs_amd; fi
  echo -n "$v c_host "; ./examples/c_host 10000000 1024 100 200 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('solve %.1f us  churn tick %.1f us' % (d['fast_path']['us_per_solve'], d['churn']['us_per_tick']))"
done; done
unset LD_LIBRARY_PATH
for v in old new; do
  if [ $v = old ]; then export RIO_GP_LIB=$PWD/ab/librio_gp_old.so; else unset RIO_GP_LIB; fi
  echo -n "$v sync fast tick (python, 300 calls): "; python - <<'PY'
import sys, time
sys.path[:0] = ["rio-rs_amd", "oracle"]
import rio_gp, synth
cfg = synth.config("c3w")
g = rio_gp.GpuPlacement(cfg["n"], cfg["m"]); g.set_nodes(cfg["cap"], cfg["alive"]); g.set_objects(cfg["n"], cfg["load"], cfg["aff"]); g.set_assign(cfg["cur"])
for _ in range(20): g.tick()
t0 = time.perf_counter()
for _ in range(300): g.tick()
print("%.1f us" % ((time.perf_counter() - t0) / 300 * 1e6))
PY
done
