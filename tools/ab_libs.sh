#!/bin/bash
# same-box A/B of two builds of the library (GAMD_LIB): ms per step of the secondaries that the neighbour stage weighs on
out=gpurun_out/r05o; mkdir -p $out
for rep in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export GAMD_LIB=$PWD/gamd_amd/libgamd_hip_old.so; else unset GAMD_LIB; fi
  for w in c5 c3 c1; do
    python3 bench.py --workload $w --steps 300 --warmup 30 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['config']['timed_region']
print('$rep $lib $w ms/step %.4f  p50 %.4f  rebuilds %d  nbr reuse %.1f us rebuild %.1f us' % (d['ms_per_step'], t['step_ms']['p50'], t['rebuilds_in_timed'], 1e3*t['neighbour_stage_ms']['reuse_step'], 1e3*t['neighbour_stage_ms']['rebuild_step']))" | tee -a $out/ab.txt
  done
done
done
