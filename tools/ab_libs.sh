#!/bin/bash
# Same-box A/B of builds of the library: every gamd_amd/libgamd_hip_<name>.so named in LIBS (loaded through GAMD_LIB) and the
# in-tree one ("new").  ms per step and the neighbour stage of the workloads the neighbour code weighs on; "c2x" = C2 with an
# exact rebuild every step.  Usage (through gpurun): LIBS="old b" bash tools/ab_libs.sh <tag> [workloads ...]
tag=${1:-ab}; shift
ws=${@:-c5 c3 c1 c2x}
out=gpurun_out/$tag; mkdir -p $out
for rep in 1 2; do
for lib in ${LIBS:-old} new; do
  if [ $lib = new ]; then unset GAMD_LIB; else export GAMD_LIB=$PWD/gamd_amd/libgamd_hip_$lib.so; fi
  for w in $ws; do
    extra=""; wl=$w
    if [ $w = c2x ]; then wl=c2; extra="--skin 0"; fi
    python3 bench.py --workload $wl $extra --steps 300 --warmup 30 --no-secondary --no-cpu-baseline --line full 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['config']['timed_region']; n=t.get('neighbour_stage_ms') or {}
st=d.get('stages_ms') or {}
print('$rep $lib $w ms/step %.4f  p50 %.4f  rebuilds %d  nbr reuse %.1f us rebuild %.1f us | conv %.1f us  encoder %.1f us  node interval %.1f us' % (d['ms_per_step'], t['step_ms']['p50'], t['rebuilds_in_timed'], 1e3*n.get('reuse_step', float('nan')), 1e3*n.get('rebuild_step', float('nan')), 1e3*st.get('conv_edge',[float('nan')])[0], 1e3*st.get('edge_encode',[float('nan')])[0], 1e3*st.get('node_mid',[float('nan')])[0]))" | tee -a $out/ab.txt
  done
done
done
