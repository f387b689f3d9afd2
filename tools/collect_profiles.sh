#!/bin/bash
# Copy the summaries of a tools/gpu_profile_round.sh run (gpurun_out/<tag>/) into profiles/ under the round's names.
# Usage: bash tools/collect_profiles.sh r03a r03_a   (raw logs go to profiles/<round>_raw/)
set -eu
src=gpurun_out/$1; dst=profiles; name=$2
hdr='`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary --steps 50 --warmup 5 --workload <w>` (tools/gpu_profile_round.sh)'
for w in c1 c1_batch c2 c2_batch8 c3 c5 c5b dft; do { echo "# $name — $hdr"; echo; cat $src/trace_$w.md; } > $dst/${name}_trace_$w.md; done
hdr2='`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary --steps 50 --warmup 5 --workload <w> --edge-dtype f16x3` (tools/gpu_profile_round.sh)'
for w in c2 dft; do { echo "# $name — $hdr2"; echo; cat $src/trace_${w}_f16x3.md; } > $dst/${name}_trace_${w}_f16x3.md; done
{ echo "# $name — ${hdr2/f16x3/bf16}"; echo; cat $src/trace_dft_bf16.md; } > $dst/${name}_trace_dft_bf16.md
for w in c2 c5 c2_f16x3; do cp $src/pmc_$w.md $dst/${name}_pmc_$w.md; done
tail -1 $src/bench_default.json > $dst/${name}_bench_default.json
tail -1 $src/bench_f16x3.json > $dst/${name}_bench_f16x3.json
[ -f $src/bench_default_detail.json ] && cp $src/bench_default_detail.json $dst/${name}_bench_default_detail.json
[ -f $src/bench_driver_cmd.json ] && tail -1 $src/bench_driver_cmd.json > $dst/${name}_bench_driver_cmd.json
cp $src/pmc_conv_edge.json $dst/pmc_conv_edge.json
if [ -f $src/gather_hbm.json ]; then cp $src/gather_hbm.json $dst/gather_hbm.json; cp $src/gather_summary.md $dst/${name}_gather_summary.md; fi
raw=$dst/${name%%_*}_raw
mkdir -p $raw
cp $src/conv_variants_sched.log $raw/${name}_conv_variants_sched.log
cp $src/conv_variants_cycles.log $raw/${name}_conv_variants_cycles.log
cp $src/f16x3_marks.log $raw/${name}_f16x3_marks.log
for f in node_variants.log node_marks_c5.log node_marks_c1.log node_marks_10000.log bf16_variants_c5.log bf16_marks.log enc_variants.log bf16_overlap_probe.log mfma_korder_probe.log c5_error.log; do [ -f $src/$f ] && cp $src/$f $raw/${name}_$f; done
true
