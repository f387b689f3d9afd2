#!/bin/bash
# Round profile on the GPU box: bench lines, rocprofv3 kernel traces and PMC passes (each --pmc pass on its own, with
# --kernel-trace only), summaries written under gpurun_out/$1/.  Usage (through gpurun): bash tools/gpu_profile_round.sh r02
set -u
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
B="python3 bench.py --no-cpu-baseline --no-secondary"
for w in c2 c3 c5 c5b c1 c1_batch c2_batch8 dft; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$w -- $B --steps 50 --warmup 5 --workload $w > $out/trace_$w.log 2>&1
  python3 tools/profile_summary.py stats $out/trace_$w > $out/trace_$w.md
done
# the opt-in split-fp16 edge MLP: C2 and the DFT-water configuration
for w in c2 dft; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_${w}_f16x3 -- $B --steps 50 --warmup 5 --workload $w --edge-dtype f16x3 > $out/trace_${w}_f16x3.log 2>&1
  python3 tools/profile_summary.py stats $out/trace_${w}_f16x3 > $out/trace_${w}_f16x3.md
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_dft_bf16 -- $B --steps 50 --warmup 5 --workload dft --edge-dtype bf16 > $out/trace_dft_bf16.log 2>&1
python3 tools/profile_summary.py stats $out/trace_dft_bf16 > $out/trace_dft_bf16.md
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/pmc_sq_c2_f16x3 -- $B --steps 5 --warmup 2 --workload c2 --edge-dtype f16x3 > $out/pmc_sq_c2_f16x3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_fetch_c2_f16x3 -- $B --steps 5 --warmup 2 --workload c2 --edge-dtype f16x3 > $out/pmc_fetch_c2_f16x3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_write_c2_f16x3 -- $B --steps 5 --warmup 2 --workload c2 --edge-dtype f16x3 > $out/pmc_write_c2_f16x3.log 2>&1
python3 tools/profile_summary.py pmc $out/pmc_sq_c2_f16x3 $out/pmc_fetch_c2_f16x3 $out/pmc_write_c2_f16x3 > $out/pmc_c2_f16x3.md
GAMD_LIB=gamd_amd/libgamd_hip_prof.so GAMD_F16X3_TIME=1 python3 tools/f16x3_marks.py > $out/f16x3_marks.log 2>&1
for w in c2 c5; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/pmc_sq_$w -- $B --steps 5 --warmup 2 --workload $w > $out/pmc_sq_$w.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_fetch_$w -- $B --steps 5 --warmup 2 --workload $w > $out/pmc_fetch_$w.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_write_$w -- $B --steps 5 --warmup 2 --workload $w > $out/pmc_write_$w.log 2>&1
  python3 tools/profile_summary.py pmc $out/pmc_sq_$w $out/pmc_fetch_$w $out/pmc_write_$w > $out/pmc_$w.md
done
python3 tools/profile_summary.py pmcjson $out/pmc_fetch_c2 $out/pmc_write_c2 'k_conv_edge<' $out/pmc_conv_edge.json $out/trace_c2 > /dev/null
# the neighbour gather where it reaches HBM (10^5 / 10^6 atoms): live run + trace + three separate --pmc passes each
for cfg in "100000 bf16" "1000000 bf16" "1000000 f32"; do set -- $cfg; bash tools/gpu_pmc_gather.sh $tag/gather $1 $2 > /dev/null 2>&1; done
python3 tools/profile_summary.py gatherjson $out/gather_hbm.json $out/gather/100000_bf16 $out/gather/1000000_bf16 $out/gather/1000000_f32 > /dev/null
cat $out/gather/*/summary.md > $out/gather_summary.md
# the bf16 conv kernel: timing ablations / priority variants and segment marks (profiling build)
python3 tools/bf16_variants.py 0 2048 1 2 4 256 384 263 903 > $out/bf16_variants_c5.log 2>&1
GAMD_LIB=gamd_amd/libgamd_hip_prof.so GAMD_BF16_VARIANT=64 python3 tools/bf16_marks.py > $out/bf16_marks.log 2>&1
python3 tools/enc_variants.py 160 0 32 96 224 > $out/enc_variants.log 2>&1
./gamd_amd/csrc/probes/bf16_overlap_probe > $out/bf16_overlap_probe.log 2>&1
./gamd_amd/csrc/probes/mfma_korder_probe > $out/mfma_korder_probe.log 2>&1
python3 tools/c5_error.py > $out/c5_error.log 2>&1
# conv-layer kernel: scheduling variants A/B and the s_memtime marks (profiling build)
python3 tools/conv_variants.py 3592 0 8 520 1544 3848 > $out/conv_variants_sched.log 2>&1
python3 tools/conv_variants.py 3593 --cycles > $out/conv_variants_cycles.log 2>&1
# the bench records LAST, with this run's counter records in place: bench.py reports roofline.traffic / neighbour_gather_hbm only
# for the kernel sources they were measured on (hash), and on this box profiles/ still holds the previous sources' records
cp $out/pmc_conv_edge.json profiles/pmc_conv_edge.json; cp $out/gather_hbm.json profiles/gather_hbm.json
# (the ONE stdout line is the compact contract record; the full record — per-step distributions, per-kernel list, all twelve
#  secondary workloads — goes to --detail)
python3 bench.py --secondary full --detail $out/bench_default_detail.json > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail $out/bench_driver_cmd_detail.json > $out/bench_driver_cmd.json 2>> $out/bench_default.err   # the driver's command
python3 bench.py --no-cpu-baseline --no-secondary --edge-dtype f16x3 --detail - > $out/bench_f16x3.json 2>> $out/bench_default.err
# keep the merge-back small: only the summaries and the per-kernel stats csv
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete; find $out -name "*agent_info.csv" -delete
ls -la $out | head -50
