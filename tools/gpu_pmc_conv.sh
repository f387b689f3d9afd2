#!/bin/bash
# Only the two PMC passes behind profiles/pmc_conv_edge.json (HBM bytes per k_conv_edge launch, stamped with the hash of the
# kernel sources).  Run through gpurun after any change to the sources bench.py hashes:  bash tools/gpu_pmc_conv.sh r02x
set -u
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
B="python3 bench.py --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_fetch_c2 -- $B --steps 5 --warmup 2 --workload c2 > $out/pmc_fetch_c2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_write_c2 -- $B --steps 5 --warmup 2 --workload c2 > $out/pmc_write_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c2 -- $B --steps 50 --warmup 5 --workload c2 > $out/trace_c2.log 2>&1
python3 tools/profile_summary.py stats $out/trace_c2 > $out/trace_c2.md
python3 tools/profile_summary.py pmcjson $out/pmc_fetch_c2 $out/pmc_write_c2 'k_conv_edge<' $out/pmc_conv_edge.json $out/trace_c2
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete; find $out -name "*agent_info.csv" -delete
