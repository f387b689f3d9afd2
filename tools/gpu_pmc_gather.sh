#!/bin/bash
# The neighbour gather where it reaches HBM (round-4 review, item 3): the LJ workload at <n_atoms> atoms (node tables hn / S / D
# of 512 B per atom each: 10^5 atoms = 154 MB, 10^6 atoms = 1.5 GB — beyond the 32 MiB of L2 and, at 10^6, the 256 MiB Infinity
# Cache), conv-layer edge kernel of the given edge dtype.  One live-timed run, one rocprofv3 kernel trace, and THREE separate
# --pmc passes (each with --kernel-trace only, MI355X_MICROARCH.md section HBM): FETCH_SIZE | WRITE_SIZE | TCC hit / miss.
#   bash tools/gpu_pmc_gather.sh <tag> <n_atoms> <f32|bf16|f16x3>
set -u
tag=${1:-gather}; n=${2:-100000}; dt=${3:-f32}
out=gpurun_out/$tag/${n}_$dt
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
S="python3 tools/size_scan.py $n --edge-dtype $dt --steps 3"
$S --json $out/live.json > $out/live.md 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $S > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/fetch -- $S > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- $S > $out/write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/tcc -- $S > $out/tcc.log 2>&1
python3 tools/profile_summary.py gather $out $n $dt > $out/summary.md 2> $out/summary.err
cat $out/summary.md
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete; find $out -name "*agent_info.csv" -delete
