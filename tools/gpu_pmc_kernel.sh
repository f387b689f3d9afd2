#!/bin/bash
# SQ stall-accounting counters of one workload's kernels (one rocprofv3 --pmc pass per group, --kernel-trace only):
#   bash tools/gpu_pmc_kernel.sh <tag> <workload> [extra bench args]
# WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (MI355X_MICROARCH.md, rocprofv3 PMC slots)
set -u
tag=${1:-pmc}; wl=${2:-c5}; shift 2
out=gpurun_out/$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
B="python3 bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 2 --workload $wl $*"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $out/a -- $B > $out/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $out/b -- $B > $out/b.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for grp in ("a", "b"):
    per = collections.defaultdict(float)                       # (kernel, dispatch, counter) -> sum over the counter's instances
    for f in glob.glob(f"{out}/{grp}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            per[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
    order = sorted(acc, key=lambda k: -sum(sum(v) for v in acc[k].values()))
    for k in order[:8]:
        n = max(len(v) for v in acc[k].values())
        print(grp, k[:40].ljust(40), f"launches {n:3d}", " ".join(f"{c.replace('SQ_', '')}={sum(v) / len(v):.4g}" for c, v in sorted(acc[k].items())))
PY
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete; find $out -name "*agent_info.csv" -delete
