#!/usr/bin/env python3
"""Timing ablations of k_node (profiling build, GAMD_NODE_VARIANT; wrong results for every variant but 0):
    python tools/node_variants.py [workload ...]      (default c2 c5 c1)
Prints the node kernel's interval between two conv layers (live HIP events, as bench.py reports it) per variant."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "production", 1: "no piece loads (agg = 0)", 2: "all weights from one hot KB", 3: "1 + 2", 4: "no GEMMs", 7: "1 + 2 + 4"}
for wl in sys.argv[1:] or ["c2", "c5", "c1"]:
    for v in (0, 1, 2, 3, 4, 7):
        env = dict(os.environ, GAMD_LIB=os.path.join(ROOT, "gamd_amd", "libgamd_hip_prof.so"), GAMD_NODE_VARIANT=str(v))
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--no-cpu-baseline", "--line", "full", "--steps", "100", "--warmup", "10"],
                           env=env, capture_output=True, text=True)
        try:
            d = json.loads(p.stdout.strip().splitlines()[-1])
            k = [x for x in d["roofline"]["kernels"] if x["kernel"].startswith("k_node")][0]
            print(f"{wl} variant {v} ({NAMES[v]:30s}) node interval {k['avg_launch_ms'] * 1e3:6.2f} us   step {d['ms_per_step']:.4f} ms")
        except Exception as exc:
            print(f"{wl} variant {v}: failed ({exc}) {p.stderr[-300:]}")
