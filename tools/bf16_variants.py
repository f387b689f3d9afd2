#!/usr/bin/env python3
"""Timing ablations of k_conv_edge_bf16 at C5 (profiling build: GAMD_LIB=gamd_amd/libgamd_hip_prof.so).  Each variant runs in
its own process (the variant is read once); results are wrong by construction for every variant but 0.
    python tools/bf16_variants.py 0 1 2 4 8 16 ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "production", 1: "SiLU -> x/2", 2: "gathers from one hot row", 4: "no piece stores", 8: "no LDS weight fill",
         16: "no GEMMs (SiLU dead-code eliminated)", 128: "GEMMs without LDS weight reads", 256: "GEMMs without MFMAs (all live)",
         257: "256 + SiLU -> x/2", 263: "256+1+2+4", 9: "no fill + SiLU -> x/2", 384: "GEMMs without MFMAs and LDS reads",
         391: "384+1+2+4", 512: "e from one hot tile", 775: "263 + 512", 903: "everything off", 1024: "prio 1 around the GEMMs",
         2048: "WITHOUT prio in the SiLU blocks", 4096: "static prio for waves 4-7", 8192: "prio everywhere but the GEMMs",
         16384: "S + D block at prio too", 65536: "message block at prio too", 81920: "both", 32768: "SiLU fused into the consuming GEMM", 34816: "fused, no priority", 163840: "fused, whole GEMM at prio",
         32: "S / D gathers from one hot row", 3: "1+2", 7: "1+2+4", 23: "1+2+4+16", 31: "all"}
WL = os.environ.get("GAMD_VARIANT_WORKLOAD", "c5").split(",")          # e.g. "c5" or "c2,--edge-dtype,bf16"
for v in [int(x) for x in sys.argv[1:]] or [0]:
    env = dict(os.environ, GAMD_LIB=os.path.join(ROOT, "gamd_amd", "libgamd_hip_prof.so"), GAMD_BF16_VARIANT=str(v))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", *WL, "--no-secondary", "--no-cpu-baseline", "--line", "full",
                        "--steps", "100", "--warmup", "10"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        print(f"variant {v:2d} ({NAMES.get(v, '?'):28s}) conv launch {d['roofline']['avg_launch_ms'] * 1e3:7.2f} us   step {d['ms_per_step']:.4f} ms")
    except Exception as exc:
        print(f"variant {v}: failed ({exc}) {p.stderr[-300:]}")
