#!/usr/bin/env python3
"""Step time of the LJ workload (rho* = 0.5, cutoff 3 sigma, skin reuse, BAOAB on device) over the system size:

    python tools/size_scan.py [N ...] [--edge-dtype f32|bf16|f16x3] [--steps K] [--json out.json] [--widths enc,hidden,emb]

GPU box only; the C2 line of bench.py is the N = 10 000 point.  With --steps K it times exactly K steps after 2 warm-up
steps (the form tools/gpu_pmc_gather.sh profiles under rocprofv3: at 10^5 - 10^6 atoms the node tables the conv-layer edge
kernel gathers from — hn, S, D: 512 B per atom each — no longer fit the 8 x 4 MiB L2s / the 256 MiB Infinity Cache)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamd_amd import workloads                                           # noqa: E402
from gamd_amd.engine import GamdForce                                    # noqa: E402
from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("sizes", nargs="*", type=int)
ap.add_argument("--edge-dtype", default="f32", choices=["f32", "bf16", "f16x3"])
ap.add_argument("--steps", type=int, default=0)
ap.add_argument("--json", default="")
ap.add_argument("--widths", default="128,128,128", help="encoding_size,hidden_dim,edge_embedding_dim of the LJ model")
args = ap.parse_args()
sizes = args.sizes or [1000, 2000, 5000, 10000, 20000, 50000, 100000, 200000]
enc_w, hid_w, emb_w = (int(v) for v in args.widths.split(","))
sd = make_state_dict(ModelConfig(kind="lj", encoding_size=enc_w, hidden_dim=hid_w, edge_embedding_dim=emb_w), 0, 7.0, 2.2)
rc = 3.0 * workloads.LJ_SIGMA
rows = []
print(f"edge dtype {args.edge_dtype}, widths {enc_w} / {hid_w} / {emb_w}\n")
print("| atoms | edges | ms / step | atom-steps/s | conv kernel ms / launch | GB device memory | encoder ms | node interval ms | 32-edge tiles per wave slot (2 048) |")
print("|---|---|---|---|---|---|---|---|---|")
for n in sizes:
    pos, box = workloads.lj_box(n)
    eng = GamdForce(sd, n, box, rc, scaler=SHIPPED_SCALERS["lj"], neighbor_skin=rc / 6, edge_dtype=args.edge_dtype)
    x = torch.from_numpy(pos).float().cuda()
    v = torch.from_numpy(workloads.maxwell_boltzmann(n, temperature_k=100.0, seed=1)).float().cuda()
    f = eng.forward(x, denormalize=True).clone()
    steps = args.steps or (100 if n <= 20000 else 30)
    warm = 2 if args.steps else 10
    eng.md_run(x, v, f, warm, temperature_k=100.0)
    torch.cuda.synchronize()
    eng.timing_enable(True)
    t0 = time.perf_counter()
    eng.md_run(x, v, f, steps, temperature_k=100.0, first_step=warm)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    conv_ms, conv_n = eng.timing_read()
    stages = eng.timing_read_stages()
    eng.timing_enable(False)
    free, total = torch.cuda.mem_get_info()
    e = eng.counts()[0]
    enc_ms, node_ms = (stages[k][0] / max(stages[k][1], 1) for k in ("edge_encode", "node_mid"))
    print(f"| {n} | {e} | {dt * 1e3:.3f} | {n / dt:.3e} | {conv_ms / max(conv_n, 1):.4f} | {(total - free) / 2**30:.2f} | {enc_ms:.4f} | {node_ms:.4f} | {e / 32 / 2048:.2f} |")
    rows.append({"atoms": n, "edges": e, "ms_per_step": dt * 1e3, "atom_steps_per_s": n / dt,
                 "conv_ms_per_launch": conv_ms / max(conv_n, 1), "conv_launches": conv_n, "edge_dtype": args.edge_dtype,
                 "device_GB": (total - free) / 2 ** 30})
    assert torch.isfinite(x).all() and eng.last_status in (0, 1)
    eng.close()
    del eng, x, v, f
    torch.cuda.empty_cache()
if args.json:
    json.dump(rows, open(args.json, "w"), indent=1)
