#!/usr/bin/env python3
"""s_memtime breakdown of k_conv_edge_f16x3 at C2 (profiling build): GAMD_LIB=gamd_amd/libgamd_hip_prof.so GAMD_F16X3_TIME=1"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamd_amd.engine import GamdForce                                   # noqa: E402
from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS  # noqa: E402
from gamd_amd import workloads as wk                                      # noqa: E402

SEG = ["phase 1 (GEMM, SiLU/split, DMA issue)", "barrier 1", "phase 2 (+ S/D quads)", "barrier 2", "idx loads + phase 3 (+ hn rows 0-7)",
       "barrier 3", "phase 4 (GEMM, message, segment sum, next e)", "barrier 4", "piece stores"]
pos, box = wk.lj_box(10000)
sd = make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2)
eng = GamdForce(sd, 10000, box, 10.2, scaler=SHIPPED_SCALERS["lj"], edge_dtype="f16x3")
x = torch.from_numpy(pos).float().cuda()
for _ in range(3):
    eng.forward(x, inplace=True)
t = eng._dbg(5, (256, 8, 16), np.int64).astype(np.float64)            # 8 waves per workgroup
tiles = t[:, :, 15]
tot = t[:, :, :9].sum(-1)
print(f"E = {eng.counts()[0]}, tiles per wave {tiles.mean():.2f}; ticks per tile {tot.sum() / tiles.sum():.0f}")
for i, nm in enumerate(SEG):
    a = t[:, :, i].sum() / tiles.sum()
    print(f"  {nm:58s} {a:8.0f} ticks/tile  {100 * a / (tot.sum() / tiles.sum()):5.1f} %")
