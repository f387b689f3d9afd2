#!/usr/bin/env python3
"""s_memtime breakdown of k_conv_edge_bf16 at C5 (profiling build, ABL bit 64): where a wave's time goes inside a tile.
s_memtime ticks are not core cycles on this part: read the columns as proportions.
    GAMD_LIB=gamd_amd/libgamd_hip_prof.so GAMD_BF16_VARIANT=64 python tools/bf16_marks.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamd_amd.engine import GamdForce                                   # noqa: E402
from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS  # noqa: E402
from gamd_amd import workloads as wk                                      # noqa: E402

SEG = ["bias init + wait for e", "GEMM 1", "S / D gather issue", "SiLU 1 + pack", "S + D add (waits for the gathers)", "GEMM 2",
       "SiLU 2 + pack", "hn gather issue", "bias + GEMM 3", "SiLU 3 + pack", "chunk metadata + b4 init", "GEMM 4",
       "message + segment sum (waits for hn)", "piece stores (+ wait for next e, idx)"]
pos, box, species, bonds = wk.water_box(2000, mol_per_20A3=251.0, seed=3456, jitter=0.0, wrap=False)
sd = make_state_dict(ModelConfig(kind="water", use_bond=True), 3, 2.9, 1.1)
eng = GamdForce(sd, pos.shape[0], box, 4.2, bond=bonds, scaler=SHIPPED_SCALERS["tip4p"], edge_dtype="bf16")
x = torch.from_numpy(pos).float().cuda()
for _ in range(3):
    eng.forward(x, species=species, inplace=True)
t = eng._dbg(5, (256, 8, 16), np.int64).astype(np.float64)
tiles = t[:, :, 15]
live = tiles > 0
tot = t[:, :, :14].sum(-1)
print(f"E = {eng.counts()[0]}, tiles per wave: mean {tiles[live].mean():.2f} max {tiles.max():.0f}; ticks per tile {tot[live].sum() / tiles[live].sum():.0f}")
for i, nm in enumerate(SEG):
    a = t[:, :, i][live].sum() / tiles[live].sum()
    lo, hi = t[:, :4, i][live[:, :4]].sum() / tiles[:, :4][live[:, :4]].sum(), t[:, 4:, i][live[:, 4:]].sum() / tiles[:, 4:][live[:, 4:]].sum()
    print(f"  {nm:42s} {a:8.0f} ticks/tile  {100 * a / (tot[live].sum() / tiles[live].sum()):5.1f} %   (waves 0-3 {lo:.0f}, 4-7 {hi:.0f})")
