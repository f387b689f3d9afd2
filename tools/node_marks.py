#!/usr/bin/env python3
"""s_memtime marks of k_node's mode-1 launches (post(l-1) + pre(l)) on the C2 workload, profiling build with
GAMD_NODE_TIME=1: average ticks per segment over the first 512 workgroups, and the spread of workgroup start / end times.
GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GAMD_LIB", os.path.join(ROOT, "gamd_amd", "libgamd_hip_prof.so"))
os.environ["GAMD_NODE_TIME"] = "1"
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gamd_amd.engine import GamdForce
from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS
from gamd_amd.workloads import lj_box

arg = sys.argv[1] if len(sys.argv) > 1 else "10000"
species = None
if arg in ("c5", "c3"):
    # the water workloads of bench.py: C5 = 2 000 molecules, bf16 edge MLP (the node GEMMs then run in split-fp16, k_node<0, true>)
    from gamd_amd.workloads import water_box
    nmol, dens, scal, seed, dtype = {"c3": (1390, 258.0, "tip3p", 2345, "f32"), "c5": (2000, 251.0, "tip4p", 3456, "bf16")}[arg]
    pos, box, species, bonds = water_box(nmol, mol_per_20A3=dens, seed=seed, jitter=0.0, wrap=False)
    n = pos.shape[0]
    sd = make_state_dict(ModelConfig(kind="water", use_bond=True), 3, 2.9, 1.1)
    eng = GamdForce(sd, n, box, 4.2, bond=bonds, scaler=SHIPPED_SCALERS[scal], edge_dtype=dtype)
elif arg == "c1":
    g = np.load(os.path.join(ROOT, "tests", "golden", "lj258_seed0.npz"))
    pos, box, n = np.mod(g["pos"], 27.27), 27.27, 258
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2)
    eng = GamdForce(sd, n, box, 7.5, scaler=SHIPPED_SCALERS["lj"])
else:
    n = int(arg)
    pos, box = lj_box(n)
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2)
    eng = GamdForce(sd, n, box, 3.0 * 3.4, scaler=SHIPPED_SCALERS["lj"])
p = torch.from_numpy(pos).float().cuda()
for _ in range(3):
    eng.forward(p, species=species, inplace=True)
torch.cuda.synchronize()
t = eng._dbg(5, (256 * 8, 16), np.int64).reshape(512, 4, 16).astype(np.float64)
nwg = min(512, (n + 15) // 16)
t = t[:nwg]
names = ["pieces summed", "exchange 1", "GEMM phi_edge", "SiLU + exchange 2", "GEMM phi + residual", "LayerNorm + exchange 3",
         "GEMM S", "GEMM D", "GEMM P"]
seg = np.diff(t[:, :, :10], axis=2)
print(f"{nwg} workgroups sampled; ticks per segment (mean over waves / max over waves), total {seg.sum(2).mean():.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:24s} {seg[:, :, i].mean():8.0f} {seg[:, :, i].max():8.0f}")
t0 = t[:, :, 0].min()
print(f"workgroup start spread: {t[:, 0, 0].min() - t0:.0f} .. {t[:, 0, 0].max() - t0:.0f} ticks; end: {t[:, 0, 9].min() - t0:.0f} .. {t[:, 0, 9].max() - t0:.0f}")
