#!/usr/bin/env python3
"""Error of the bf16 edge MLP (128 / 128 / 128: conv_edge_bf16.hip, fp16 node tables, log2 e folded into the weights) against the
CPU oracle over random models and boxes: LJ and water, LayerNorm and BatchNorm, 1-5 conv layers, with and without the bond
feature, 1-3 boxes per handle.  GPU box only.
    python tools/bf16_error_sweep.py [n_cases] [GAMD_LIB=... for another build]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, d))
import gamd_oracle as orc                                                  # noqa: E402  (checker only)
from helpers import rel_err                                                # noqa: E402
from gamd_amd import workloads                                             # noqa: E402
from gamd_amd.engine import GamdForce                                      # noqa: E402
from gamd_amd.weights import ModelConfig, make_state_dict                  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(6006)
errs = []
for i in range(n_cases):
    kind = ["lj", "water"][i % 2]
    cfg = ModelConfig(kind=kind, conv_layer=int(rng.integers(1, 6)), use_bond=kind == "water" and rng.random() < 0.6,
                      use_layer_norm=rng.random() < 0.7)
    seed = int(rng.integers(0, 10000))
    sd = make_state_dict(cfg, seed, 2.9, 1.1)
    if kind == "lj":
        pos, box = workloads.lj_box(int(rng.integers(100, 1500)), seed=seed)
        species = bonds = feat = None
        rc = 7.5
    else:
        pos, box, species, bonds = workloads.water_box(int(rng.integers(40, 500)), seed=seed)
        feat, rc = torch.from_numpy(species.astype(np.float32)).view(-1, 1), 4.2
        if not cfg.use_bond:
            bonds = None
    n = pos.shape[0]
    eng = GamdForce(sd, n, box, rc, bond=bonds, edge_dtype="bf16")
    p = torch.remainder(torch.from_numpy(pos).float(), float(box))
    out = eng.forward(p, species=species).cpu().numpy()
    edges = orc.neighbor_edges(p, box, rc, "jaxmd")
    ref = orc.forward(sd, p, edges, box, feat=feat, bond=bonds).numpy()
    e = rel_err(out, ref)
    errs.append(e)
    print(f"{i:3d} {kind:5s} L={cfg.conv_layer} {'LN' if cfg.use_layer_norm else 'BN'} bond={int(cfg.use_bond)} n={n:5d} E={edges.shape[1]:7d} err {e:.3e}")
    eng.close()
errs = np.array(errs)
print(f"bf16 vs oracle over {n_cases} random models: median {np.median(errs):.2e}  p90 {np.percentile(errs, 90):.2e}  max {errs.max():.2e}  (restated tolerance 1e-2)")
