#!/usr/bin/env python3
"""Config 5 (bf16 edge MLP): achieved error against the fp32 path and the CPU oracle, same inputs as
tests/test_gpu_parity.py::test_c5_bf16_edge_mlp_against_fp32_path_and_oracle.  GPU box only.
    python tools/c5_error.py [n_molecules]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gamd_oracle as orc                                                  # noqa: E402  (checker only)
from helpers import rel_err, per_atom_err                                  # noqa: E402
from gamd_amd import workloads                                             # noqa: E402
from gamd_amd.engine import GamdForce                                      # noqa: E402
from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS  # noqa: E402

nmol = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
pos, box, species, bonds = workloads.water_box(nmol, mol_per_20A3=251.0, seed=3456)
sd = make_state_dict(ModelConfig(kind="water", use_bond=True), 3, 2.9, 1.1)
p = torch.from_numpy(pos).float()
kw = dict(bond=bonds, scaler=SHIPPED_SCALERS["tip4p"])
e32 = GamdForce(sd, pos.shape[0], box, 4.2, **kw)
e16 = GamdForce(sd, pos.shape[0], box, 4.2, edge_dtype="bf16", **kw)
a = e32.forward(p, species=species).cpu().numpy().copy()
b = e16.forward(p, species=species).cpu().numpy().copy()
edges = torch.from_numpy(e32.debug_edges()).long()
same = np.array_equal(np.sort(e16.debug_edges(), axis=1), np.sort(e32.debug_edges(), axis=1)) or \
    set(map(tuple, e16.debug_edges().T)) == set(map(tuple, e32.debug_edges().T))
ref = orc.forward(sd, p, edges, box, feat=torch.from_numpy(species.astype(np.float32)).view(-1, 1), bond=bonds).numpy()
med, p99, worst, cnt = per_atom_err(b, ref)
print(f"c5 ({pos.shape[0]} atoms, {edges.shape[1]} edges): bf16 vs fp32 path {rel_err(b, a):.3e}, bf16 vs oracle {rel_err(b, ref):.3e}, "
      f"fp32 vs oracle {rel_err(a, ref):.3e}; per atom |df|/|f| median {med:.2e} p99 {p99:.2e} max {worst:.2e}; same edge set: {same}")
