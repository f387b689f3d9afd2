#!/usr/bin/env python3
"""Random feature-matrix sweep on the GPU box (test infrastructure: it checks against the oracle, so it lives under tests/; not
collected by pytest): 6 x 16 seeded combinations of widths (8..256 / 8..128 / 8..256) and 2 x 16 with hidden_dim 129..256 (fp32),
1-5 layers, the three model flavours, LayerNorm or BatchNorm, update_edge, expand_edge on / off, bond feature, skin or exact
neighbour mode, 1-3 boxes per handle and the three edge dtypes -- forces against the oracle (1e-5; bf16: 1e-2).  The fixed
sample in tests/test_gpu_round4.py::test_sampled_feature_matrix_against_the_oracle is the pytest-sized version of this."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np, torch
import gamd_oracle as orc
from helpers import rel_err, per_atom_err, edge_set
from gamd_amd.weights import ModelConfig, make_state_dict
from gamd_amd import workloads
from gamd_amd.engine import GamdForce
bad = 0; tot = 0
for sweep in range(8):
    rng = np.random.default_rng(7000 + sweep)
    for i in range(16):
        kind = ["lj", "water", "dynbox"][int(rng.integers(0, 3))]
        upd = kind == "dynbox" and rng.random() < 0.4
        enc = int(rng.integers(8, 257)); emb = enc if upd else int(rng.integers(8, 257)); hid = int(rng.integers(8, 129))
        dt = ["f32", "f16x3", "bf16"][int(rng.integers(0, 3))]
        if upd and dt != "f32": dt = "f32"
        if sweep >= 6:                              # hidden_dim above 128 (wide_d.hip): fp32 edge MLP, no update_edge
            hid, dt, upd = int(rng.integers(129, 257)), "f32", False
            if kind == "dynbox": emb = int(rng.integers(8, 257))
        cfg = ModelConfig(kind=kind, encoding_size=enc, edge_embedding_dim=emb, hidden_dim=hid, conv_layer=int(rng.integers(1, 6)),
                          use_bond=kind == "water" and rng.random() < 0.6, n_rbf=0 if (kind == "dynbox" and rng.random() < 0.4) else 40,
                          use_layer_norm=rng.random() < 0.7, update_edge=upd)
        seed = int(rng.integers(0, 10000))
        sd = make_state_dict(cfg, seed, 2.9, 1.1)
        if kind == "lj":
            pos, box = workloads.lj_box(int(rng.integers(40, 400)), seed=seed); species = bonds = feat = None; rc, fl = 7.5, "jaxmd"
        else:
            pos, box, species, bonds = workloads.water_box(int(rng.integers(15, 120)), seed=seed)
            feat, rc = torch.from_numpy(species.astype(np.float32)).view(-1, 1), 4.2
            fl = "torch" if kind == "dynbox" else "jaxmd"
            if not cfg.use_bond: bonds = None
        n = pos.shape[0]
        skin = rng.random() < 0.4
        nb = int(rng.choice([1, 1, 2, 3]))
        eng = GamdForce(sd, n, box, rc, bond=bonds, nbr_flavour=fl, edge_dtype=dt, neighbor_skin=rc / 6 if skin else 0.0, n_boxes=nb)
        p = torch.remainder(torch.from_numpy(pos).float(), float(box))
        pp = torch.cat([torch.remainder(p + 0.01 * b, float(box)) for b in range(nb)])
        sp = None if species is None else np.tile(species, nb)
        out = eng.forward(pp, species=sp).cpu().numpy()
        ok = True
        for b in range(nb):
            pb = pp[b * n:(b + 1) * n]
            edges = orc.neighbor_edges(pb, box, rc, fl)
            ref = (orc.forward_dynamic_box(sd, pb, feat, np.full(3, box, dtype=np.float32), rc) if kind == "dynbox"
                   else orc.forward(sd, pb, edges, box, feat=feat, bond=bonds)).numpy()
            err = rel_err(out[b * n:(b + 1) * n], ref)
            tol = 1e-2 if dt == "bf16" else 1e-5
            if not (err < tol): ok = False
        tot += 1
        if not ok:
            bad += 1
            print("FAIL", sweep, i, cfg, dt, skin, nb, err)
        eng.close()
print(f"matrix sweep: {tot} cases, {bad} failures")
