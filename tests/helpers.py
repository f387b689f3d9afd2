"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np
import torch

from gamd_amd.weights import ModelConfig, make_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    kind, H, D, Eh, L, bond = [str(x) for x in g["cfg"]][:6]
    n_rbf = int(g["cfg"][6]) if len(g["cfg"]) > 6 else 40
    cfg = ModelConfig(kind=kind, encoding_size=int(H), hidden_dim=int(D), edge_embedding_dim=int(Eh),
                      conv_layer=int(L), use_bond=bool(int(bond)), n_rbf=n_rbf)
    sd = make_state_dict(cfg, int(g["seed"]), float(g["length_mean"]), float(g["length_std"]))
    return g, cfg, sd


def rel_err(a, b):
    """max-norm relative error: max|a-b| / max|b| (the tolerance convention of
    BASELINE.json's "1e-5 relative fp32")."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def edge_set(edge_idx):
    e = np.asarray(edge_idx).astype(np.int64)
    key = e[0] * (e.max() + 1) + e[1]
    return np.sort(key)
