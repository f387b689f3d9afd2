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
                      conv_layer=int(L), use_bond=bool(int(bond)), n_rbf=n_rbf,
                      use_layer_norm=bool(int(g["use_layer_norm"])) if "use_layer_norm" in g else True,
                      update_edge=bool(int(g["update_edge"])) if "update_edge" in g else False)
    sd = make_state_dict(cfg, int(g["seed"]), float(g["length_mean"]), float(g["length_std"]))
    return g, cfg, sd


def rel_err(a, b):
    """max-norm relative error: max|a-b| / max|b| (the tolerance convention of
    BASELINE.json's "1e-5 relative fp32")."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def per_atom_err(a, b, floor=1e-3):
    """Per-atom relative error |a_i - b_i| / |b_i| (vector norms) over the atoms whose reference force exceeds
    `floor` x the largest one: (median, p99, max, atoms counted).  The max-norm `rel_err` lets an atom with a small force
    be off by orders of magnitude in its own terms; this statistic does not."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    nb = np.linalg.norm(b, axis=1)
    keep = nb > floor * nb.max()
    r = np.linalg.norm(a - b, axis=1)[keep] / nb[keep]
    return float(np.median(r)), float(np.percentile(r, 99)), float(r.max()), int(keep.sum())


def edge_set_diff_near_cutoff(edges_a, edges_b, pos, box, cutoff, n):
    """Directed pairs that are in exactly one of the two edge lists, with the f64 minimum-image distance of each:
    (pairs [k,2], |distance - cutoff| [k]).  Two exact searches may only disagree on pairs that sit on the cutoff to
    fp32 rounding."""
    ka = np.asarray(edges_a[0], dtype=np.int64) * n + np.asarray(edges_a[1], dtype=np.int64)
    kb = np.asarray(edges_b[0], dtype=np.int64) * n + np.asarray(edges_b[1], dtype=np.int64)
    d = np.setxor1d(ka, kb)
    i, j = d // n, d % n
    b = np.broadcast_to(np.asarray(box, dtype=np.float64), (3,))
    r = np.asarray(pos, dtype=np.float64)[j] - np.asarray(pos, dtype=np.float64)[i]
    r -= b * np.round(r / b)
    return np.stack([i, j], 1), np.abs(np.linalg.norm(r, axis=1) - cutoff)


def edge_set(edge_idx):
    e = np.asarray(edge_idx).astype(np.int64)
    key = e[0] * (e.max() + 1) + e[1]
    return np.sort(key)
