"""CPU oracle for GAMD's force-inference hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain PyTorch (CPU, fp32) restatement
of the reference algorithm; it exists to *check* the HIP path and to be timed
as the "reference CPU path" (`bench.py` -> `cpu_baseline`, kind "port").  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import it.  The product package `gamd_amd/` never does.

Pinning: the reference ships no golden vectors or tests for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference's own
`code/nn_module.py` / `code/md_module.py` executed in the build container
(`oracle/make_golden.py` -> `tests/golden/*.npz`, checked by
`tests/test_oracle_golden.py`).  The neighbour semantics of the jax-md path
(third-party: jax-md unpinned "latest" + jax/jaxlib 0.1.67, DGL 0.7.0 — none
vendored under /root/reference) are restated from the reference's call-site
arguments (graph_utils.py:21-25,51-61): edge set = {(i,j): |minimg(r_j-r_i)|^2 <
rc^2}, self pair included.  That part is "parity unpinned" by any executable
reference.

Every function cites the reference lines it follows (paths relative to
/root/reference/code).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# neighbour search
# --------------------------------------------------------------------------
def _min_image(d: Tensor, box: Tensor) -> Tensor:
    """nn_module.py:617-621 / md_module.py:66: remainder(d + L/2, L) - L/2 in fp32."""
    half = 0.5 * box
    return torch.remainder(d + half, box) - half


def _box_tensor(box, dtype=torch.float32) -> Tensor:
    """Scalar or per-axis box -> fp32 tensor [3] (nn_module.py:591-594)."""
    b = torch.as_tensor(np.asarray(box, dtype=np.float64)).to(dtype).reshape(-1)
    if b.numel() == 1:
        b = b.repeat(3)
    return b


def neighbor_edges(pos: Tensor, box, cutoff: float, flavour: str = "jaxmd",
                   block: int = 1024) -> Tensor:
    """All-pairs periodic radius search, blocked so it stays O(block*N) in memory.

    flavour "jaxmd": graph_utils.py:21-25 (mask_self=False -> self pair kept) and
        graph_utils.py:51-61 (mask = dr^2 < cutoff^2, strict).  Edge order is
        row-major (centre i, then neighbour j ascending); the reference's order
        is (i, slot) (train_network_lj.py:166-185) — the *set* is what matters.
    flavour "torch": md_module.py:93-126 (norm <= cutoff, self excluded).
        Returned rows follow md_module.py:121: row0 = column index b, row1 = row
        index a, with distance = pos[b]-pos[a]; order is (a major, b minor).

    Returns LongTensor [2, E]: row0 = centre, row1 = neighbour (flavour jaxmd).
    """
    pos = pos.to(torch.float32)
    n = pos.shape[0]
    boxt = _box_tensor(box)
    rc2 = torch.tensor(float(cutoff) ** 2, dtype=torch.float32)
    rc = torch.tensor(float(cutoff), dtype=torch.float32)
    rows, cols = [], []
    for s in range(0, n, block):
        e = min(n, s + block)
        if flavour == "jaxmd":
            # displacement(pos_i, pos_j) ; symmetric in |.|^2
            d = _min_image(pos[s:e, None, :] - pos[None, :, :], boxt)
            m = (d * d).sum(-1) < rc2
        else:
            # md_module.py:65: dist_mat[a, b] = pos[b] - pos[a]
            d = _min_image(pos[None, :, :] - pos[s:e, None, :], boxt)
            m = torch.norm(d, dim=-1) <= rc
            idx = torch.arange(s, e)
            m[idx - s, idx] = False
        a, b = torch.nonzero(m, as_tuple=True)
        rows.append(a + s)
        cols.append(b)
    a = torch.cat(rows)
    b = torch.cat(cols)
    if flavour == "jaxmd":
        return torch.stack([a, b]).long()
    return torch.stack([b, a]).long()


# --------------------------------------------------------------------------
# model pieces
# --------------------------------------------------------------------------
def rbf_centers(low: float = 0.0, high: float = 1.0, gap: float = 0.025) -> Tuple[Tensor, float]:
    """nn_module.py:237-240: ceil((high-low)/gap) centres via linspace; gamma = 1/gap."""
    num = int(np.ceil((high - low) / gap))
    return torch.tensor(np.linspace(low, high, num)).float(), 1.0 / gap


def mlp(sd: Dict[str, Tensor], prefix: str, x: Tensor, act: str,
        hidden_layer: int, activation_first: bool = False) -> Tensor:
    """nn_module.py:48-65: Sequential layout of MLP; indices are positions in
    the nn.Sequential (activations occupy slots and have no parameters)."""
    fn = {"gelu": F.gelu, "silu": F.silu}[act]

    def lin(i, v):
        return F.linear(v, sd[f"{prefix}.mlp_layer.{i}.weight"], sd[f"{prefix}.mlp_layer.{i}.bias"])

    if hidden_layer == 3 and not activation_first:       # Lin,act,Lin,act,Lin
        return lin(4, fn(lin(2, fn(lin(0, x)))))
    if hidden_layer == 2 and not activation_first:       # Lin,act,Lin
        return lin(2, fn(lin(0, x)))
    if hidden_layer == 2 and activation_first:           # act,Lin,act,Lin
        return lin(3, fn(lin(1, fn(x))))
    if hidden_layer == 1 and activation_first:           # act,Lin
        return lin(1, fn(x))
    raise ValueError("unsupported MLP layout")


def linear(sd, prefix, x):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def layer_norm(sd, prefix, x):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)


def node_norm(sd, prefix, x):
    """graph_conv.norm_layers[l] (nn_module.py:193-196, applied at :202): LayerNorm when the model was built with
    use_layer_norm=True (every rollout driver), else nn.BatchNorm1d (use_batch_norm = not use_layer_norm, :579) — in eval mode
    the affine map of its running statistics."""
    if prefix + ".running_mean" in sd:
        return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                            sd[prefix + ".bias"], training=False, eps=1e-5)
    return layer_norm(sd, prefix, x)


def edge_features(sd: Dict[str, Tensor], pos: Tensor, center: Tensor, neigh: Tensor, box) -> Tensor:
    """nn_module.py:603-634 (= :462-493): rel = pos[neigh]-pos[center], min-image,
    norm, unit vector r/(n+1e-8), standardised length, RBF of the standardised
    length (nn_module.py:248-263)."""
    boxt = _box_tensor(box)
    rel = pos[neigh] - pos[center]
    rel = _min_image(rel, boxt)
    nrm = rel.norm(dim=1).view(-1, 1)
    unit = rel / (nrm + 1e-8)
    d = (nrm - sd["length_mean"]) / sd["length_std"]
    centers = sd["edge_expand.centers"]
    gamma = 1.0 / 0.025
    radial = d - centers
    rbf = torch.exp(-gamma * (radial ** 2))
    return torch.cat((unit, d, rbf), dim=1)


def edge_features_from_dist(sd, distance: Tensor, distance_norm: Tensor) -> Tensor:
    """nn_module.py:322-336 (dynamic box): sign flip -distance/(norm+1e-8)."""
    nrm = distance_norm.view(-1, 1)
    unit = -distance / (nrm + 1e-8)
    d = (nrm - sd["length_mean"]) / sd["length_std"]
    if "edge_expand.centers" not in sd:                      # expand_edge=False, nn_module.py:333-335
        return torch.cat((unit, d), dim=1)
    radial = d - sd["edge_expand.centers"]
    rbf = torch.exp(-(1.0 / 0.025) * (radial ** 2))
    return torch.cat((unit, d, rbf), dim=1)


def bond_flags(center: Tensor, neigh: Tensor, bond: np.ndarray) -> Tensor:
    """nn_module.py:510 + :529-534: bond graph = bonds + reversed bonds;
    has_edges_between(center, neigh) -> bool[E]."""
    b = torch.as_tensor(np.asarray(bond)).long()
    if center.numel() == 0:
        return torch.zeros(0, dtype=torch.bool)
    n = int(max(b.max(), center.max(), neigh.max())) + 1
    key = torch.cat([b[:, 0] * n + b[:, 1], b[:, 1] * n + b[:, 0]])
    return torch.isin(center * n + neigh, key)


def conv_layer(sd, p: str, e: Tensor, hn: Tensor, src: Tensor, dst: Tensor, e_next: Optional[list] = None) -> Tensor:
    """SmoothConvLayerNew.forward, nn_module.py:108-148, op for op (Linear on E
    gathered rows, as written).  update_edge_emb=True (a state_dict with `<p>.edge_layer_norm.*`, :91-92): the layer
    leaves edata['e'] = edge_layer_norm(e_emb) on the graph for the layers after it (:140-141, :145-146; the assignment is
    made after local_scope() has closed, so it persists) — handed back through `e_next`."""
    edge_code = mlp(sd, p + ".edge_affine", e, "silu", 2)                     # :135
    src_code = linear(sd, p + ".src_affine", hn[src])                          # :136
    dst_code = linear(sd, p + ".dst_affine", hn[dst])                          # :137
    e_emb = mlp(sd, p + ".theta_edge", edge_code + src_code + dst_code, "silu", 2, True)  # :138
    if e_next is not None and (p + ".edge_layer_norm.weight") in sd:
        e_next.append(layer_norm(sd, p + ".edge_layer_norm", e_emb))           # :141 (width H == Eh or torch raises)
    agg = torch.zeros_like(hn)
    agg.index_add_(0, dst, hn[src] * e_emb)                                    # :142 u_mul_e -> sum
    return mlp(sd, p + ".phi", linear(sd, p + ".phi_dst", hn) + linear(sd, p + ".phi_edge", agg),
               "silu", 1, True)                                                # :147


SELF_LOOP_MODES = ("dgl07_noop", "append_zero_feature_loops")


def apply_self_loop_mode(e: Tensor, src: Tensor, dst: Tensor, n: int, mode: str):
    """`fluid_graph.add_self_loop()` with the result discarded (nn_module.py:650-652, :518, :364).
    "dgl07_noop": DGL >= 0.5 (pinned 0.7.0) returns a new graph, the one in use is unchanged.
    "append_zero_feature_loops": an in-place add_self_loop (DGL < 0.5) appends one edge i -> i per node AFTER
    edata['e'] was set, so the new edges carry DGL's zero-filled embedding (what oracle/ref_stubs.py does with
    INPLACE_SELF_LOOP = True)."""
    if mode == "dgl07_noop":
        return e, src, dst
    if mode != "append_zero_feature_loops":
        raise ValueError(f"self_loop_mode must be one of {SELF_LOOP_MODES}")
    loops = torch.arange(n, dtype=src.dtype)
    return (torch.cat([e, torch.zeros((n, e.shape[1]), dtype=e.dtype)]), torch.cat([src, loops]), torch.cat([dst, loops]))


def n_conv_layers(sd) -> int:
    n = 0
    while f"graph_conv.conv.{n}.src_affine.weight" in sd:
        n += 1
    return n


@torch.no_grad()
def forward(sd: Dict[str, Tensor], pos: Tensor, edge_idx: Tensor, box,
            feat: Optional[Tensor] = None, bond: Optional[np.ndarray] = None,
            stages: Optional[dict] = None, self_loop_mode: str = "dgl07_noop") -> Tensor:
    """SimpleMDNetNew.forward (nn_module.py:672-685) when `feat` is None, else
    WaterMDNetNew.forward (nn_module.py:545-558).  edge_idx rows: centre, neighbour.
    Returns the *normalised* force [N,3]."""
    center, neigh = edge_idx[0].long(), edge_idx[1].long()
    src, dst = neigh, center                                                   # :643 dgl.graph((neigh, center))
    f = edge_features(sd, pos, center, neigh, box)                             # :644
    if bond is not None:
        f = torch.cat((f, bond_flags(center, neigh, bond).view(-1, 1).to(f.dtype)), dim=1)   # :510-511
    e = layer_norm(sd, "edge_layer_norm", mlp(sd, "edge_encoder", f, "gelu", 3))  # :646
    n = pos.shape[0]
    if feat is None:
        h = sd["node_emb"].repeat((n, 1))                                      # :681
    else:
        h = linear(sd, "node_encoder", feat)                                   # :554
    if stages is not None:
        stages["feat"], stages["e"], stages["h"] = f, e, [h]
    e, src, dst = apply_self_loop_mode(e, src, dst, n, self_loop_mode)         # :650-652
    for l in range(n_conv_layers(sd)):                                         # :200-202
        hn = node_norm(sd, f"graph_conv.norm_layers.{l}", h)
        e_next = []
        h = conv_layer(sd, f"graph_conv.conv.{l}", e, hn, src, dst, e_next) + h
        if e_next:                                                             # update_edge_emb: :145-146
            e = e_next[0]
        if stages is not None:
            stages["h"].append(h)
    return mlp(sd, "graph_decoder", h, "gelu", 2)                              # :684


@torch.no_grad()
def forward_dynamic_box(sd, pos: Tensor, feat: Tensor, box, cutoff: float,
                        stages: Optional[dict] = None, self_loop_mode: str = "dgl07_noop") -> Tensor:
    """WaterMDDynamicBoxNet.forward, nn_module.py:391-407 with build_graph
    :338-365 (md_module.get_neighbor: <=, no self; bond=None configs)."""
    edge_idx = neighbor_edges(pos, box, cutoff, "torch")
    center, neigh = edge_idx[0], edge_idx[1]
    boxt = _box_tensor(box)
    dist = _min_image(pos[center] - pos[neigh], boxt)       # md_module.py:65,121: pos[b]-pos[a], b=row0
    f = edge_features_from_dist(sd, dist, dist.norm(dim=1))
    e = layer_norm(sd, "edge_layer_norm", mlp(sd, "edge_encoder", f, "gelu", 3))
    h = linear(sd, "node_encoder", feat)
    src, dst = neigh, center
    if stages is not None:
        stages["feat"], stages["e"], stages["h"], stages["edge_idx"] = f, e, [h], edge_idx
    e, src, dst = apply_self_loop_mode(e, src, dst, pos.shape[0], self_loop_mode)   # :364
    for l in range(n_conv_layers(sd)):
        hn = node_norm(sd, f"graph_conv.norm_layers.{l}", h)
        e_next = []
        h = conv_layer(sd, f"graph_conv.conv.{l}", e, hn, src, dst, e_next) + h
        if e_next:                                                             # update_edge_emb: :145-146
            e = e_next[0]
        if stages is not None:
            stages["h"].append(h)
    return mlp(sd, "graph_decoder", h, "gelu", 2)


def denormalize(pred: np.ndarray, var: np.ndarray, mean: np.ndarray) -> np.ndarray:
    """train_network_lj.py:128-131: pred*sqrt(var)+mean with f64 scaler scalars."""
    return pred * np.sqrt(var) + mean


@torch.no_grad()
def predict_forces(sd, pos: np.ndarray, box: float, cutoff: float,
                   var=np.array([1.0]), mean=np.array([0.0]),
                   feat: Optional[Tensor] = None, bond=None) -> np.ndarray:
    """ParticleNetLightning.predict_forces, LJ/train_network_lj.py:133-157 (and
    water/train_network_tip3p.py:142-159): neighbour search on the f32 cast of the
    raw positions, np.mod in f64, f32 forward, f64 denormalise."""
    pos32 = torch.from_numpy(np.asarray(pos)).float()
    # graph_utils.py:31: the searcher wraps with jnp.mod in f32
    edge_idx = neighbor_edges(torch.remainder(pos32, _box_tensor(box)), box, cutoff, "jaxmd")
    posw = torch.from_numpy(np.mod(np.asarray(pos, dtype=np.float64), np.array(box))).float()
    pred = forward(sd, posw, edge_idx, box, feat=feat, bond=bond).numpy()
    return denormalize(pred, var, mean)


# --------------------------------------------------------------------------
# integrator: the split BAOAB scheme of hack_integrator.py
# --------------------------------------------------------------------------
def baoab_first_half(x, v, f_last, inv_m, dt, a, b_sigma, noise):
    """HackLangevinIntegrator, hack_integrator.py:141-165 without constraints:
    B: v += dt/2 f/m ; A: x += dt/2 v ; O: v = a v + b sigma xi ; A: x += dt/2 v."""
    v = v + (0.5 * dt) * f_last * inv_m
    x = x + (0.5 * dt) * v
    v = a * v + b_sigma * noise
    x = x + (0.5 * dt) * v
    return x, v


def baoab_second_half(v, f, inv_m, dt):
    """HackHalfVelocityIntegrator, hack_integrator.py:171-178: v += dt/2 f/m."""
    return v + (0.5 * dt) * f * inv_m


def remove_cm_motion(v, mass):
    """OpenMM's CMMotionRemover (third-party, absent here; frequency 1), which the first-half integrators run through
    addUpdateContextState() at the top of every step (hack_integrator.py:142, :272) when the System carries one — the water
    drivers' openmmtools WaterBox does, and hack_integrator.py:226-235 takes 3 degrees of freedom off for it:
    v_i -= sum_j m_j v_j / sum_j m_j  (restated from OpenMM's ReferenceRemoveCMMotionKernel)."""
    v = np.asarray(v, dtype=np.float64)
    m = np.broadcast_to(np.asarray(mass, dtype=np.float64).reshape(-1, 1) if np.ndim(mass) else np.float64(mass), (v.shape[0], 1))
    return v - (m * v).sum(axis=0) / m.sum()


YS_WEIGHTS = {1: [1.0], 3: [0.8289815435887510, -0.6579630871775020, 0.8289815435887510],
              5: [0.2967324292201065, 0.2967324292201065, -0.1869297168804260, 0.2967324292201065,
                  0.2967324292201065]}                       # hack_integrator.py:183-187


def nhc_init(chain_length, freq):
    """xi = vxi = 0, G_i = -frequency^2 (hack_integrator.py:252-256)."""
    return dict(xi=np.zeros(chain_length), vxi=np.zeros(chain_length), G=np.full(chain_length, -freq ** 2))


def nhc_propagate(st, ke2, dt, kT, freq, ndf, n_c=5, n_ys=5):
    """propagateNHC, hack_integrator.py:289-316 (float64 globals).  Returns the velocity scale."""
    xi, vxi, G = st["xi"], st["vxi"], st["G"]
    M = len(xi)
    Q = kT / freq ** 2
    Qs = np.full(M, Q); Qs[0] = ndf * Q                                   # :262-265
    scale = 1.0
    G[0] = (ke2 - ndf * kT) / Qs[0]
    for _ in range(n_c):
        for w in YS_WEIGHTS[n_ys]:
            wdt = w * dt / n_c
            vxi[M - 1] += 0.25 * wdt * G[M - 1]
            for j in range(M - 2, -1, -1):
                aa = np.exp(-0.125 * wdt * vxi[j + 1])
                vxi[j] = aa * (aa * vxi[j] + 0.25 * wdt * G[j])
            scale *= np.exp(-0.5 * wdt * vxi[0])
            xi += 0.5 * wdt * vxi
            G[0] = (scale * scale * ke2 - ndf * kT) / Qs[0]
            for j in range(M - 1):
                aa = np.exp(-0.125 * wdt * vxi[j + 1])
                vxi[j] = aa * (aa * vxi[j] + 0.25 * wdt * G[j])
                G[j + 1] = (Qs[j] * vxi[j] ** 2 - kT) / Qs[j + 1]
            vxi[M - 1] += 0.25 * wdt * G[M - 1]
    return scale


def nhc_first_half(st, x, v, f_last, mass, dt, kT, freq, ndf, remove_com=False):
    """HackNoseHooverIntegrator step, hack_integrator.py:271-280 (x A, v A/ps, f kJ/mol/nm, no constraints):
    propagateNHC() :271 (KE2 from the velocities as they are, v *= scale), THEN addUpdateContextState() :272 — the
    System's CMMotionRemover when it has one (``remove_com``) — then the kick."""
    ke2 = float(np.sum(mass * (0.1 * v) ** 2))
    v = v * nhc_propagate(st, ke2, dt, kT, freq, ndf)
    if remove_com:
        v = remove_cm_motion(v, mass)
    v = v + 0.5 * dt * f_last * 10.0 / mass
    return x + dt * v, v


def nhc_second_half(st, v, f, mass, dt, kT, freq, ndf):
    """HackHalfNoseHooverIntegrator step, hack_integrator.py:427-430."""
    v = v + 0.5 * dt * f * 10.0 / mass
    ke2 = float(np.sum(mass * (0.1 * v) ** 2))
    return v * nhc_propagate(st, ke2, dt, kT, freq, ndf)


# --------------------------------------------------------------------------
# rigid water: the water drivers build their systems with rigid TIP3P/TIP4P-Ew molecules and
# the hacked integrators call addConstrainPositions / addConstrainVelocities (hack_integrator.py:145-164,178,
# 277-280,427-428).  OpenMM (third-party, absent here) solves those constraints with SETTLE for 3-site water;
# the constrained update is the unique solution of the SHAKE / RATTLE equations (displacements along the bonds
# of the reference geometry, weighted by 1/m), so the oracle iterates those equations to convergence in float64.
# --------------------------------------------------------------------------
def water_constraints(n_atoms, r_oh, r_hh):
    """O,H,H triples (water/train_utils.py:25-26 order): bonds O-H1, O-H2, H1-H2 and their lengths."""
    o = np.arange(0, n_atoms, 3)
    pairs = np.concatenate([np.stack([o, o + 1], 1), np.stack([o, o + 2], 1), np.stack([o + 1, o + 2], 1)])
    lengths = np.concatenate([np.full(len(o), r_oh), np.full(len(o), r_oh), np.full(len(o), r_hh)])
    return pairs, lengths


def shake_positions(x_ref, x_new, inv_m, pairs, lengths, tol=1e-13, max_iter=2000):
    """ConstrainPositions: move x_new along the bonds of the reference geometry x_ref (which satisfies the
    constraints) until |x_i - x_j| = d for every pair."""
    x = np.array(x_new, dtype=np.float64)
    x_ref = np.asarray(x_ref, dtype=np.float64)
    i, j = pairs[:, 0], pairs[:, 1]
    r_ref = x_ref[i] - x_ref[j]
    wi, wj = inv_m[i].reshape(-1, 1), inv_m[j].reshape(-1, 1)
    n_mol_pairs = len(pairs) // 3
    for _ in range(max_iter):
        worst = 0.0
        for k in range(3):                      # the three bond families touch disjoint atom pairs: Gauss-Seidel over families
            sl = slice(k * n_mol_pairs, (k + 1) * n_mol_pairs)
            r = x[i[sl]] - x[j[sl]]
            diff = lengths[sl] ** 2 - np.sum(r * r, axis=1)
            worst = max(worst, float(np.max(np.abs(diff) / lengths[sl] ** 2)))
            g = diff / (2.0 * np.sum(r * r_ref[sl], axis=1) * (wi[sl, 0] + wj[sl, 0]))
            x[i[sl]] += (g * wi[sl, 0])[:, None] * r_ref[sl]
            x[j[sl]] -= (g * wj[sl, 0])[:, None] * r_ref[sl]
        if worst < tol:
            break
    return x


def rattle_velocities(x, v, inv_m, pairs, tol=1e-14, max_iter=2000):
    """ConstrainVelocities: remove the relative velocity along every constrained bond."""
    v = np.array(v, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    i, j = pairs[:, 0], pairs[:, 1]
    r = x[i] - x[j]
    r2 = np.sum(r * r, axis=1)
    wi, wj = inv_m[i], inv_m[j]
    n_mol_pairs = len(pairs) // 3
    for _ in range(max_iter):
        worst = 0.0
        for k in range(3):
            sl = slice(k * n_mol_pairs, (k + 1) * n_mol_pairs)
            rv = np.sum(r[sl] * (v[i[sl]] - v[j[sl]]), axis=1)
            worst = max(worst, float(np.max(np.abs(rv))))
            g = -rv / (r2[sl] * (wi[sl] + wj[sl]))
            v[i[sl]] += (g * wi[sl])[:, None] * r[sl]
            v[j[sl]] -= (g * wj[sl])[:, None] * r[sl]
        if worst < tol:
            break
    return v


def baoab_first_half_rigid(x, v, f_last, inv_m, dt, a, b_sigma, noise, pairs, lengths, acc_unit=10.0):
    """HackLangevinIntegrator with constraints, hack_integrator.py:141-165, line by line.
    b_sigma: per-atom b*sqrt(kT/m) in the length unit; acc_unit: length units per nm (force is kJ/mol/nm)."""
    w = inv_m.reshape(-1)
    v = v + (0.5 * dt) * acc_unit * f_last * inv_m                      # :145
    v = rattle_velocities(x, v, w, pairs)                                # :146
    for stage in range(2):
        x1 = x + (0.5 * dt) * v                                          # :149 / :160
        xc = shake_positions(x, x1, w, pairs, lengths)                   # :150-151 / :161-162
        v = v + (xc - x1) / (0.5 * dt)                                   # :152 / :163
        x = xc
        v = rattle_velocities(x, v, w, pairs)                            # :153 / :164
        if stage == 0:
            v = a * v + b_sigma * noise                                  # :157
            v = rattle_velocities(x, v, w, pairs)                        # :158
    return x, v


def baoab_second_half_rigid(x, v, f, inv_m, dt, pairs, acc_unit=10.0):
    """HackHalfVelocityIntegrator, hack_integrator.py:177-178."""
    return rattle_velocities(x, v + (0.5 * dt) * acc_unit * f * inv_m, inv_m.reshape(-1), pairs)


def nhc_first_half_rigid(st, x, v, f_last, mass, dt, kT, freq, ndf, pairs, lengths, acc_unit=10.0, remove_com=False):
    """HackNoseHooverIntegrator with constraints, hack_integrator.py:271-280: propagateNHC, updateContextState (the
    CMMotionRemover, ``remove_com``), v kick, x += dt v, ConstrainPositions, v += (x - x1)/dt (no ConstrainVelocities in this
    half)."""
    ke2 = float(np.sum(mass * (v / acc_unit) ** 2))
    v = v * nhc_propagate(st, ke2, dt, kT, freq, ndf)
    if remove_com:
        v = remove_cm_motion(v, mass)
    v = v + 0.5 * dt * f_last * acc_unit / mass
    x1 = x + dt * v
    xc = shake_positions(x, x1, (1.0 / mass).reshape(-1), pairs, lengths)
    return xc, v + (xc - x1) / dt


def nhc_second_half_rigid(st, x, v, f, mass, dt, kT, freq, ndf, pairs, acc_unit=10.0):
    """HackHalfNoseHooverIntegrator with constraints, hack_integrator.py:427-430."""
    v = v + 0.5 * dt * f * acc_unit / mass
    v = rattle_velocities(x, v, (1.0 / mass).reshape(-1), pairs)
    ke2 = float(np.sum(mass * (v / acc_unit) ** 2))
    return v * nhc_propagate(st, ke2, dt, kT, freq, ndf)
