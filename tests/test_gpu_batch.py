"""Several independent boxes in one set of launches (gamd_config.n_boxes; the reference's several-graphs-per-forward,
build_graph_batches + dgl.batch, nn_module.py:655-661, :520-527, :676-679).  The bar: a batch is BIT-IDENTICAL to its
boxes evaluated one by one (forces, trajectories, thermostat chains), on every neighbour path."""
import os

import numpy as np
import pytest
import torch

import gamd_oracle as orc
from helpers import GOLDEN, load_golden, rel_err, edge_set
from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS
from gamd_amd import workloads

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _engine(*a, **kw):
    from gamd_amd.engine import GamdForce
    return GamdForce(*a, **kw)


def _lj_boxes(nb, n=258, seed=0):
    """nb different configurations of the reference's LJ system: its snapshot plus per-box noise (box 0: the snapshot)."""
    g = load_golden("lj258_seed0")[0]
    base = np.mod(g["pos"], float(g["box"]))[:n]
    rng = np.random.default_rng(seed)
    return [base + (rng.normal(0, 0.3, base.shape) if b else 0.0) for b in range(nb)], float(g["box"]), float(g["cutoff"])


@pytest.mark.parametrize("skin", [0.0, 1.25])
@pytest.mark.parametrize("nb", [2, 5])
def test_batch_forces_are_bit_identical_to_the_boxes_one_by_one(nb, skin):
    pos, box, rc = _lj_boxes(nb)
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 5.3, 1.6)
    batch = _engine(sd, 258, box, rc, n_boxes=nb, neighbor_skin=skin, scaler=SHIPPED_SCALERS["lj"])
    x = torch.from_numpy(np.concatenate(pos)).float()
    out = batch.forward(x).cpu().numpy()
    den = batch.forward(x.view(nb, 258, 3), denormalize=True).cpu().numpy()      # [B, n, 3] is accepted too
    edges = batch.debug_edges()
    assert out.shape == (nb * 258, 3)
    assert np.all(edges[0] // 258 == edges[1] // 258)                # no edge crosses boxes
    row_ptr, col = batch.debug_csr()
    assert all(row_ptr[b * 258] % 16 == 0 for b in range(nb))        # every box starts on a chunk boundary
    assert int((col == nb * 258).sum()) == batch.counts()[0] - edges.shape[1] <= 15 * (nb - 1)
    single = _engine(sd, 258, box, rc, neighbor_skin=skin, scaler=SHIPPED_SCALERS["lj"])
    for b in range(nb):
        one = single.forward(torch.from_numpy(pos[b]).float()).cpu().numpy()
        assert np.array_equal(out[b * 258:(b + 1) * 258], one), b
        one_d = single.forward(torch.from_numpy(pos[b]).float(), denormalize=True).cpu().numpy()
        assert np.array_equal(den[b * 258:(b + 1) * 258], one_d), b
        eb = edges[:, edges[0] // 258 == b] - b * 258
        assert np.array_equal(edge_set(eb), edge_set(single.debug_edges())), b
    # box 0 is the reference's snapshot: its forces are the reference's
    g = load_golden("lj258_seed0")[0]
    assert rel_err(out[:258], g["out_norm"]) < TOL
    batch.close(); single.close()


def test_model_level_batch_golden_through_the_native_path():
    """The reference's own two-graph forward (tests/golden/lj258_batch2_seed0): ONE batched evaluation, within 1e-5 of the
    reference and bit-identical to the per-graph evaluation."""
    from gamd_amd.compat import ParticleNetLightningLJ
    g = dict(np.load(os.path.join(GOLDEN, "lj258_batch2_seed0.npz"), allow_pickle=False))
    kind, H, D, Eh, L, bond = [str(x) for x in g["cfg"]][:6]
    cfg = ModelConfig(kind=kind, encoding_size=int(H), hidden_dim=int(D), edge_embedding_dim=int(Eh), conv_layer=int(L))
    sd = make_state_dict(cfg, int(g["seed"]), float(g["length_mean"]), float(g["length_std"]))
    m = ParticleNetLightningLJ(state_dict=sd, num_atoms=258, box_size=float(g["box"]), cutoff=float(g["cutoff"]))
    pos_lst = [torch.from_numpy(g[f"pos{i}"]).cuda() for i in range(2)]
    edge_lst = [torch.from_numpy(g[f"edge_idx{i}"]).long().cuda() for i in range(2)]
    out = m.pnet_model(pos_lst, edge_lst).cpu().numpy()
    assert set(m._engines) == {2} and m._engines[2].n_boxes == 2           # one engine, two boxes: not a loop over graphs
    assert rel_err(out, g["out_norm"]) < TOL
    singles = np.concatenate([m.pnet_model([p], [e]).cpu().numpy() for p, e in zip(pos_lst, edge_lst)])
    assert np.array_equal(out, singles)
    # numpy inputs and a third graph
    out3 = m.pnet_model([g["pos0"], g["pos1"], g["pos0"]], [g["edge_idx0"], g["edge_idx1"], g["edge_idx0"]]).cpu().numpy()
    assert np.array_equal(out3[:516], out) and np.array_equal(out3[516:], out[:258])


def test_water_batch_with_bonds_and_species():
    g, cfg, sd = load_golden("tip3p774_seed3")
    box, rc, n = float(g["box"]), float(g["cutoff"]), 774
    species = g["node_feat"].reshape(-1) != 0
    rng = np.random.default_rng(3)
    pos = [np.mod(g["pos"], box) + (rng.normal(0, 0.05, (n, 3)) if b else 0.0) for b in range(3)]
    for dtype in ("f32", "f16x3", "bf16"):
        batch = _engine(sd, n, box, rc, bond=g["bond"], n_boxes=3, edge_dtype=dtype)
        single = _engine(sd, n, box, rc, bond=g["bond"], edge_dtype=dtype)
        out = batch.forward(torch.from_numpy(np.concatenate(pos)).float(), species=species).cpu().numpy()   # one box's species
        for b in range(3):
            one = single.forward(torch.from_numpy(pos[b]).float(), species=species).cpu().numpy()
            assert np.array_equal(out[b * n:(b + 1) * n], one), (dtype, b)
        if dtype != "bf16":
            assert rel_err(out[:n], g["out_norm"]) < TOL
        batch.close(); single.close()


def test_dynamic_box_batch_has_a_box_per_graph():
    """WaterMDDynamicBoxNet.forward(pos_lst, x, box_size_lst, cutoff) with several graphs (nn_module.py:391-407, :366-377):
    every graph has its own orthorhombic box; generic-width kernels (the shipped DFT widths)."""
    from types import SimpleNamespace
    from gamd_amd.compat import ParticleNetLightningDFT
    g, cfg, sd = load_golden("dynbox384_dftcfg_seed5")
    n, rc = g["pos"].shape[0], float(g["cutoff"])
    scales = [1.0, 1.04, 0.97]
    boxes = [g["box"] * np.float32(s) for s in scales]
    poss = [(g["pos"] * np.float32(s)).astype(np.float32) for s in scales]
    m = ParticleNetLightningDFT(SimpleNamespace(cutoff=rc), sd, num_atoms=n)
    feat = torch.from_numpy(g["node_feat"]).cuda()
    out = m.pnet_model([torch.from_numpy(p) for p in poss], torch.cat([feat] * 3), boxes, rc).cpu().numpy()
    assert out.shape == (3 * n, 3)
    assert rel_err(out[:n], g["out_norm"]) < TOL                       # graph 0 is the reference golden
    for b in range(3):
        one = m.pnet_model([torch.from_numpy(poss[b])], feat, [boxes[b]], rc).cpu().numpy()
        assert np.array_equal(out[b * n:(b + 1) * n], one), b
        ref = orc.forward_dynamic_box(sd, torch.from_numpy(poss[b]), torch.from_numpy(g["node_feat"]), boxes[b], rc).numpy()
        assert rel_err(one, ref) < TOL, b
    eb = m._engines[3].debug_edges()
    assert not np.any(eb[0] == eb[1]) and np.all(eb[0] // n == eb[1] // n)      # torch flavour: no self edges; no cross-box edges


def _md_inputs(pos, nb, n, mass, seed=1):
    x = torch.from_numpy(np.concatenate(pos)).float().cuda()
    v = torch.from_numpy(np.concatenate([workloads.maxwell_boltzmann(n, mass_amu=mass, temperature_k=200.0, seed=seed + b)
                                         for b in range(nb)])).float().cuda()
    return x, v


def test_batch_md_run_equals_single_box_runs_with_seed_plus_box():
    """BAOAB over a batch, exact neighbour rebuild every step: box b follows the trajectory a single-box run with seed + b
    follows, bit for bit (same noise stream, same forces)."""
    nb, n = 4, 258
    pos, box, rc = _lj_boxes(nb, seed=5)
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 5.3, 1.6)
    batch = _engine(sd, n, box, rc, n_boxes=nb, scaler=SHIPPED_SCALERS["lj"])
    single = _engine(sd, n, box, rc, scaler=SHIPPED_SCALERS["lj"])
    x, v = _md_inputs(pos, nb, n, 39.9)
    f = batch.forward(x, denormalize=True).clone()
    x0, v0, f0 = x.clone(), v.clone(), f.clone()
    batch.md_run(x, v, f, 12, temperature_k=200.0, seed=77, first_step=3)
    for b in range(nb):
        sl = slice(b * n, (b + 1) * n)
        xs, vs, fs = x0[sl].clone(), v0[sl].clone(), f0[sl].clone()
        single.md_run(xs, vs, fs, 12, temperature_k=200.0, seed=77 + b, first_step=3)
        assert torch.equal(x[sl], xs) and torch.equal(v[sl], vs) and torch.equal(f[sl], fs), b
    assert not torch.equal(v[:n] - v0[:n], v[n:2 * n] - v0[n:2 * n])      # the boxes really have different noise
    batch.close(); single.close()


def test_batch_md_run_in_skin_mode_tracks_the_single_box_runs():
    """Verlet-skin reuse on a batch (integrator halves fused into the skin check): the rebuild is triggered by any atom of
    any box, so the order inside a CSR row — not the set — can differ from a single-box run; forces and trajectories agree
    to fp32 rounding, the edge set is exact at the end."""
    nb, n = 6, 258
    pos, box, rc = _lj_boxes(nb, seed=9)
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 5.3, 1.6)
    batch = _engine(sd, n, box, rc, n_boxes=nb, scaler=SHIPPED_SCALERS["lj"], neighbor_skin=rc / 6.0)
    single = _engine(sd, n, box, rc, scaler=SHIPPED_SCALERS["lj"], neighbor_skin=rc / 6.0)
    x, v = _md_inputs(pos, nb, n, 39.9)
    f = batch.forward(x, denormalize=True).clone()
    x0, v0, f0 = x.clone(), v.clone(), f.clone()
    batch.md_run(x, v, f, 60, temperature_k=200.0, seed=5)
    assert batch.skin_stats()[0] >= 2
    for b in range(nb):
        sl = slice(b * n, (b + 1) * n)
        xs, vs, fs = x0[sl].clone(), v0[sl].clone(), f0[sl].clone()
        single.md_run(xs, vs, fs, 60, temperature_k=200.0, seed=5 + b)
        d = (x[sl] - xs).cpu().numpy()
        d -= box * np.round(d / box)
        assert np.abs(d).max() < 2e-3 and rel_err(v[sl].cpu().numpy(), vs.cpu().numpy()) < 2e-3, b
    exact = _engine(sd, n, box, rc, n_boxes=nb)
    exact.forward(x)
    assert np.array_equal(edge_set(exact.debug_edges()), edge_set(batch.debug_edges()))
    batch.close(); single.close(); exact.close()


def test_batch_rigid_water_and_nose_hoover_chains_per_box():
    from gamd_amd import workloads as wl
    g, cfg, sd = load_golden("tip3p774_seed3")
    nmol, nb = 64, 3
    boxes = [wl.water_box(nmol, seed=20 + b, jitter=0.0, wrap=False) for b in range(nb)]
    pos, box, species, bonds = boxes[0]
    n = 3 * nmol
    md = dict(dt_ps=0.0005, mass_amu=wl.MASS_O, mass_h_amu=wl.MASS_H, rigid_water=True, r_oh=wl.TIP3P_R_OH, r_hh=wl.TIP3P_R_HH)
    batch = _engine(sd, n, box, 4.2, bond=bonds, n_boxes=nb, scaler=SHIPPED_SCALERS["tip3p"])
    single = _engine(sd, n, box, 4.2, bond=bonds, scaler=SHIPPED_SCALERS["tip3p"])
    sp = np.tile(species, nb)
    x = torch.from_numpy(np.concatenate([b[0] for b in boxes])).float().cuda()
    v = torch.from_numpy(np.concatenate([wl.maxwell_boltzmann(n, mass_amu=wl.MASS_O, temperature_k=250.0, seed=3 + b)
                                         for b in range(nb)])).float().cuda()
    f = batch.forward(x, species=sp, denormalize=True).clone()
    x0, v0, f0 = x.clone(), v.clone(), f.clone()
    # Langevin with SETTLE
    batch.md_run(x, v, f, 8, temperature_k=250.0, seed=11, species=sp, **md)
    for b in range(nb):
        sl = slice(b * n, (b + 1) * n)
        xs, vs, fs = x0[sl].clone(), v0[sl].clone(), f0[sl].clone()
        single.md_run(xs, vs, fs, 8, temperature_k=250.0, seed=11 + b, species=species, **md)
        assert torch.equal(x[sl], xs) and torch.equal(v[sl], vs) and torch.equal(f[sl], fs), b
    # Nose-Hoover: one chain per box
    x, v, f = x0.clone(), v0.clone(), f0.clone()
    chain = batch.md_run_nhc(x, v, f, 5, temperature_k=250.0, species=sp, **md)
    assert tuple(chain.shape) == (nb, 32)
    for b in range(nb):
        sl = slice(b * n, (b + 1) * n)
        xs, vs, fs = x0[sl].clone(), v0[sl].clone(), f0[sl].clone()
        ch = single.md_run_nhc(xs, vs, fs, 5, temperature_k=250.0, species=species, **md)
        assert torch.equal(x[sl], xs) and torch.equal(v[sl], vs), b
        assert torch.equal(chain[b], ch), b
    assert not torch.equal(chain[0], chain[1])
    batch.close(); single.close()


@pytest.mark.parametrize("skin_frac", [0.0, 1.0 / 6.0])
def test_batch_overflow_is_regrown_also_in_the_middle_of_an_md_run(skin_frac):
    """A too-small edge capacity: the forward call regrows and retries (status 1); inside an enqueued md_run the device
    freezes every box at the last consistent step, the host regrows and resumes.  Exact mode: the trajectory is the
    ample-buffer one bit for bit.  Skin mode: the regrow forces a candidate rebuild the ample run does not have, which
    changes the order inside CSR rows (not the set), so the trajectories agree to fp32 rounding."""
    nb, n = 5, 258
    pos, box, rc = _lj_boxes(nb, seed=2)
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 5.3, 1.6)
    kw = dict(n_boxes=nb, scaler=SHIPPED_SCALERS["lj"], neighbor_skin=skin_frac * rc)
    ample = _engine(sd, n, box, rc, **kw)
    x = torch.from_numpy(np.concatenate(pos)).float().cuda()
    ref = ample.forward(x).cpu().numpy()
    e_now = ample.counts()[0]
    small = _engine(sd, n, box, rc, edge_capacity=2000, **kw)
    out = small.forward(x).cpu().numpy()
    assert small.last_status == 1 and np.array_equal(out, ref)
    small.close()
    # capacity that holds the first list but not what the contracting boxes need later
    tight = _engine(sd, n, box, rc, edge_capacity=e_now + 40, **kw)
    # velocities that pull every box's atoms towards its centre: the edge count grows step by step
    centre = torch.tensor([box / 2] * 3, device="cuda")
    v = (-(torch.remainder(x, box).view(nb, n, 3) - centre)).reshape(-1, 3).contiguous() * 1.5
    xa, va = x.clone(), v.clone()
    fa = ample.forward(xa, denormalize=True).clone()
    xt, vt = x.clone(), v.clone()
    ft = tight.forward(xt, denormalize=True).clone()
    assert tight.last_status == 0
    ample.md_run(xa, va, fa, 30, temperature_k=0.0, gamma_per_ps=0.0, seed=1)
    tight.md_run(xt, vt, ft, 30, temperature_k=0.0, gamma_per_ps=0.0, seed=1)
    assert tight.last_status == 1, "the run was meant to outgrow its edge buffer"
    assert ample.counts()[0] > e_now + 40
    assert np.array_equal(edge_set(ample.debug_edges()), edge_set(tight.debug_edges()))
    if skin_frac == 0.0:
        assert torch.equal(xa, xt) and torch.equal(va, vt) and torch.equal(fa, ft)
    else:
        assert rel_err(xt.cpu().numpy(), xa.cpu().numpy()) < 1e-5 and rel_err(vt.cpu().numpy(), va.cpu().numpy()) < 1e-4
        assert rel_err(ft.cpu().numpy(), fa.cpu().numpy()) < 1e-4
    ample.close(); tight.close()


def test_batch_argument_checks():
    from gamd_amd._lib import GamdError
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 5.3, 1.6)
    eng = _engine(sd, 258, 27.27, 7.5, n_boxes=3)
    with pytest.raises(ValueError, match="pos must be"):
        eng.forward(torch.zeros(258, 3))
    with pytest.raises(ValueError, match="box must be"):
        eng.forward(torch.zeros(774, 3), box=np.ones((2, 3)))
    eng.close()
    with pytest.raises(GamdError, match="at most"):
        _engine(sd, 1 << 20, 1.0e3, 7.5, n_boxes=8)


def test_batch_on_the_generic_width_kernels_and_with_appended_self_loops():
    """The trainers' default widths (256 / 128 / 256, wide.hip) and self_loop_mode 1 (one zero-embedding loop appended per
    atom) on a batch: bit-identical to the boxes one by one; box 0 reproduces the reference goldens."""
    g, cfg, sd = load_golden("lj258_w256_seed9")
    box, rc, n = float(g["box"]), float(g["cutoff"]), 258
    rng = np.random.default_rng(12)
    pos = [np.mod(g["pos"], box) + (rng.normal(0, 0.2, (n, 3)) if b else 0.0) for b in range(3)]
    batch = _engine(sd, n, box, rc, n_boxes=3, neighbor_skin=rc / 6.0)
    single = _engine(sd, n, box, rc, neighbor_skin=rc / 6.0)
    out = batch.forward(torch.from_numpy(np.concatenate(pos)).float()).cpu().numpy()
    assert rel_err(out[:n], g["out_norm"]) < TOL
    for b in range(3):
        assert np.array_equal(out[b * n:(b + 1) * n], single.forward(torch.from_numpy(pos[b]).float()).cpu().numpy()), b
    batch.close(); single.close()

    g, cfg, sd = load_golden("lj258_selfloop_inplace_seed0")
    pos = [np.mod(g["pos"], box) + (rng.normal(0, 0.2, (n, 3)) if b else 0.0) for b in range(3)]
    kw = dict(self_loop_mode="append_zero_feature_loops", keep_stages=True)
    batch = _engine(sd, n, box, rc, n_boxes=3, **kw)
    single = _engine(sd, n, box, rc, **kw)
    out = batch.forward(torch.from_numpy(np.concatenate(pos)).float()).cpu().numpy()
    assert rel_err(out[:n], g["out_norm"]) < TOL
    # stage getters on a batch: the padding slots are not edges (debug_edge_rows masks them out of debug_e / debug_feat)
    rows = batch.debug_edge_rows()
    eb, fb = batch.debug_e()[rows], batch.debug_feat(44)[rows]
    edges = batch.debug_edges()
    assert eb.shape[0] == edges.shape[1] == fb.shape[0]
    off = 0
    for b in range(3):
        one = single.forward(torch.from_numpy(pos[b]).float()).cpu().numpy()
        assert np.array_equal(out[b * n:(b + 1) * n], one), b
        e1 = single.debug_e()
        assert np.array_equal(eb[off:off + e1.shape[0]], e1), b
        assert np.array_equal(edges[:, off:off + e1.shape[0]] - b * n, single.debug_edges()), b
        hb = batch.debug_h(4)[b * n:(b + 1) * n]
        assert np.array_equal(hb, single.debug_h(4)), b
        off += e1.shape[0]
    batch.close(); single.close()


def test_model_level_batch_with_the_in_place_self_loop_reading():
    """pnet_model(pos_lst, edge_lst) on the caller's edge lists with self_loop_mode 1: the CSR rows of a batch are
    [the caller's edges][the appended loop][padding slots]; both graphs reproduce the reference's in-place-add_self_loop output."""
    from gamd_amd.compat import ParticleNetLightningLJ
    g, cfg, sd = load_golden("lj258_selfloop_inplace_seed0")
    box = float(g["box"])
    m = ParticleNetLightningLJ(state_dict=sd, num_atoms=258, box_size=box, cutoff=float(g["cutoff"]),
                               self_loop_mode="append_zero_feature_loops")
    posw = torch.from_numpy(np.mod(g["pos"], box)).float().cuda()
    e = torch.from_numpy(g["edge_idx"]).long().cuda()
    one = m.pnet_model([posw], [e]).cpu().numpy()
    three = m.pnet_model([posw, posw, posw], [e, e, e]).cpu().numpy()
    assert rel_err(one, g["out_norm"]) < TOL
    for b in range(3):
        assert np.array_equal(three[b * 258:(b + 1) * 258], one), b


def test_batch_with_isolated_atoms_at_box_boundaries():
    """Sparse boxes in the torch flavour (no self edges): atoms without neighbours, also as the LAST atom of a box, whose CSR
    row then consists of padding slots only.  Batch == boxes one by one, isolated atoms aggregate to zero."""
    rng = np.random.default_rng(5)
    n, box, rc, nb = 97, 11.0, 2.0, 4
    cfg = ModelConfig(kind="water")
    sd = make_state_dict(cfg, 9, 2.0, 0.7)
    species = (np.arange(n) % 3 == 0)
    # atoms are renumbered in cell order: the atom with the highest id in the highest cell is the last of its box's sorted
    # order — put it there, away from everything (all other atoms keep >= 2 A from that corner and its periodic images)
    pos = [rng.uniform(1.5, box - 3.0, (n, 3)) for _ in range(nb)]
    for p in pos:
        p[-1] = box - 0.5
    batch = _engine(sd, n, box, rc, nbr_flavour="torch", n_boxes=nb)
    single = _engine(sd, n, box, rc, nbr_flavour="torch")
    out = batch.forward(torch.from_numpy(np.concatenate(pos)).float(), species=species).cpu().numpy()
    edges = batch.debug_edges()
    deg = np.bincount(edges[0], minlength=nb * n)
    assert (deg == 0).sum() >= nb
    seen_isolated_last = 0
    for b in range(nb):
        one = single.forward(torch.from_numpy(pos[b]).float(), species=species).cpu().numpy()
        assert np.array_equal(out[b * n:(b + 1) * n], one), b
        seen_isolated_last += int(deg[(b + 1) * n - 1] == 0 and batch.debug_perm()[(b + 1) * n - 1] == (b + 1) * n - 1)
        ref = orc.forward(sd, torch.from_numpy(pos[b]).float(), torch.from_numpy(single.debug_edges()).long(), box,
                          feat=torch.from_numpy(species.astype(np.float32)).view(-1, 1)).numpy()
        assert rel_err(one, ref) < TOL
    assert seen_isolated_last == nb
    batch.close(); single.close()


def test_three_boxes_with_one_non_cubic_box_argument():
    """n_boxes == 3 and a box of three numbers (round-4 advisor): [Lx, Ly, Lz] is ONE orthorhombic box for all three graphs —
    as the constructor box, as forward's default (self.box, stored as [3]) and as an explicit 1-D argument — never three
    cubic boxes.  The dynamic-box golden's box is [20, 21, 22.5]; every box of the batch reproduces the reference output."""
    g, cfg, sd = load_golden("dynbox384_seed4")
    n, rc, box = g["pos"].shape[0], float(g["cutoff"]), g["box"].astype(np.float32)
    assert box.shape == (3,) and len(set(box.tolist())) == 3
    pos3 = torch.from_numpy(np.concatenate([g["pos"]] * 3)).float()
    species = np.tile(g["node_feat"].reshape(-1) != 0, 3)
    eng = _engine(sd, n, box, rc, nbr_flavour="torch", cfg=cfg, n_boxes=3)
    for arg in (None, box, np.tile(box, (3, 1))):
        out = eng.forward(pos3, box=arg, species=species).cpu().numpy()
        for b in range(3):
            assert rel_err(out[b * n:(b + 1) * n], g["out_norm"]) < TOL, (b, arg)
        assert eng.counts()[0] - sum(eng.debug_csr()[1] >= 3 * n) == 3 * g["edge_idx"].shape[1]
    # per-box cubic edges of a 3-box batch are spelled [3, 1]; they are NOT what [Lx, Ly, Lz] means
    cubic = eng.forward(pos3, box=box.reshape(3, 1), species=species).cpu().numpy()
    assert rel_err(cubic[n:2 * n], g["out_norm"]) > 1e-3
    eng.close()


def test_model_level_call_with_graphs_of_different_sizes():
    """dgl.batch takes graphs of any sizes (nn_module.py:655-661): `pnet_model([pos_a, pos_b, pos_c], [e_a, e_b, e_c])` with 258,
    200 and 258 atoms.  Runs of equal size share a batched engine; the output is the concatenation in list order and equals
    the graphs one by one bit for bit (independent graphs) and the oracle within the fp32 bar."""
    from gamd_amd.compat import ParticleNetLightningLJ
    g, cfg, sd = load_golden("lj258_seed0")
    box, rc = float(g["box"]), float(g["cutoff"])
    rng = np.random.default_rng(21)
    pa = np.mod(g["pos"], box).astype(np.float32)
    pb = np.mod(g["pos"][:200] + rng.normal(0, 0.2, (200, 3)), box).astype(np.float32)
    pc = np.mod(g["pos"] + rng.normal(0, 0.2, (258, 3)), box).astype(np.float32)
    poss = [torch.from_numpy(p) for p in (pa, pb, pc)]
    edges = [orc.neighbor_edges(p, box, rc, "jaxmd") for p in poss]
    m = ParticleNetLightningLJ(state_dict=sd, num_atoms=258, box_size=box, cutoff=rc)
    out = m.pnet_model(poss, edges).cpu().numpy()
    assert out.shape == (258 + 200 + 258, 3)
    assert rel_err(out[:258], g["out_norm"]) < TOL                       # graph 0 is the reference golden
    off = 0
    for p, e in zip(poss, edges):
        one = m.pnet_model([p], [e]).cpu().numpy()
        assert np.array_equal(out[off:off + p.shape[0]], one)
        assert rel_err(one, orc.forward(sd, p, e, box).numpy()) < TOL
        off += p.shape[0]
