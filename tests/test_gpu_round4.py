"""Round-4 parity tests (all through the C ABI):

* exact edge-SET equality with the oracle's blocked O(N^2) search at the FULL sizes of BASELINE configs 2, 3, 5 (both
  readings), in exact mode and in Verlet-skin mode on the positions an on-device MD run of >= 200 steps ends at — the
  oracle is no longer fed the GPU's own edge list there;
* a per-atom error statistic (median / p99 of |df_i| / |f_i|) next to the max-norm bar;
* the generic-width kernels (wide.hip) on the fixed-box models with the trainers' DEFAULT widths 256 / 128 / 256
  (LJ/train_network_lj.py:394-396): jax-md flavour (self edges), bond feature, skin reuse, a short md_run — against
  outputs of the reference's own modules (tests/golden/*_w256_*).
"""
import numpy as np
import pytest
import torch

import gamd_oracle as orc
from helpers import load_golden, rel_err, edge_set, per_atom_err, edge_set_diff_near_cutoff
from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS
from gamd_amd import workloads

pytestmark = pytest.mark.gpu
TOL = 1e-5
P99_TOL = 1e-5          # per-atom |df_i| / |f_i|, 99th percentile, fp32 and split-fp16 paths
NEAR_CUTOFF = 1e-4      # pairs this close to the cutoff may flip between two fp32 searches; nothing else may differ


def _engine(*a, **kw):
    from gamd_amd.engine import GamdForce
    return GamdForce(*a, **kw)


def _workload(name):
    """(state_dict, pos, box, cutoff, species, bonds, scaler, md kwargs, edge dtype) of a BASELINE config, as bench.py builds it."""
    if name == "c2":
        pos, box = workloads.lj_box(10000)
        return (make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2), pos, box, 3.0 * workloads.LJ_SIGMA, None, None,
                SHIPPED_SCALERS["lj"], dict(dt_ps=0.002, mass_amu=39.9), "f32")
    nmol, dens, scal, seed, dtype = {"c3": (1390, 258.0, "tip3p", 2345, "f32"), "c5": (2000, 251.0, "tip4p", 3456, "bf16"),
                                     "c5b": (2667, 251.0, "tip4p", 3456, "bf16")}[name]
    pos, box, species, bonds = workloads.water_box(nmol, mol_per_20A3=dens, seed=seed, jitter=0.0, wrap=False)
    md = dict(dt_ps=0.0005, mass_amu=workloads.MASS_O, mass_h_amu=workloads.MASS_H, rigid_water=True,
              r_oh=workloads.TIP3P_R_OH, r_hh=workloads.TIP3P_R_HH)
    return (make_state_dict(ModelConfig(kind="water", use_bond=True), 3, 2.9, 1.1), pos, box, 4.2, species, bonds,
            SHIPPED_SCALERS[scal], md, dtype)


def _assert_same_edge_set(eng, x_host, box, rc, n, what):
    """GPU CSR vs the oracle's own search on the same (wrapped, fp32) positions."""
    xw = torch.remainder(torch.from_numpy(np.asarray(x_host)).float(), float(box))
    ref = orc.neighbor_edges(xw, box, rc, "jaxmd").numpy()
    got = eng.debug_edges()
    pairs, dist = edge_set_diff_near_cutoff(got, ref, xw.numpy(), box, rc, n)
    assert len(pairs) <= 8 and (len(pairs) == 0 or dist.max() < NEAR_CUTOFF), \
        f"{what}: {len(pairs)} directed pairs differ, the farthest {dist.max() if len(pairs) else 0:.3e} from the cutoff"
    # no duplicates, every atom has its self edge (jax-md flavour, mask_self=False)
    assert len(np.unique(edge_set(got))) == got.shape[1]
    assert int((got[0] == got[1]).sum()) == n
    return got.shape[1], len(pairs)


@pytest.mark.parametrize("name", ["c2", "c3", "c5", "c5b"])
def test_full_size_edge_sets_equal_the_oracles_own_search(name):
    """Exact mode at the start positions, then skin mode (cutoff / 6, the reference's dr_threshold) after 240 MD steps on
    the device (several candidate rebuilds and many reuse steps), compared at the positions the run ends at."""
    sd, pos, box, rc, species, bonds, scaler, md, dtype = _workload(name)
    n = pos.shape[0]
    eng = _engine(sd, n, box, rc, bond=bonds, scaler=scaler, edge_dtype=dtype)
    p = torch.from_numpy(pos).float()
    eng.forward(p, species=species)
    e0, d0 = _assert_same_edge_set(eng, pos, box, rc, n, f"{name} exact mode")
    eng.close()

    eng = _engine(sd, n, box, rc, bond=bonds, scaler=scaler, edge_dtype=dtype, neighbor_skin=rc / 6.0)
    x = p.cuda()
    v = torch.from_numpy(workloads.maxwell_boltzmann(n, mass_amu=md["mass_amu"], temperature_k=300.0, seed=5)).float().cuda()
    f = eng.forward(x, species=species, denormalize=True)
    eng.md_run(x, v, f, 240, temperature_k=300.0, gamma_per_ps=5.0, seed=3, species=species, **md)
    assert torch.isfinite(x).all() and torch.isfinite(f).all()
    rebuilds = eng.skin_stats()[0]
    assert rebuilds >= 2, rebuilds                  # the list really was reused AND rebuilt on the way
    xh = x.cpu().numpy()
    moved = np.abs(xh - pos - float(box) * np.round((xh - pos) / float(box))).max()
    assert moved > 0.2                              # atoms left their start positions
    e1, d1 = _assert_same_edge_set(eng, xh, box, rc, n, f"{name} skin mode after 240 steps ({rebuilds} rebuilds)")
    # the exact rebuild at the same positions gives the same set (and the same count) as the skin path
    eng2 = _engine(sd, n, box, rc, bond=bonds, scaler=scaler, edge_dtype=dtype)
    eng2.forward(x, species=species)
    assert np.array_equal(edge_set(eng2.debug_edges()), edge_set(eng.debug_edges()))
    eng.close(); eng2.close()


@pytest.mark.parametrize("name,edge_dtype", [("c2", "f32"), ("c2", "f16x3"), ("c3", "f32"), ("c3", "f16x3")])
def test_full_size_forces_per_atom_error(name, edge_dtype):
    """The oracle on its OWN edge list (not the GPU's) at the full size; max-norm bar and the per-atom statistic."""
    sd, pos, box, rc, species, bonds, scaler, md, _ = _workload(name)
    n = pos.shape[0]
    eng = _engine(sd, n, box, rc, bond=bonds, scaler=scaler, edge_dtype=edge_dtype)
    p = torch.from_numpy(pos).float()
    out = eng.forward(p, species=species).cpu().numpy()
    pw = torch.remainder(p, float(box))
    edges = orc.neighbor_edges(pw, box, rc, "jaxmd")
    assert edges.shape[1] == eng.counts()[0]
    feat = None if species is None else torch.from_numpy(species.astype(np.float32)).view(-1, 1)
    ref = orc.forward(sd, pw, edges, box, feat=feat, bond=bonds).numpy()
    med, p99, worst, cnt = per_atom_err(out, ref)
    print(f"{name} {edge_dtype}: max-norm {rel_err(out, ref):.2e}; per atom median {med:.2e} p99 {p99:.2e} max {worst:.2e} "
          f"over {cnt} of {n} atoms")
    assert rel_err(out, ref) < TOL
    assert cnt > 0.9 * n and p99 < P99_TOL, (med, p99, worst, cnt)
    eng.close()


# ---- generic-width kernels on the fixed-box models (trainer-default widths) ---------------------------------------
WIDE = ["lj258_w256_seed9", "tip3p774_w256_seed10"]


def _wide_case(name, skin_frac=0.0, widths=(256, 128, 256), **kw):
    g, cfg, sd = load_golden(name)
    assert (cfg.encoding_size, cfg.hidden_dim, cfg.edge_embedding_dim) == widths
    box, rc, n = float(g["box"]), float(g["cutoff"]), g["pos"].shape[0]
    if skin_frac:
        kw["neighbor_skin"] = skin_frac * rc
    bond = g["bond"] if "bond" in g else None
    species = (g["node_feat"].reshape(-1) != 0) if "node_feat" in g else None
    eng = _engine(sd, n, box, rc, bond=bond, scaler=(g["scaler_mean"], g["scaler_var"]), **kw)
    return g, cfg, sd, eng, box, rc, n, bond, species


@pytest.mark.parametrize("name", WIDE)
def test_wide_fixed_box_goldens_stage_by_stage(name):
    _check_stages(*_wide_case(name, keep_stages=True))


# use_layer_norm=False, the model constructors' and the trainers' default: BatchNorm1d between the conv layers (eval mode,
# non-trivial running statistics), folded by gamd_finalize_weights into the node kernels' affine slots; 128- and 256-wide
@pytest.mark.parametrize("name,widths", [("lj258_bn_seed11", (128, 128, 128)), ("tip3p774_bn_w256_seed12", (256, 128, 256))])
@pytest.mark.parametrize("edge_dtype", ["f32", "bf16", "f16x3"])
def test_batchnorm_checkpoints_stage_by_stage(name, widths, edge_dtype):
    case = _wide_case(name, widths=widths, keep_stages=True, edge_dtype=edge_dtype)
    assert not case[1].use_layer_norm and "graph_conv.norm_layers.0.running_var" in case[2]
    _check_stages(*case, tol={"f32": TOL, "bf16": 1e-2, "f16x3": 1e-5}[edge_dtype], stages=edge_dtype == "f32")


def _check_stages(g, cfg, sd, eng, box, rc, n, bond, species, tol=TOL, stages=True):
    posw = np.mod(g["pos"], box).astype(np.float32)
    out = eng.forward(torch.from_numpy(posw), species=species).cpu().numpy()
    edges = eng.debug_edges()
    assert np.array_equal(edge_set(edges), edge_set(g["edge_idx"]))
    assert int((edges[0] == edges[1]).sum()) == n                     # jax-md flavour: self edges
    s = int(g["edge_stride"])
    gkey = g["edge_idx"][0].astype(np.int64) * n + g["edge_idx"][1]
    pos_of = {k: i for i, k in enumerate(edges[0] * n + edges[1])}
    rows = np.array([pos_of[k] for k in gkey[::s]])
    nf = g["feat_rows"].shape[1]
    assert nf == (45 if bond is not None else 44)
    if stages:
        assert rel_err(eng.debug_feat(nf)[rows], g["feat_rows"]) < TOL
        assert rel_err(eng.debug_e()[rows], g["e_rows"]) < TOL
        hs = int(g["h_stride"]) if "h_stride" in g else 1
        for l in range(g["h_layers"].shape[0]):
            assert rel_err(eng.debug_h(l)[::hs], g["h_layers"][l]) < TOL, f"h_{l}"
    assert rel_err(out, g["out_norm"]) < tol
    if tol == TOL:
        med, p99, worst, cnt = per_atom_err(out, g["out_norm"])
        assert p99 < P99_TOL, (med, p99, worst)
    den = eng.forward(torch.from_numpy(posw), species=species, denormalize=True).cpu().numpy()
    assert rel_err(den, g["forces"]) < tol
    eng.close()


@pytest.mark.parametrize("name", WIDE)
def test_wide_fixed_box_skin_reuse_and_md_run(name):
    """Skin mode: the golden forces again, then along a random walk the edge SET equals the exact rebuild's at every step;
    then a short deterministic MD run (T = 0) against the oracle integrator driven by oracle forces."""
    g, cfg, sd, eng, box, rc, n, bond, species = _wide_case(name, skin_frac=1.0 / 6.0)
    exact = _engine(sd, n, box, rc, bond=bond, scaler=(g["scaler_mean"], g["scaler_var"]))
    posw = np.mod(g["pos"], box)
    assert rel_err(eng.forward(torch.from_numpy(posw).float(), species=species).cpu().numpy(), g["out_norm"]) < TOL
    rng = np.random.default_rng(2)
    x = posw.copy()
    for step in range(12):
        x = x + rng.normal(0, 0.04, x.shape)
        a = eng.forward(torch.from_numpy(x).float(), species=species).cpu().numpy()
        b = exact.forward(torch.from_numpy(x).float(), species=species).cpu().numpy()
        assert np.array_equal(edge_set(eng.debug_edges()), edge_set(exact.debug_edges())), step
        assert rel_err(a, b) < TOL
    assert 1 <= eng.skin_stats()[0] < 12                                # reused at least once, rebuilt at least once
    exact.close()
    # md_run, free atoms (the LJ case) / unconstrained per-species masses (the water case): T = 0 -> no noise
    mean, var = g["scaler_mean"], g["scaler_var"]
    feat = torch.from_numpy(g["node_feat"]) if "node_feat" in g else None
    mass = (np.where(species, workloads.MASS_O, workloads.MASS_H).reshape(-1, 1) if species is not None else 39.9)
    xg = torch.from_numpy(posw).float().cuda()
    vg = torch.from_numpy(np.random.default_rng(1).normal(0, 1.0, (n, 3))).float().cuda()
    fg = eng.forward(xg, species=species, denormalize=True).clone()
    xr, vr, fr = xg.cpu().double().numpy(), vg.cpu().double().numpy(), fg.cpu().double().numpy()
    dt, gamma, steps = 0.001, 25.0, 3
    kw = dict(mass_amu=workloads.MASS_O, mass_h_amu=workloads.MASS_H, species=species) if species is not None else dict(mass_amu=39.9)
    eng.md_run(xg, vg, fg, steps, dt_ps=dt, temperature_k=0.0, gamma_per_ps=gamma, **kw)
    a = np.exp(-gamma * dt)
    for _ in range(steps):
        xr, vr = orc.baoab_first_half(xr, vr, fr, 10.0 / mass, dt, a, 0.0, 0.0)
        xr = np.mod(xr, box)
        fr = orc.predict_forces(sd, xr, box, rc, var=var, mean=mean, feat=feat, bond=bond)
        vr = orc.baoab_second_half(vr, fr, 10.0 / mass, dt)
    assert rel_err(xg.cpu().numpy(), xr) < 1e-5
    assert rel_err(vg.cpu().numpy(), vr) < 1e-4
    assert rel_err(fg.cpu().numpy(), fr) < 1e-4
    eng.close()


def test_wide_kernels_match_the_128_wide_ones_with_self_edges_and_bonds():
    """kernel_select = FORCE_GENERIC_WIDTH on the 128-wide TIP3P golden: wide.hip with the jax-md flavour and the bond
    feature reproduces the reference golden too (and the specialised kernels to fp32 rounding)."""
    from gamd_amd.engine import KSEL_FORCE_GENERIC_WIDTH
    g, cfg, sd = load_golden("tip3p774_seed3")
    box, rc, n = float(g["box"]), float(g["cutoff"]), g["pos"].shape[0]
    species = g["node_feat"].reshape(-1) != 0
    posw = torch.from_numpy(np.mod(g["pos"], box)).float()
    outs = []
    for ksel in (0, KSEL_FORCE_GENERIC_WIDTH):
        eng = _engine(sd, n, box, rc, bond=g["bond"], kernel_select=ksel, neighbor_skin=rc / 6.0)
        outs.append(eng.forward(posw, species=species).cpu().numpy())
        eng.close()
    assert rel_err(outs[1], g["out_norm"]) < TOL and rel_err(outs[1], outs[0]) < TOL


def test_f16x3_first_call_on_a_cold_device_matches_f32():
    """the split-fp16 kernel's prologue must wait for its own W1 copy.  The very first launch of a fresh
    engine (cold L2 for the weight image) is compared with the fp32 path, and the run is repeated bit for bit."""
    sd, pos, box, rc, species, bonds, scaler, md, _ = _workload("c2")
    p = torch.from_numpy(pos).float()
    e16 = _engine(sd, 10000, box, rc, scaler=scaler, edge_dtype="f16x3")
    first = e16.forward(p).cpu().numpy().copy()                         # first call of this engine
    again = e16.forward(p).cpu().numpy().copy()
    e32 = _engine(sd, 10000, box, rc, scaler=scaler)
    ref = e32.forward(p).cpu().numpy()
    assert np.array_equal(first, again)
    assert rel_err(first, ref) < TOL
    e16.close(); e32.close()


def test_throughput_and_latency_kernels_agree_bit_for_bit_on_the_reference_snapshot():
    """The host picks the latency-oriented or the throughput kernels by edge count, so they must give the same bits.  On the
    reference's own LJ snapshot they did not before round 4: hipcc fused different multiply-add pairs of the edge-feature code
    in k_edge_encode and k_edge_encode_small (292 of 6 114 feature rows differed in the last bit).  The feature code is now
    compiled without floating-point contraction."""
    for name in ("lj258_seed0", "tip3p774_seed3"):
        g, cfg, sd = load_golden(name)
        box, rc, n = float(g["box"]), float(g["cutoff"]), g["pos"].shape[0]
        bond = g["bond"] if "bond" in g else None
        species = (g["node_feat"].reshape(-1) != 0) if "node_feat" in g else None
        p = torch.from_numpy(np.mod(g["pos"], box)).float()
        res = []
        for lim in (-1, 1000000):
            eng = _engine(sd, n, box, rc, bond=bond, keep_stages=True, small_tile_limit=lim)
            out = eng.forward(p, species=species).cpu().numpy().copy()
            res.append((out, eng.debug_feat(g["feat_rows"].shape[1]), eng.debug_e()))
            eng.close()
        for a, b in zip(res[0], res[1]):
            assert np.array_equal(a, b), name
        assert rel_err(res[0][0], g["out_norm"]) < TOL


# ---- widths other than 128 / 256 (build_model accepts any: nn_module.py:561-601; --hidden_dim, LJ/train_network_lj.py:395) ----
def test_reduced_width_golden_runs_on_the_gpu():
    """tests/golden/lj64_h32: encoding 32 / hidden 32 / edge embedding 32, 2 layers, the reference's own outputs.  The library
    zero-pads to its 128-wide blocks and normalises over the true widths; every stage within 1e-5."""
    g, cfg, sd = load_golden("lj64_h32")
    assert (cfg.encoding_size, cfg.hidden_dim, cfg.edge_embedding_dim, cfg.conv_layer) == (32, 32, 32, 2)
    box, rc, n = float(g["box"]), float(g["cutoff"]), g["pos"].shape[0]
    for kw in (dict(), dict(small_tile_limit=-1), dict(kernel_select=1)):          # latency, throughput and generic-width kernels
        eng = _engine(sd, n, box, rc, scaler=(g["scaler_mean"], g["scaler_var"]), keep_stages=True, **kw)
        posw = np.mod(g["pos"], box).astype(np.float32)
        out = eng.forward(torch.from_numpy(posw)).cpu().numpy()
        edges = eng.debug_edges()
        assert np.array_equal(edge_set(edges), edge_set(g["edge_idx"]))
        s = int(g["edge_stride"])
        gkey = g["edge_idx"][0].astype(np.int64) * n + g["edge_idx"][1]
        pos_of = {k: i for i, k in enumerate(edges[0] * n + edges[1])}
        rows = np.array([pos_of[k] for k in gkey[::s]])
        e = eng.debug_e()
        assert e.shape[1] == 32 and rel_err(e[rows], g["e_rows"]) < TOL, kw
        for l in range(g["h_layers"].shape[0]):
            h = eng.debug_h(l)
            assert h.shape == (n, 32) and rel_err(h, g["h_layers"][l]) < TOL, (kw, l)
        assert rel_err(out, g["out_norm"]) < TOL, kw
        eng.close()


@pytest.mark.parametrize("enc,hid,emb,kind", [(96, 64, 128, "lj"), (128, 100, 200, "water"), (200, 128, 72, "water"), (256, 48, 256, "lj")])
def test_odd_widths_against_the_oracle(enc, hid, emb, kind):
    """Widths that are neither 128 nor 256, mixed: the oracle (pinned by the reduced-width and the wide goldens) on seeded
    weights, through the specialised kernels (enc, emb <= 128) or the generic-width ones."""
    cfg = ModelConfig(kind=kind, encoding_size=enc, hidden_dim=hid, edge_embedding_dim=emb, conv_layer=3, use_bond=kind == "water")
    sd = make_state_dict(cfg, 21, 2.9, 1.1)
    if kind == "water":
        pos, box, species, bonds = workloads.water_box(100, seed=8)
        feat, rc = torch.from_numpy(species.astype(np.float32)).view(-1, 1), 4.2
    else:
        pos, box = workloads.lj_box(300, seed=8)
        species = bonds = feat = None
        rc = 7.5
    n = pos.shape[0]
    eng = _engine(sd, n, box, rc, bond=bonds)
    p = torch.from_numpy(pos).float()
    out = eng.forward(p, species=species).cpu().numpy()
    edges = orc.neighbor_edges(torch.remainder(p, float(box)), box, rc, "jaxmd")
    assert np.array_equal(edge_set(eng.debug_edges()), edge_set(edges.numpy()))
    ref = orc.forward(sd, torch.remainder(p, float(box)), edges, box, feat=feat, bond=bonds).numpy()
    assert rel_err(out, ref) < TOL
    med, p99, worst, cnt = per_atom_err(out, ref)
    assert p99 < P99_TOL, (med, p99, worst)
    eng.close()


def test_dft_configuration_with_skin_reuse_in_the_md_loop():
    """The DFT-water model's reference searches neighbours from scratch every call (md_module.get_neighbor: <=, no self
    edges); the Verlet-skin path applies the same test to a candidate list, so the edge set is the same — checked along a
    rigid-water MD run of the shipped DFT widths against the exact engine and the reference-pinned oracle search."""
    g, cfg, sd = load_golden("dynbox384_dftcfg_seed5")
    n, rc, box = g["pos"].shape[0], float(g["cutoff"]), g["box"]
    species = g["node_feat"].reshape(-1) != 0
    skin = _engine(sd, n, box, rc, nbr_flavour="torch", cfg=cfg, neighbor_skin=rc / 6.0, scaler=(0.0, 25.0))
    exact = _engine(sd, n, box, rc, nbr_flavour="torch", cfg=cfg, scaler=(0.0, 25.0))
    x = torch.from_numpy(g["pos"]).float().cuda()
    assert rel_err(skin.forward(x, box=box, species=species).cpu().numpy(), g["out_norm"]) < TOL
    v = torch.from_numpy(np.random.default_rng(0).normal(0, 3.0, (n, 3))).float().cuda()
    f = skin.forward(x, box=box, species=species, denormalize=True).clone()
    for chunk in range(6):
        skin.md_run(x, v, f, 20, dt_ps=0.001, mass_amu=workloads.MASS_O, mass_h_amu=workloads.MASS_H, temperature_k=300.0,
                    seed=2, first_step=20 * chunk, species=species, box=box)
        ref_f = exact.forward(x, box=box, species=species, denormalize=True).cpu().numpy()
        assert np.array_equal(edge_set(skin.debug_edges()), edge_set(exact.debug_edges())), chunk
        ref_edges = orc.neighbor_edges(torch.remainder(x.cpu(), torch.from_numpy(box)), box, rc, "torch").numpy()
        assert np.array_equal(edge_set(skin.debug_edges()), edge_set(ref_edges)), chunk
        assert rel_err(f.cpu().numpy(), ref_f) < TOL, chunk
    assert skin.skin_stats()[0] >= 2
    skin.close(); exact.close()


# ---- update_edge=True (--update_edge, water/train_network_real_large.py:83,362; nn_module.py:91-92, :140-146) --------------
UPDATE = ["dynbox384_update_dftcfg_seed13", "dynbox384_update_seed14"]


@pytest.mark.parametrize("name", UPDATE)
@pytest.mark.parametrize("small_tile_limit", [0, -1])
def test_update_edge_emb_matches_the_reference_goldens(name, small_tile_limit):
    """Every conv layer hands LayerNorm(e_emb) to the layers after it.  Outputs of WaterMDDynamicBoxNet(update_edge=True)
    itself; both conv kernels (one tile per workgroup / one tile per wave), per-layer node states against the oracle."""
    g, cfg, sd = load_golden(name)
    assert cfg.update_edge and "graph_conv.conv.0.edge_layer_norm.weight" in sd
    n = g["pos"].shape[0]
    eng = _engine(sd, n, g["box"], float(g["cutoff"]), nbr_flavour="torch", keep_stages=True, small_tile_limit=small_tile_limit)
    species = g["node_feat"].reshape(-1) != 0
    out = eng.forward(torch.from_numpy(g["pos"]), box=g["box"], species=species).cpu().numpy()
    assert np.array_equal(edge_set(eng.debug_edges()), edge_set(g["edge_idx"]))
    st = {}
    ref = orc.forward_dynamic_box(sd, torch.from_numpy(g["pos"]), torch.from_numpy(g["node_feat"]), g["box"],
                                  float(g["cutoff"]), stages=st).numpy()
    for l, h_ref in enumerate(st["h"]):
        assert rel_err(eng.debug_h(l), h_ref.numpy()) < TOL, f"h_{l}"
    assert rel_err(out, ref) < TOL
    assert rel_err(out, g["out_norm"]) < TOL
    med, p99, worst, cnt = per_atom_err(out, g["out_norm"])
    assert p99 < P99_TOL, (med, p99, worst)
    # the update is not a no-op: the same weights without the per-layer LayerNorms give other forces
    plain = {k: v for k, v in sd.items() if ".edge_layer_norm." not in k or k.startswith("edge_layer_norm")}
    other = _engine(plain, n, g["box"], float(g["cutoff"]), nbr_flavour="torch")
    assert rel_err(other.forward(torch.from_numpy(g["pos"]), box=g["box"], species=species).cpu().numpy(), g["out_norm"]) > 1e-2
    other.close()
    eng.close()


def test_update_edge_emb_with_skin_reuse_batches_and_odd_widths():
    """Skin reuse and a batch of boxes run the same kernels: equal to the exact / one-by-one evaluation; a width that is
    zero-padded (96) against the oracle; mismatched widths are refused as torch refuses them."""
    g, cfg, sd = load_golden(UPDATE[0])
    n, box, rc = g["pos"].shape[0], g["box"], float(g["cutoff"])
    species = g["node_feat"].reshape(-1) != 0
    exact = _engine(sd, n, box, rc, nbr_flavour="torch")
    skin = _engine(sd, n, box, rc, nbr_flavour="torch", neighbor_skin=rc / 6)
    rng = np.random.default_rng(4)
    x = g["pos"].copy()
    frames = []
    for step in range(6):
        x = (x + rng.normal(0, 0.05, x.shape)).astype(np.float32)
        a = exact.forward(torch.from_numpy(x), box=box, species=species).cpu().numpy()
        b = skin.forward(torch.from_numpy(x), box=box, species=species).cpu().numpy()
        assert rel_err(b, a) < TOL
        frames.append((x.copy(), a))
    batch = _engine(sd, n, box, rc, nbr_flavour="torch", n_boxes=3)
    xb = np.concatenate([f[0] for f in frames[:3]])
    ob = batch.forward(torch.from_numpy(xb), box=np.tile(np.asarray(box, dtype=np.float32), (3, 1)),
                       species=np.tile(species, 3)).cpu().numpy()
    assert np.array_equal(ob, np.concatenate([f[1] for f in frames[:3]]))          # bit-identical to the boxes one by one
    for e in (exact, skin, batch):
        e.close()
    c96 = ModelConfig(kind="dynbox", update_edge=True, encoding_size=96, hidden_dim=64, edge_embedding_dim=96, conv_layer=3)
    sd96 = make_state_dict(c96, 21, 3.1, 1.2)
    eng = _engine(sd96, n, box, rc, nbr_flavour="torch")
    out = eng.forward(torch.from_numpy(g["pos"]), box=box, species=species).cpu().numpy()
    ref = orc.forward_dynamic_box(sd96, torch.from_numpy(g["pos"]), torch.from_numpy(g["node_feat"]), box, rc).numpy()
    assert rel_err(out, ref) < TOL
    eng.close()
    bad = make_state_dict(ModelConfig(kind="dynbox", update_edge=True, encoding_size=128, edge_embedding_dim=256, conv_layer=2), 1)
    from gamd_amd._lib import GamdError
    with pytest.raises((GamdError, ValueError)):
        _engine(bad, n, box, rc, nbr_flavour="torch")


# ---- split-fp16 edge MLP for every width (wide_lp.hip + the generic-width encoder writing split operands) -----------------
@pytest.mark.parametrize("name", ["lj258_w256_seed9", "tip3p774_w256_seed10", "tip3p774_bn_w256_seed12", "dynbox384_dftcfg_seed5",
                                  "dynbox384_h256_e128_seed7", "dynbox384_h128_e256_noexpand_seed8", "dynbox384_noexpand_seed6",
                                  "lj64_h32"])
def test_split_fp16_edge_mlp_on_the_generic_width_goldens(name):
    """edge_dtype="f16x3" outside 128 / 128 / 128: trainers' default widths on the fixed-box models, the DFT-water configuration,
    mixed widths, expand_edge=False, a zero-padded narrow model -- outputs of the reference modules at the fp32 bar (1e-5),
    per-atom p99 included; fp32 engine alongside (the two differ by rounding only).  edge_dtype="bf16" (wide_lp.hip) on the
    same cases at config 5's restated tolerance (1e-2)."""
    g, cfg, sd = load_golden(name)
    n = g["pos"].shape[0]
    dyn = cfg.kind == "dynbox"
    box = g["box"] if dyn else float(g["box"])
    kw = dict(nbr_flavour="torch") if dyn else dict(bond=g["bond"] if "bond" in g else None)
    if "scaler_mean" in g:
        kw["scaler"] = (g["scaler_mean"], g["scaler_var"])
    species = (g["node_feat"].reshape(-1) != 0) if "node_feat" in g else None
    pos = torch.from_numpy(g["pos"] if dyn else np.mod(g["pos"], box)).float()
    outs = {}
    for dt in ("f16x3", "f32", "bf16"):
        eng = _engine(sd, n, box, float(g["cutoff"]), edge_dtype=dt, **kw)
        outs[dt] = (eng.forward(pos, box=box, species=species) if dyn else eng.forward(pos, species=species)).cpu().numpy()
        if dt != "f32":
            assert np.array_equal(edge_set(eng.debug_edges()), edge_set(g["edge_idx"]))
            again = (eng.forward(pos, box=box, species=species) if dyn else eng.forward(pos, species=species)).cpu().numpy()
            assert np.array_equal(again, outs[dt])                      # run-to-run bit identity
        eng.close()
    assert rel_err(outs["bf16"], g["out_norm"]) < 1e-2 and rel_err(outs["bf16"], outs["f32"]) > 1e-6      # bf16 really ran
    assert rel_err(outs["f16x3"], g["out_norm"]) < TOL
    med, p99, worst, cnt = per_atom_err(outs["f16x3"], g["out_norm"])
    assert p99 < P99_TOL, (med, p99, worst)
    assert rel_err(outs["f16x3"], outs["f32"]) < TOL


def test_split_fp16_generic_width_with_skin_batches_and_large_boxes():
    """The generic-width split-fp16 kernels under skin reuse, in a batch (bit-identical to the boxes one by one) and on a box
    large enough for several tiles per wave (trainers' default widths on the C2 workload's box at 4 000 atoms, against the fp32
    engine: the reference modules cannot run that size here in reasonable time, the fp32 kernels are pinned at it)."""
    g, cfg, sd = load_golden("tip3p774_w256_seed10")
    box, rc, n = float(g["box"]), float(g["cutoff"]), g["pos"].shape[0]
    species = g["node_feat"].reshape(-1) != 0
    scal = (g["scaler_mean"], g["scaler_var"])
    one = _engine(sd, n, box, rc, bond=g["bond"], scaler=scal, edge_dtype="f16x3")
    skin = _engine(sd, n, box, rc, bond=g["bond"], scaler=scal, edge_dtype="f16x3", neighbor_skin=rc / 6)
    rng = np.random.default_rng(8)
    x = np.mod(g["pos"], box)
    frames = []
    for step in range(5):
        x = x + rng.normal(0, 0.04, x.shape)
        a = one.forward(torch.from_numpy(x).float(), species=species).cpu().numpy()
        b = skin.forward(torch.from_numpy(x).float(), species=species).cpu().numpy()
        assert rel_err(b, a) < TOL
        frames.append((x.copy(), a))
    batch = _engine(sd, n, box, rc, bond=g["bond"], scaler=scal, edge_dtype="f16x3", n_boxes=3)
    xb = np.concatenate([f[0] for f in frames[:3]])
    ob = batch.forward(torch.from_numpy(xb).float(), species=np.tile(species, 3)).cpu().numpy()
    assert np.array_equal(ob, np.concatenate([f[1] for f in frames[:3]]))
    for e in (one, skin, batch):
        e.close()
    # the same in bf16 (wide_lp.hip): skin == exact to the restated tolerance (identical edge sets; row order may differ after a
    # rebuild), batch == boxes one by one bit for bit
    oneb = _engine(sd, n, box, rc, bond=g["bond"], scaler=scal, edge_dtype="bf16")
    batchb = _engine(sd, n, box, rc, bond=g["bond"], scaler=scal, edge_dtype="bf16", n_boxes=3)
    singles = [oneb.forward(torch.from_numpy(f[0]).float(), species=species).cpu().numpy() for f in frames[:3]]
    obb = batchb.forward(torch.from_numpy(xb).float(), species=np.tile(species, 3)).cpu().numpy()
    assert np.array_equal(obb, np.concatenate(singles))
    assert rel_err(singles[0], frames[0][1]) < 1e-2
    oneb.close(); batchb.close()
    pos, lbox = workloads.lj_box(4000)
    wide = make_state_dict(ModelConfig(kind="lj", encoding_size=256, hidden_dim=128, edge_embedding_dim=256), 3, 7.0, 2.2)
    f32 = _engine(wide, 4000, lbox, 3.0 * workloads.LJ_SIGMA)
    f16 = _engine(wide, 4000, lbox, 3.0 * workloads.LJ_SIGMA, edge_dtype="f16x3")
    xa = torch.from_numpy(pos).float()
    ref, got = f32.forward(xa).cpu().numpy(), f16.forward(xa).cpu().numpy()
    assert rel_err(got, ref) < TOL
    med, p99, worst, cnt = per_atom_err(got, ref)
    assert p99 < P99_TOL, (med, p99, worst)
    assert np.array_equal(got, f16.forward(xa).cpu().numpy())          # run-to-run bit identity
    b16 = _engine(wide, 4000, lbox, 3.0 * workloads.LJ_SIGMA, edge_dtype="bf16")
    assert rel_err(b16.forward(xa).cpu().numpy(), ref) < 1e-2
    f32.close(); f16.close(); b16.close()


# ---- the feature matrix, sampled: widths x model flavour x normalisation x update_edge x expand_edge x edge dtype ------------------
def _matrix_cases():
    rng = np.random.default_rng(2026)
    cases = []
    for i in range(14):
        kind = ["lj", "water", "dynbox"][i % 3]
        upd = kind == "dynbox" and i % 2 == 0
        enc = int(rng.choice([24, 64, 100, 128, 160, 256]))
        emb = enc if upd else int(rng.choice([32, 72, 128, 200, 256]))
        cases.append(dict(kind=kind, encoding_size=enc, edge_embedding_dim=emb, hidden_dim=int(rng.choice([16, 64, 96, 128])),
                          conv_layer=int(rng.integers(1, 6)), use_bond=kind == "water" and i % 4 != 1,
                          n_rbf=0 if (kind == "dynbox" and i % 3 == 0) else 40, use_layer_norm=bool(i % 5 != 2), update_edge=upd,
                          edge_dtype="f16x3" if i % 2 else "f32", skin=bool(i % 3 == 1), seed=100 + i))
    return cases


@pytest.mark.parametrize("case", _matrix_cases(), ids=lambda c: "{kind}-{encoding_size}-{hidden_dim}-{edge_embedding_dim}-L{conv_layer}-{edge_dtype}".format(**c))
def test_sampled_feature_matrix_against_the_oracle(case):
    """Fourteen seeded combinations of everything a state_dict / constructor can select (any width up to 256 / 128 / 256, 1-5
    layers, LayerNorm or BatchNorm, update_edge, expand_edge on / off, bond feature, the three model flavours, fp32 or
    split-fp16 edge MLP, exact or skin neighbour mode) against the oracle, whose branches are each pinned by a
    reference-generated golden (tests/test_oracle_golden.py): forces at the fp32 bar, per-atom p99 included."""
    c = dict(case)
    edge_dtype, skin, seed = c.pop("edge_dtype"), c.pop("skin"), c.pop("seed")
    cfg = ModelConfig(**c)
    sd = make_state_dict(cfg, seed, 2.9, 1.1)
    if cfg.kind == "lj":
        pos, box = workloads.lj_box(260, seed=seed)
        species = bonds = feat = None
        rc, flavour = 7.5, "jaxmd"
    else:
        pos, box, species, bonds = workloads.water_box(90, seed=seed)
        feat, rc = torch.from_numpy(species.astype(np.float32)).view(-1, 1), 4.2
        flavour = "torch" if cfg.kind == "dynbox" else "jaxmd"
        if not cfg.use_bond:
            bonds = None
    n = pos.shape[0]
    eng = _engine(sd, n, box, rc, bond=bonds, nbr_flavour=flavour, edge_dtype=edge_dtype, neighbor_skin=rc / 6 if skin else 0.0)
    p = torch.remainder(torch.from_numpy(pos).float(), float(box))
    out = eng.forward(p, species=species).cpu().numpy()
    if skin:                                   # a second call on moved positions reuses the candidate list
        p = torch.remainder(p + 0.03 * torch.from_numpy(np.random.default_rng(seed).normal(size=pos.shape)).float(), float(box))
        out = eng.forward(p, species=species).cpu().numpy()
        assert eng.skin_stats()[0] == 1
    edges = orc.neighbor_edges(p, box, rc, flavour)
    assert np.array_equal(edge_set(eng.debug_edges()), edge_set(edges.numpy()))
    if cfg.kind == "dynbox":
        ref = orc.forward_dynamic_box(sd, p, feat, np.full(3, box, dtype=np.float32), rc).numpy()
    else:
        ref = orc.forward(sd, p, edges, box, feat=feat, bond=bonds).numpy()
    assert rel_err(out, ref) < TOL, case
    med, p99, worst, cnt = per_atom_err(out, ref)
    assert p99 < P99_TOL, (case, med, p99, worst)
    eng.close()


# ---- the reduced-precision kernels at the edges of their input space --------------------------------------------------------------
@pytest.mark.parametrize("edge_dtype,widths", [("f16x3", (128, 128, 128)), ("bf16", (128, 128, 128)), ("f16x3", (256, 128, 256)),
                                               ("bf16", (256, 128, 256)), ("f16x3", (96, 64, 160))])
def test_reduced_precision_kernels_on_sparse_tiny_and_overflowing_inputs(edge_dtype, widths):
    """What the frozen-flag / tail / padding paths of the two-wave kernels see: (a) a gas with isolated atoms and rows shorter than
    a chunk, edge capacity far too small (detected on device, regrown, retried: status 1); (b) no edge at all (E = 0: every
    wave idles through the barriers); (c) fewer atoms than one tile; (d) an overflow in the middle of an enqueued MD run
    (freeze, regrow, resume).  Forces against the fp32 engine at the mode's tolerance, bit-reproducible run to run."""
    tol = TOL if edge_dtype == "f16x3" else 1e-2
    enc, hid, emb = widths
    cfg = ModelConfig(kind="water", encoding_size=enc, hidden_dim=hid, edge_embedding_dim=emb, conv_layer=3)
    sd = make_state_dict(cfg, 9, 2.0, 0.7)
    rng = np.random.default_rng(5)
    # (a) sparse gas + regrow
    n, box, rc = 128, 16.0, 3.0
    pos = torch.from_numpy(rng.uniform(0, box, (n, 3))).float()
    species = np.arange(n) % 3 == 0
    ref = _engine(sd, n, box, rc, nbr_flavour="torch")
    want = ref.forward(pos, species=species).cpu().numpy()
    eng = _engine(sd, n, box, rc, nbr_flavour="torch", edge_capacity=40, edge_dtype=edge_dtype)
    out = eng.forward(pos, species=species).cpu().numpy()
    assert eng.last_status == 1 and np.array_equal(edge_set(eng.debug_edges()), edge_set(ref.debug_edges()))
    assert (np.bincount(eng.debug_edges()[0], minlength=n) == 0).any()
    assert rel_err(out, want) < tol
    assert np.array_equal(out, eng.forward(pos, species=species).cpu().numpy()) and eng.last_status == 0
    # (b) no edges at all: a 3 x 3 x 3 lattice of spacing 4 in a box of 12, cutoff 3 (no self edges in this flavour)
    lat = torch.tensor([[4.0 * i + 0.5, 4.0 * j + 0.5, 4.0 * k + 0.5] for i in range(3) for j in range(3) for k in range(3)])
    e0, r0 = (_engine(sd, 27, 12.0, 3.0, nbr_flavour="torch", edge_dtype=dt) for dt in (edge_dtype, "f32"))
    o0 = e0.forward(lat, species=species[:27]).cpu().numpy()
    assert e0.counts()[0] == 0 and np.isfinite(o0).all() and rel_err(o0, r0.forward(lat, species=species[:27]).cpu().numpy()) < tol
    # (c) fewer atoms than one 32-edge tile has slots
    p7 = pos[:7] * 0.2
    e7, r7 = (_engine(sd, 7, 16.0, rc, nbr_flavour="torch", edge_dtype=dt) for dt in (edge_dtype, "f32"))
    o7 = e7.forward(p7, species=species[:7]).cpu().numpy()
    assert 0 < e7.counts()[0] < 64 and rel_err(o7, r7.forward(p7, species=species[:7]).cpu().numpy()) < tol
    for e in (ref, eng, e0, r0, e7, r7):
        e.close()
    # (d) overflow inside an enqueued run: equal to the ample-buffer run bit for bit (exact rebuild every step)
    nl, steps = 1500, 8
    sdl = make_state_dict(ModelConfig(kind="lj", encoding_size=enc, hidden_dim=hid, edge_embedding_dim=emb, conv_layer=2), 2, 5.0, 1.7)
    posl, boxl = workloads.lj_box(nl, seed=4)
    res = []
    for cap in (0, 4000):
        x = torch.from_numpy(posl).float().cuda()
        v = torch.from_numpy(workloads.maxwell_boltzmann(nl, 300.0, seed=3)).float().cuda()
        big = _engine(sdl, nl, boxl, 7.5, scaler=SHIPPED_SCALERS["lj"], edge_dtype=edge_dtype)
        f = big.forward(x, denormalize=True).clone()
        big.close()
        eng = _engine(sdl, nl, boxl, 7.5, scaler=SHIPPED_SCALERS["lj"], edge_dtype=edge_dtype, edge_capacity=cap)
        eng.md_run(x, v, f, steps, seed=11, sync=False)
        assert eng.sync_status() == (1 if cap else 0)
        res.append((x.cpu().numpy(), v.cpu().numpy(), f.cpu().numpy()))
        eng.close()
    for a, b in zip(*res):
        assert np.isfinite(a).all() and np.array_equal(a, b)


@pytest.mark.parametrize("name", ["lj258_seed0", "tip3p774_w256_seed10", "dynbox384_h128_e256_noexpand_seed8"])
def test_debug_getter_reads_the_operand_form_edge_embeddings(name):
    """debug_e on the reduced-precision engines: the encoder's output in operand form (split-fp16 pairs / bf16 fragments, one or
    two 128-blocks) de-fragmented by the host equals the fp32 engine's e to the format's precision."""
    g, cfg, sd = load_golden(name)
    n = g["pos"].shape[0]
    dyn = cfg.kind == "dynbox"
    box = g["box"] if dyn else float(g["box"])
    kw = dict(nbr_flavour="torch") if dyn else dict(bond=g["bond"] if "bond" in g else None)
    species = (g["node_feat"].reshape(-1) != 0) if "node_feat" in g else None
    pos = torch.from_numpy(g["pos"] if dyn else np.mod(g["pos"], box)).float()
    es = {}
    for dt in ("f32", "f16x3", "bf16"):
        eng = _engine(sd, n, box, float(g["cutoff"]), edge_dtype=dt, keep_stages=True, **kw)
        eng.forward(pos, box=box, species=species) if dyn else eng.forward(pos, species=species)
        es[dt] = eng.debug_e()
        eng.close()
    assert es["f32"].shape == (len(g["edge_idx"][0]), cfg.edge_embedding_dim)
    assert rel_err(es["f16x3"], es["f32"]) < 5e-6
    assert 1e-5 < rel_err(es["bf16"], es["f32"]) < 1e-2
