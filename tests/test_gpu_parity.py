"""Parity tests proper: the HIP path, called through the C ABI, against
(a) the committed outputs of the reference's own modules (tests/golden), and
(b) the CPU oracle on seeded inputs, plus size-independent properties at the full C2 size.

Tolerance: BASELINE.json north_star — forces within 1e-5 relative (fp32), relative =
max|a-b| / max|b|.  Integer outputs (edge sets) are compared exactly.
"""
import numpy as np
import pytest
import torch

import gamd_oracle as orc
from helpers import load_golden, rel_err, edge_set
from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS
from gamd_amd import workloads

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _engine(*a, **kw):
    from gamd_amd.engine import GamdForce
    return GamdForce(*a, **kw)


@pytest.mark.parametrize("edge_dtype", ["f32", "f16x3"])
@pytest.mark.parametrize("name", ["lj258_seed0", "lj258_pert_seed1", "tip3p774_seed3"])
def test_golden_stages_and_forces(name, edge_dtype):
    """Every stage against the reference's own outputs, 1e-5 relative.  edge_dtype "f32": fp32 MFMA (bit-exact fp32
    FMAs); "f16x3": the same GEMMs on the fp16 matrix pipe with every operand split into hi + lo fp16
    (W x = Wh xh + Wh xl + Wl xh, fp32 accumulate) -- same goldens, same bar."""
    g, cfg, sd = load_golden(name)
    box, rc, n = float(g["box"]), float(g["cutoff"]), g["pos"].shape[0]
    bond = g["bond"] if "bond" in g else None
    eng = _engine(sd, n, box, rc, bond=bond, scaler=(g["scaler_mean"], g["scaler_var"]), keep_stages=True,
                  edge_dtype=edge_dtype)
    posw = np.mod(g["pos"], box).astype(np.float32)
    species = (g["node_feat"].reshape(-1) != 0) if "node_feat" in g else None
    out = eng.forward(torch.from_numpy(posw), species=species).cpu().numpy()
    # integer work: bit-exact edge set
    edges = eng.debug_edges()
    assert edges.shape[1] == g["edge_idx"].shape[1]
    assert np.array_equal(edge_set(edges), edge_set(g["edge_idx"]))
    # stage tensors against the reference's, matched through the edge keys
    s = int(g["edge_stride"])
    gkey = g["edge_idx"][0].astype(np.int64) * n + g["edge_idx"][1]
    hkey = edges[0] * n + edges[1]
    pos_of = {k: i for i, k in enumerate(hkey)}
    rows = np.array([pos_of[k] for k in gkey[::s]])
    nf = g["feat_rows"].shape[1]
    assert rel_err(eng.debug_feat(nf)[rows], g["feat_rows"]) < TOL
    assert rel_err(eng.debug_e()[rows], g["e_rows"]) < TOL
    if "h_layers" in g:
        for l in range(g["h_layers"].shape[0]):
            assert rel_err(eng.debug_h(l), g["h_layers"][l]) < TOL, f"h_{l}"
    assert rel_err(out, g["out_norm"]) < TOL
    den = eng.forward(torch.from_numpy(posw), species=species, denormalize=True).cpu().numpy()
    assert rel_err(den, g["forces"]) < TOL
    eng.close()


def test_compat_predict_forces_matches_reference_api():
    """ParticleNetLightning.predict_forces contract: np f64 in -> np f64 out, denormalised."""
    from gamd_amd.compat import ParticleNetLightningLJ, ParticleNetLightningWater
    g, cfg, sd = load_golden("lj258_pert_seed1")
    m = ParticleNetLightningLJ(state_dict=sd).cuda().eval()
    m.training_mean, m.training_var = g["scaler_mean"], g["scaler_var"]
    f = m.predict_forces(g["pos"])                       # un-wrapped f64 positions, like the driver
    assert isinstance(f, np.ndarray) and f.dtype == np.float64 and f.shape == (258, 3)
    assert rel_err(f, g["forces"]) < TOL
    f2 = m.predict_forces(g["pos"])                      # fresh array each call, stable result
    assert f2 is not f and np.array_equal(f, f2)
    g, cfg, sd = load_golden("tip3p774_seed3")
    w = ParticleNetLightningWater(state_dict=sd)
    w.training_mean, w.training_var = g["scaler_mean"], g["scaler_var"]
    fw = w.predict_forces(torch.from_numpy(g["node_feat"]).cuda(), g["pos"])
    assert fw.dtype == np.float64 and rel_err(fw, g["forces"]) < TOL
    # the opt-in split-fp16 GEMMs through the same wrapper: same golden, same bar
    w16 = ParticleNetLightningWater(state_dict=sd, edge_dtype="f16x3")
    w16.training_mean, w16.training_var = g["scaler_mean"], g["scaler_var"]
    assert rel_err(w16.predict_forces(torch.from_numpy(g["node_feat"]).cuda(), g["pos"]), g["forces"]) < TOL


def test_compat_dft_predict_forces_takes_the_box_per_call():
    """DFT-water wrapper (water/train_network_real_large.py:148-162): predict_forces(feat, pos, box_size) with the
    shipped widths (256/256/128, 5 layers), and the model-level call pnet_model([pos], feat, [box], cutoff)."""
    from gamd_amd.compat import ParticleNetLightningDFT
    from types import SimpleNamespace
    g, cfg, sd = load_golden("dynbox384_dftcfg_seed5")
    n = g["pos"].shape[0]
    m = ParticleNetLightningDFT(SimpleNamespace(cutoff=float(g["cutoff"])), sd, num_atoms=n).cuda().eval()
    mean, var = SHIPPED_SCALERS["dft"]
    m.training_mean, m.training_var = mean, var
    feat = torch.from_numpy(g["node_feat"]).cuda()
    shift = np.array([g["box"][0], -2 * g["box"][1], 0.0])            # un-wrapped input: np.mod happens inside
    f = m.predict_forces(feat, g["pos"].astype(np.float64) + shift, g["box"])
    assert f.dtype == np.float64 and f.shape == (n, 3)
    assert rel_err(f, g["out_norm"].astype(np.float64) * np.sqrt(var) + mean) < TOL
    out = m.pnet_model([torch.from_numpy(g["pos"])], feat, [g["box"]], float(g["cutoff"])).cpu().numpy()
    assert rel_err(out, g["out_norm"]) < TOL
    with pytest.raises(ValueError):
        m.pnet_model([torch.from_numpy(g["pos"])], feat, [g["box"]], 3.0)
    # a different box on the next call (NPT-style): matches the oracle on that box
    box2 = g["box"] * np.float32(1.03)
    pos2 = (g["pos"] * np.float32(1.03)).astype(np.float32)
    f2 = m.predict_forces(feat, pos2.astype(np.float64), box2)
    ref2 = orc.forward_dynamic_box(sd, torch.from_numpy(pos2), torch.from_numpy(g["node_feat"]), box2,
                                   float(g["cutoff"])).numpy().astype(np.float64) * np.sqrt(var) + mean
    assert rel_err(f2, ref2) < TOL


@pytest.mark.parametrize("n,box,rc,flavour", [
    (300, 30.0, 7.5, "jaxmd"),           # 4 cells / axis
    (200, 14.0, 6.0, "jaxmd"),           # 2 cells / axis  (all-cells sweep)
    (64, 9.0, 5.0, "jaxmd"),             # 1 cell / axis, box < 2 rc
    (500, (20.0, 31.0, 12.5), 5.5, "jaxmd"),   # orthorhombic, mixed cell counts 3/5/2
    (384, (20.0, 21.0, 22.5), 4.6, "torch"),   # dynamic-box flavour: <=, no self
    (97, 11.0, 2.0, "torch"),            # sparse: isolated atoms exist
])
def test_neighbor_sets_match_oracle(n, box, rc, flavour):
    rng = np.random.default_rng(n)
    b = np.broadcast_to(np.asarray(box, dtype=np.float64), (3,))
    pos = rng.uniform(-0.5, 1.5, (n, 3)) * b            # includes images outside the box
    sd = make_state_dict(ModelConfig(kind="lj"), 0)
    eng = _engine(sd, n, box, rc, nbr_flavour=flavour)
    p32 = torch.from_numpy(pos).float()
    eng.build_neighbors(p32)
    edges = eng.debug_edges()
    ref = orc.neighbor_edges(torch.remainder(p32, torch.from_numpy(b).float()), box, rc, flavour)
    ref = ref.numpy() if flavour == "jaxmd" else ref.numpy()
    d = orc._min_image(p32[:, None] - p32[None], orc._box_tensor(box)).norm(dim=-1)
    assert float((d - rc).abs().min()) > 1e-5          # fixture keeps clear of the cutoff
    assert np.array_equal(edge_set(edges), edge_set(ref))
    row_ptr, col = eng.debug_csr()
    assert row_ptr[0] == 0 and row_ptr[-1] == edges.shape[1] and np.all(np.diff(row_ptr) >= 0)
    has_self = np.any(edges[0] == edges[1])
    assert has_self == (flavour == "jaxmd")
    eng.close()


def test_dynamic_box_flavour_matches_reference_golden():
    """WaterMDDynamicBoxNet path (nn_module.py:391-407): '<=' cutoff, no self edges, orthorhombic box
    passed per call."""
    g, cfg, sd = load_golden("dynbox384_seed4")
    n = g["pos"].shape[0]
    eng = _engine(sd, n, g["box"], float(g["cutoff"]), nbr_flavour="torch",
                  cfg=ModelConfig(kind="water", use_bond=False))
    species = g["node_feat"].reshape(-1) != 0
    out = eng.forward(torch.from_numpy(g["pos"]), box=g["box"], species=species).cpu().numpy()
    edges = eng.debug_edges()
    assert np.array_equal(edge_set(edges), edge_set(g["edge_idx"]))
    assert rel_err(out, g["out_norm"]) < TOL
    eng.close()


@pytest.mark.parametrize("name", ["dynbox384_dftcfg_seed5", "dynbox384_noexpand_seed6", "dynbox384_h256_e128_seed7",
                                  "dynbox384_h128_e256_noexpand_seed8"])
def test_wide_and_unexpanded_configs_match_reference_golden(name):
    """The DFT-water configuration of WaterMDDynamicBoxNet (encoding 256 / edge embedding 256 / hidden 128, 5 layers,
    water/test_script/test_nosehoover_hb.py:69-81), expand_edge=False (nn_module.py:333-335) and mixed widths:
    the generic-width kernels (csrc/wide.hip) against outputs of the reference module."""
    g, cfg, sd = load_golden(name)
    n = g["pos"].shape[0]
    eng = _engine(sd, n, g["box"], float(g["cutoff"]), nbr_flavour="torch", cfg=cfg, keep_stages=True)
    species = g["node_feat"].reshape(-1) != 0
    out = eng.forward(torch.from_numpy(g["pos"]), box=g["box"], species=species).cpu().numpy()
    edges = eng.debug_edges()
    assert np.array_equal(edge_set(edges), edge_set(g["edge_idx"]))
    # per-stage check against the oracle (itself pinned to this golden in test_oracle_golden.py)
    st = {}
    ref = orc.forward_dynamic_box(sd, torch.from_numpy(g["pos"]), torch.from_numpy(g["node_feat"]), g["box"],
                                  float(g["cutoff"]), stages=st).numpy()
    key = lambda e: np.asarray(e[0]).astype(np.int64) * n + np.asarray(e[1]).astype(np.int64)
    o_dev, o_ref = np.argsort(key(edges), kind="stable"), np.argsort(key(st["edge_idx"].numpy()), kind="stable")
    assert rel_err(eng.debug_feat(cfg.edge_in)[o_dev], st["feat"].numpy()[o_ref]) < TOL
    assert rel_err(eng.debug_e()[o_dev], st["e"].numpy()[o_ref]) < TOL
    for l, h_ref in enumerate(st["h"]):
        assert rel_err(eng.debug_h(l), h_ref.numpy()) < TOL, f"h_{l}"
    assert rel_err(out, ref) < TOL
    assert rel_err(out, g["out_norm"]) < TOL
    eng.close()


def test_wide_kernels_reproduce_the_128_wide_path():
    """kernel_select = GAMD_KSEL_FORCE_GENERIC_WIDTH routes the shipped 128-wide configuration through csrc/wide.hip:
    same goldens, same bar."""
    g, cfg, sd = load_golden("tip3p774_seed3")
    box, rc, n = float(g["box"]), float(g["cutoff"]), g["pos"].shape[0]
    posw = torch.from_numpy(np.mod(g["pos"], box).astype(np.float32))
    species = g["node_feat"].reshape(-1) != 0
    eng = _engine(sd, n, box, rc, bond=g["bond"], scaler=(g["scaler_mean"], g["scaler_var"]), kernel_select=1)
    out = eng.forward(posw, species=species).cpu().numpy()
    eng.close()
    assert rel_err(out, g["out_norm"]) < TOL
    eng = _engine(sd, n, box, rc, bond=g["bond"], scaler=(g["scaler_mean"], g["scaler_var"]))
    out0 = eng.forward(posw, species=species).cpu().numpy()
    eng.close()
    assert rel_err(out, out0) < TOL


def test_isolated_atoms_and_regrow():
    """zero in-degree atoms aggregate to 0 (torch flavour) and a too-small edge capacity is detected
    on device, regrown and retried (status 1), like jax-md's buffer overflow (graph_utils.py:41-42)."""
    rng = np.random.default_rng(5)
    n, box, rc = 128, 16.0, 3.0
    pos = rng.uniform(0, box, (n, 3))
    cfg = ModelConfig(kind="water")
    sd = make_state_dict(cfg, 9, 2.0, 0.7)
    eng = _engine(sd, n, box, rc, nbr_flavour="torch", edge_capacity=40)
    species = (np.arange(n) % 3 == 0)
    p = torch.from_numpy(pos).float()
    out = eng.forward(p, species=species).cpu().numpy()
    assert eng.last_status == 1 and eng.counts()[2] >= eng.counts()[0]
    edges = torch.from_numpy(eng.debug_edges()).long()
    deg = np.bincount(edges[0].numpy(), minlength=n)
    assert (deg == 0).any()
    feat = torch.from_numpy(species.astype(np.float32)).view(-1, 1)
    ref = orc.forward(sd, p, edges, box, feat=feat).numpy()
    assert rel_err(out, ref) < TOL
    out2 = eng.forward(p, species=species).cpu().numpy()
    assert eng.last_status == 0 and np.array_equal(out, out2)      # bit-reproducible
    eng.close()


@pytest.fixture(scope="module")
def c2():
    pos, box = workloads.lj_box(10000)
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2)
    eng = _engine(sd, 10000, box, 3.0 * workloads.LJ_SIGMA, scaler=SHIPPED_SCALERS["lj"])
    yield eng, sd, pos, box
    eng.close()


def test_c2_full_size_against_oracle(c2):
    """BASELINE config 2 (10 000 atoms, cutoff 3 sigma) against the CPU oracle on the same edges."""
    eng, sd, pos, box = c2
    p = torch.from_numpy(pos).float()
    out = eng.forward(p).cpu().numpy()
    edges = eng.debug_edges()
    assert 55 * 10000 < edges.shape[1] < 70 * 10000
    ref = orc.forward(sd, p, torch.from_numpy(edges).long(), box).numpy()
    assert rel_err(out, ref) < TOL


def test_c2_properties(c2):
    """Size-independent properties at the full size: determinism, invariance under whole-box
    translations by lattice vectors, equivariance under atom permutation."""
    eng, sd, pos, box = c2
    p = torch.from_numpy(pos).float()
    a = eng.forward(p).cpu().numpy().copy()
    e0 = eng.counts()[0]
    b = eng.forward(p).cpu().numpy().copy()
    assert np.array_equal(a, b)                                   # bit-reproducible, no atomics
    # other periodic images: the fp32 cast of the shifted coordinates moves atoms by ~1e-5 A, so a
    # handful of the 6e5 pairs that sit within that distance of the cutoff may flip membership
    shifted = pos + np.array([box, -box, 0.0])
    c = eng.forward(torch.from_numpy(shifted).float()).cpu().numpy().copy()
    assert abs(eng.counts()[0] - e0) <= 16
    # an edge flipping at the cutoff changes its two atoms (and their neighbourhoods) by O(1e-2):
    # the model is not smooth there.  Everything else must agree to fp32 accuracy.
    per_atom = np.abs(c - a).max(axis=1) / np.abs(a).max()
    assert np.mean(per_atom < 1e-4) > 0.98 and np.median(per_atom) < 1e-5
    perm = np.random.default_rng(0).permutation(10000)
    d = eng.forward(torch.from_numpy(pos[perm]).float()).cpu().numpy().copy()
    assert eng.counts()[0] == e0 and rel_err(d, a[perm]) < TOL


def test_c3_water_full_size_against_oracle():
    """BASELINE config 3: 1 390 TIP3P molecules (4 170 atoms), bonds + species."""
    pos, box, species, bonds = workloads.water_box(1390)
    cfg = ModelConfig(kind="water", use_bond=True)
    sd = make_state_dict(cfg, 3, 2.9, 1.1)
    eng = _engine(sd, pos.shape[0], box, 4.2, bond=bonds, scaler=SHIPPED_SCALERS["tip3p"])
    p = torch.from_numpy(pos).float()
    out = eng.forward(p, species=species).cpu().numpy()
    edges = torch.from_numpy(eng.debug_edges()).long()
    feat = torch.from_numpy(species.astype(np.float32)).view(-1, 1)
    ref = orc.forward(sd, p, edges, box, feat=feat, bond=bonds).numpy()
    assert 25 * 4170 < edges.shape[1] < 40 * 4170
    assert rel_err(out, ref) < TOL
    eng.close()


@pytest.mark.parametrize("name", ["lj258_seed0", "tip3p774_seed3"])
def test_model_level_forward_with_explicit_edges(name):
    """pnet_model([pos], [edge_idx]) / ([pos], feat, [edge_idx]) with the REFERENCE's own edge list and
    order (nn_module.py:672-685, :545-558): no radius search, atoms not renumbered."""
    from gamd_amd.compat import ParticleNetLightningLJ, ParticleNetLightningWater
    g, cfg, sd = load_golden(name)
    box = float(g["box"])
    posw = torch.from_numpy(np.mod(g["pos"], box)).float().cuda()
    edge_idx = torch.from_numpy(g["edge_idx"]).long().cuda()
    if "node_feat" in g:
        m = ParticleNetLightningWater(state_dict=sd)
        out = m.pnet_model([posw], torch.from_numpy(g["node_feat"]).cuda(), [edge_idx])
    else:
        m = ParticleNetLightningLJ(state_dict=sd)
        out = m.pnet_model([posw], [edge_idx])
    assert rel_err(out.cpu().numpy(), g["out_norm"]) < TOL
    # a shuffled edge list gives the same forces (only the summation order inside a row changes)
    perm = torch.randperm(edge_idx.shape[1], generator=torch.Generator().manual_seed(0)).cuda()
    eng = m._get_engine()
    species = (g["node_feat"].reshape(-1) != 0) if "node_feat" in g else None
    out2 = eng.forward_edges(posw, edge_idx[:, perm], species=species).cpu().numpy()
    assert rel_err(out2, g["out_norm"]) < TOL
    # an empty edge list: every atom isolated -> decoder of the residual stream only
    out3 = eng.forward_edges(posw, torch.zeros((2, 0), dtype=torch.long), species=species).cpu().numpy()
    feat = torch.from_numpy(g["node_feat"]) if "node_feat" in g else None
    ref3 = orc.forward(sd, posw.cpu(), torch.zeros((2, 0), dtype=torch.long), box, feat=feat,
                       bond=g["bond"] if "bond" in g else None).numpy()
    assert rel_err(out3, ref3) < TOL
    with pytest.raises(Exception, match="outside"):
        eng.forward_edges(posw, torch.tensor([[0, 5], [1, 100000]]), species=species)


BF16_TOL = 1e-2      # BASELINE config 5: tolerance re-stated for the bf16 edge-MLP; achieved ~4e-3


def test_c5_bf16_edge_mlp_against_fp32_path_and_oracle():
    """BASELINE config 5: TIP4P-Ew-sized water box (2 000 molecules = 6 000 network atoms, M-sites dropped,
    train_utils.py:58-59), edge-MLP GEMMs on bf16 MFMA with fp32 accumulate.  Edge sets must be identical to the
    fp32 path (the neighbour search stays fp32); forces within the restated tolerance of the fp32 path and of the
    fp32 CPU oracle."""
    pos, box, species, bonds = workloads.water_box(2000, mol_per_20A3=251.0, seed=3456)
    cfg = ModelConfig(kind="water", use_bond=True)
    sd = make_state_dict(cfg, 3, 2.9, 1.1)
    p = torch.from_numpy(pos).float()
    e32 = _engine(sd, pos.shape[0], box, 4.2, bond=bonds, scaler=SHIPPED_SCALERS["tip4p"])
    e16 = _engine(sd, pos.shape[0], box, 4.2, bond=bonds, scaler=SHIPPED_SCALERS["tip4p"], edge_dtype="bf16")
    a = e32.forward(p, species=species).cpu().numpy().copy()
    b = e16.forward(p, species=species).cpu().numpy().copy()
    assert np.array_equal(edge_set(e16.debug_edges()), edge_set(e32.debug_edges()))
    err = rel_err(b, a)
    assert 1e-5 < err < BF16_TOL, err            # really is the reduced-precision path, and within tolerance
    edges = torch.from_numpy(e32.debug_edges()).long()
    feat = torch.from_numpy(species.astype(np.float32)).view(-1, 1)
    ref = orc.forward(sd, p, edges, box, feat=feat, bond=bonds).numpy()
    assert rel_err(b, ref) < BF16_TOL and rel_err(a, ref) < TOL
    b2 = e16.forward(p, species=species).cpu().numpy()
    assert np.array_equal(b, b2)                 # still bit-reproducible
    e32.close(); e16.close()


def test_bf16_lj_golden_within_restated_tolerance():
    g, cfg, sd = load_golden("lj258_seed0")
    box = float(g["box"])
    eng = _engine(sd, 258, box, float(g["cutoff"]), scaler=(g["scaler_mean"], g["scaler_var"]), edge_dtype="bf16")
    out = eng.forward(torch.from_numpy(np.mod(g["pos"], box)).float()).cpu().numpy()
    assert rel_err(out, g["out_norm"]) < BF16_TOL
    eng.close()


def test_tiny_and_crowded_systems():
    """Edge cases of the neighbour build: fewer atoms than one 32-row tile; a single over-full cell
    (> 64 atoms: serial sort fallback) with rows longer than a wave."""
    sd = make_state_dict(ModelConfig(kind="lj"), 11, 3.0, 1.0)
    # 3 atoms, far apart: only the self edges remain
    pos = np.array([[1.0, 1.0, 1.0], [10.0, 10.0, 10.0], [19.0, 3.0, 7.0]])
    eng = _engine(sd, 3, 25.0, 3.0)
    out = eng.forward(torch.from_numpy(pos).float()).cpu().numpy()
    edges = eng.debug_edges()
    assert edges.shape[1] == 3 and np.array_equal(edges[0], edges[1])
    ref = orc.forward(sd, torch.from_numpy(pos).float(), torch.from_numpy(edges).long(), 25.0).numpy()
    assert rel_err(out, ref) < TOL
    eng.close()
    # 200 atoms in a box smaller than 2 cutoffs: one cell, every pair within the cutoff is an edge (degree ~120)
    rng = np.random.default_rng(21)
    pos = rng.uniform(0, 9.0, (200, 3))
    eng = _engine(sd, 200, 9.0, 4.4)
    p = torch.from_numpy(pos).float()
    out = eng.forward(p).cpu().numpy()
    edges = eng.debug_edges()
    ref_e = orc.neighbor_edges(p, 9.0, 4.4, "jaxmd").numpy()
    assert np.array_equal(edge_set(edges), edge_set(ref_e))
    assert np.bincount(edges[0]).max() > 64
    ref = orc.forward(sd, p, torch.from_numpy(edges).long(), 9.0).numpy()
    assert rel_err(out, ref) < TOL
    eng.close()


def test_box_changes_between_calls():
    """WaterMDDynamicBoxNet takes the box per call (nn_module.py:391-396): the cell grid is rebuilt when it changes."""
    g, cfg, sd = load_golden("dynbox384_seed4")
    n = g["pos"].shape[0]
    eng = _engine(sd, n, g["box"], float(g["cutoff"]), nbr_flavour="torch", cfg=ModelConfig(kind="water", use_bond=False))
    species = g["node_feat"].reshape(-1) != 0
    feat = torch.from_numpy(g["node_feat"])
    for scale in (1.0, 1.6, 0.8, 1.0):
        box = (g["box"] * scale).astype(np.float32)
        pos = torch.from_numpy(g["pos"] * scale)
        out = eng.forward(pos, box=box, species=species).cpu().numpy()
        ref = orc.forward_dynamic_box(sd, pos, feat, box, float(g["cutoff"])).numpy()
        assert rel_err(out, ref) < TOL, scale
    eng.close()


def test_error_conventions():
    """Errors come back as negative status + message (GamdError), never as a crash."""
    from gamd_amd._lib import GamdError
    sdw = make_state_dict(ModelConfig(kind="water", use_bond=True), 1)
    with pytest.raises(ValueError, match="bond"):
        _engine(sdw, 30, 12.0, 3.0)                                  # use_bond model without a bond list
    eng = _engine(sdw, 30, 12.0, 3.0, bond=np.array([[0, 1], [0, 2]]))
    pos = torch.rand(30, 3) * 12
    with pytest.raises(GamdError, match="species"):
        eng.forward(pos)                                             # water model needs species
    with pytest.raises(ValueError, match=r"\[30, 3\]"):
        eng.forward(torch.rand(31, 3))
    with pytest.raises(GamdError, match="positive"):
        eng.forward(pos, box=[12.0, -1.0, 12.0], species=np.zeros(30))
    with pytest.raises(GamdError, match="4 bonded"):
        _engine(sdw, 30, 12.0, 3.0, bond=np.array([[0, k] for k in range(1, 7)]))
    eng.close()


def test_error_conventions_for_widths_and_constraints():
    """Unsupported widths and inconsistent integrator parameters are refused with a message, not run."""
    from gamd_amd._lib import GamdError
    # (hidden_dim up to 256 since round 6, fp32 edge MLP only: tests/test_gpu_hidden256.py)
    with pytest.raises(ValueError, match="hidden_dim"):
        cfg = ModelConfig(kind="water", hidden_dim=257, encoding_size=256, edge_embedding_dim=256, conv_layer=1)
        _engine(make_state_dict(cfg, 1), 30, 12.0, 3.0, cfg=cfg)
    with pytest.raises(GamdError, match="hidden_dim above 128"):
        cfg = ModelConfig(kind="water", hidden_dim=256, encoding_size=256, edge_embedding_dim=256, conv_layer=1)
        _engine(make_state_dict(cfg, 1), 30, 12.0, 3.0, cfg=cfg, edge_dtype="f16x3")
    wide = ModelConfig(kind="water", encoding_size=256, edge_embedding_dim=256, conv_layer=2)
    # (reduced-precision edge MLPs exist for every width since round 4; appended self loops are an fp32-only switch)
    with pytest.raises((GamdError, ValueError)):
        _engine(make_state_dict(ModelConfig(kind="lj"), 1), 30, 12.0, 3.0, edge_dtype="bf16", self_loop_mode="append_zero_feature_loops")
    # a 128-wide state_dict handed to a 256-wide configuration: strict shape check (load_state_dict semantics)
    with pytest.raises(KeyError, match="shape"):
        _engine(make_state_dict(ModelConfig(kind="water", conv_layer=2), 1), 30, 12.0, 3.0, cfg=wide)
    eng = _engine(make_state_dict(ModelConfig(kind="water"), 1), 31, 12.0, 3.0)
    x, v, f = (torch.rand(31, 3, device="cuda") for _ in range(3))
    sp = np.arange(31) % 3 == 0
    with pytest.raises(GamdError, match="multiple of 3"):
        eng.md_run(x, v, f, 1, mass_amu=16.0, mass_h_amu=1.0, species=sp, rigid_water=True, r_oh=0.96, r_hh=1.51)
    eng.close()
    eng = _engine(make_state_dict(ModelConfig(kind="water"), 1), 30, 12.0, 3.0)
    x, v, f = (torch.rand(30, 3, device="cuda") for _ in range(3))
    sp = np.arange(30) % 3 == 0
    with pytest.raises(GamdError, match="mass_h_amu"):
        eng.md_run(x, v, f, 1, mass_amu=16.0, species=sp, rigid_water=True, r_oh=0.96, r_hh=1.51)
    with pytest.raises(GamdError, match="r_hh"):
        eng.md_run(x, v, f, 1, mass_amu=16.0, mass_h_amu=1.0, species=sp, rigid_water=True, r_oh=0.96, r_hh=2.0)
    eng.close()


def test_verlet_skin_reuse_gives_the_exact_edge_set_every_step():
    """neighbor_skin > 0 (graph_utils.py:21-25,36-44 semantics: list built with cutoff + skin, rebuilt when an atom
    has moved skin/2, exact cutoff re-applied every call): along a random walk the edge SET and the forces equal
    those of the rebuild-every-call engine, while the candidate list is rebuilt only now and then."""
    rng = np.random.default_rng(21)
    n, rc, skin = 1500, 7.5, 1.2
    pos, box = workloads.lj_box(n, seed=4)
    sd = make_state_dict(ModelConfig(kind="lj"), 2, 5.0, 1.7)
    exact = _engine(sd, n, box, rc)
    reuse = _engine(sd, n, box, rc, neighbor_skin=skin)
    x = pos.copy()
    rebuilds = []
    for step in range(40):
        if step == 25:
            x[7] += np.array([3.0, -2.0, 0.5])                       # one atom jumps: immediate rebuild
        p = torch.from_numpy(x).float()
        f0 = exact.forward(p).cpu().numpy()
        f1 = reuse.forward(p).cpu().numpy()
        assert np.array_equal(edge_set(exact.debug_edges()), edge_set(reuse.debug_edges())), step
        assert rel_err(f1, f0) < TOL, step
        rebuilds.append(reuse.skin_stats()[0])
        x = x + rng.normal(0.0, 0.06, x.shape)
    assert rebuilds[0] == 1 and rebuilds[-1] < 12                    # reused on most steps
    assert rebuilds[25] == rebuilds[24] + 1                          # the jump forced one
    assert rebuilds[-1] >= 3                                         # and the walk did too
    n_cand = reuse.skin_stats()[1]
    # another box on the next call: candidates are rebuilt for it
    b2 = box * 1.02
    p = torch.from_numpy(x * 1.02).float()
    f0, f1 = exact.forward(p, box=b2).cpu().numpy(), reuse.forward(p, box=b2).cpu().numpy()
    assert reuse.skin_stats()[0] == rebuilds[-1] + 1
    assert np.array_equal(edge_set(exact.debug_edges()), edge_set(reuse.debug_edges())) and rel_err(f1, f0) < TOL
    assert n_cand > exact.counts()[0]
    exact.close(); reuse.close()


def test_verlet_skin_in_the_md_loop_and_candidate_regrow():
    from gamd_amd._lib import GamdError
    n, rc = 2000, 7.5
    pos, box = workloads.lj_box(n, seed=8)
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2)
    eng = _engine(sd, n, box, rc, scaler=SHIPPED_SCALERS["lj"], neighbor_skin=rc / 6)
    x = torch.from_numpy(pos).float().cuda()
    v = torch.from_numpy(workloads.maxwell_boltzmann(n, temperature_k=300.0, seed=1)).float().cuda()
    f = eng.forward(x, denormalize=True).clone()
    eng.md_run(x, v, f, 150, temperature_k=300.0)
    r = eng.skin_stats()[0]
    assert 1 <= r < 40                                               # 150 steps, a handful of rebuilds
    exact = _engine(sd, n, box, rc, scaler=SHIPPED_SCALERS["lj"])
    f0 = exact.forward(x, denormalize=True).cpu().numpy()
    f1 = eng.forward(x, denormalize=True).cpu().numpy()
    assert np.array_equal(edge_set(exact.debug_edges()), edge_set(eng.debug_edges()))
    assert rel_err(f1, f0) < TOL
    exact.close(); eng.close()
    # too small a capacity: candidate / edge buffers are regrown and the call retried (status 1)
    small = _engine(sd, n, box, rc, neighbor_skin=rc / 6, edge_capacity=1000)
    out = small.forward(torch.from_numpy(pos).float()).cpu().numpy()
    assert small.last_status == 1
    ref = _engine(sd, n, box, rc)
    assert rel_err(out, ref.forward(torch.from_numpy(pos).float()).cpu().numpy()) < TOL
    small.close(); ref.close()


def test_split_fp16_at_c2_size_against_the_fp32_path():
    n, rc = 10000, 10.2
    pos, box = workloads.lj_box(n, seed=1234)
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2)
    p = torch.from_numpy(pos).float()
    a = _engine(sd, n, box, rc)
    f32 = a.forward(p).cpu().numpy()
    a.close()
    b = _engine(sd, n, box, rc, edge_dtype="f16x3")
    f16 = b.forward(p).cpu().numpy()
    again = b.forward(p).cpu().numpy()
    b.close()
    assert np.array_equal(f16, again)                                  # bit-reproducible
    assert rel_err(f16, f32) < TOL


# small_tile_limit: throughput kernel only (never small) vs latency kernel only
MAIN_ONLY, SMALL_ONLY = dict(small_tile_limit=-1), dict(small_tile_limit=1000000)


def test_small_system_conv_kernel_is_bit_identical_to_the_throughput_kernel():
    """Below ~1 tile per SIMD the conv layer runs on conv_edge_small.hip (one tile shared by four waves).  Same
    floating-point operation order per output element as conv_edge.hip: the outputs must be equal bit for bit."""
    outs = {}
    for name in ["lj258_seed0", "tip3p774_seed3"]:
        g, cfg, sd = load_golden(name)
        box, rc, n = float(g["box"]), float(g["cutoff"]), g["pos"].shape[0]
        bond = g["bond"] if "bond" in g else None
        posw = torch.from_numpy(np.mod(g["pos"], box).astype(np.float32))
        species = (g["node_feat"].reshape(-1) != 0) if "node_feat" in g else None
        for limit, kw in (("0", MAIN_ONLY), ("1000000", SMALL_ONLY)):
            eng = _engine(sd, n, box, rc, bond=bond, **kw)
            outs[limit] = eng.forward(posw, species=species).cpu().numpy().copy()
            eng.close()
        assert np.array_equal(outs["0"], outs["1000000"]), name
        assert rel_err(outs["0"], g["out_norm"]) < TOL
    # the generic-width variant against wide.hip's conv kernel (DFT-water widths and a mixed one)
    for name in ["dynbox384_dftcfg_seed5", "dynbox384_h128_e256_noexpand_seed8"]:
        g, cfg, sd = load_golden(name)
        n = g["pos"].shape[0]
        species = g["node_feat"].reshape(-1) != 0
        for limit, kw in (("0", MAIN_ONLY), ("1000000", SMALL_ONLY)):
            eng = _engine(sd, n, g["box"], float(g["cutoff"]), nbr_flavour="torch", cfg=cfg, **kw)
            outs[limit] = eng.forward(torch.from_numpy(g["pos"]), box=g["box"], species=species).cpu().numpy().copy()
            eng.close()
        assert np.array_equal(outs["0"], outs["1000000"]), name
        assert rel_err(outs["0"], g["out_norm"]) < TOL
