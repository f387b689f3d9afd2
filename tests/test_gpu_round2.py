"""GPU tests added in round 2: checkpoint-driven driver sequences, the self_loop_mode switch,
float node features, freeze-and-resume of an enqueued MD run after a neighbour-buffer overflow,
non-finite / operand-range flags, argument checks of the MD entry points, fresh output tensors, the two timing buckets
of predict_forces(verbose=True), and the 8 001-network-atom reading of BASELINE config 5.  All through the C ABI."""
from types import SimpleNamespace

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


# ---------------------------------------------------------------------------------------------------------------
# section 8 f-2: the literal driver sequences, from files on disk
# ---------------------------------------------------------------------------------------------------------------
ARGS = SimpleNamespace(use_layer_norm=True, encoding_size=128, hidden_dim=128, edge_embedding_dim=128, drop_edge=False,
                       conv_layer=4, rotate_aug=False, update_edge=False, use_part=False, data_dir='', loss='mae')


def _write_ckpt(tmp_path, sd, mean, var, lightning=True):
    path, scaler = tmp_path / "checkpoint.ckpt", tmp_path / "scaler.npz"
    obj = {"state_dict": {"pnet_model." + k: v for k, v in sd.items()}, "epoch": 29, "global_step": 1} if lightning else dict(sd)
    torch.save(obj, path)
    np.savez(scaler, mean=np.asarray(mean), var=np.asarray(var))
    return str(path), str(scaler)


@pytest.mark.parametrize("lightning", [True, False])
def test_lj_driver_sequence_from_checkpoint_files(tmp_path, lightning):
    """LJ/test_script/test_langevin.py:61-77,108: ParticleNetLightning(args).load_from_checkpoint(PATH, args=args);
    load_training_stats(SCALER_CKPT); .cuda(); .eval(); predict_forces(pos).  Lightning-shaped checkpoint
    ('state_dict' with the pnet_model. prefix, train_network_lj.py:95) and the plain state_dict of --state_ckpt_dir
    (:85-87)."""
    from gamd_amd.compat import ParticleNetLightningLJ
    g, cfg, sd = load_golden("lj258_pert_seed1")
    PATH, SCALER_CKPT = _write_ckpt(tmp_path, sd, g["scaler_mean"], g["scaler_var"], lightning)
    model = ParticleNetLightningLJ(ARGS).load_from_checkpoint(PATH, args=ARGS)
    model.load_training_stats(SCALER_CKPT)
    model.cuda()
    model.eval()
    force = model.predict_forces(g["pos"])
    assert isinstance(force, np.ndarray) and force.dtype == np.float64 and force.shape == (258, 3)
    assert rel_err(force, g["forces"]) < TOL
    # the scaler may also arrive after the engine exists (a second load_training_stats re-targets it)
    np.savez(tmp_path / "scaler2.npz", mean=np.array([1.5]), var=np.array([4.0]))
    model.load_training_stats(str(tmp_path / "scaler2.npz"))
    assert rel_err(model.predict_forces(g["pos"]), g["out_norm"].astype(np.float64) * 2.0 + 1.5) < TOL


def test_water_driver_sequence_from_checkpoint_files(tmp_path):
    """water/test_script/test_nosehoover.py:79-89,123: the same with feat (O = 1, H = 0 float [N,1] on the device)."""
    from gamd_amd.compat import ParticleNetLightningWater
    g, cfg, sd = load_golden("tip3p774_seed3")
    PATH, SCALER_CKPT = _write_ckpt(tmp_path, sd, g["scaler_mean"], g["scaler_var"])
    model = ParticleNetLightningWater(ARGS).load_from_checkpoint(PATH, args=ARGS)
    model.load_training_stats(SCALER_CKPT)
    model.cuda()
    model.eval()
    feat = torch.zeros((774, 1), dtype=torch.float32)
    feat[::3] = 1.0                                              # test_nosehoover.py:82-89
    force = model.predict_forces(feat.cuda(), g["pos"])
    assert force.dtype == np.float64 and rel_err(force, g["forces"]) < TOL


def test_dft_driver_sequence_from_checkpoint_files(tmp_path):
    """water/test_script/test_nosehoover_hb.py:64-113,130: predict_forces(feat, pos, box) in bohr with the shipped
    widths; the driver multiplies by 2625.5 / 0.0529177 afterwards."""
    from gamd_amd.compat import ParticleNetLightningDFT, HARTREE_PER_BOHR_TO_KJ_PER_MOL_NM
    g, cfg, sd = load_golden("dynbox384_dftcfg_seed5")
    mean, var = SHIPPED_SCALERS["dft"]
    PATH, SCALER_CKPT = _write_ckpt(tmp_path, sd, mean, var)
    args = SimpleNamespace(cutoff=float(g["cutoff"]), encoding_size=256, hidden_dim=128, edge_embedding_dim=256, conv_layer=5)
    model = ParticleNetLightningDFT(args, num_atoms=g["pos"].shape[0]).load_from_checkpoint(PATH, args=args)
    model.load_training_stats(SCALER_CKPT)
    model.cuda()
    model.eval()
    force = model.predict_forces(torch.from_numpy(g["node_feat"]).cuda(), g["pos"].astype(np.float64), g["box"])
    ref = g["out_norm"].astype(np.float64) * np.sqrt(var) + mean
    assert rel_err(force, ref) < TOL
    assert rel_err(force * HARTREE_PER_BOHR_TO_KJ_PER_MOL_NM, ref * (2625.5 / 0.0529177)) < TOL


def test_checkpoint_with_wrong_shapes_is_rejected(tmp_path):
    from gamd_amd.compat import ParticleNetLightningLJ
    sd = make_state_dict(ModelConfig(kind="lj"), 0)
    sd["graph_decoder.mlp_layer.2.weight"] = torch.zeros(4, 128)
    PATH, _ = _write_ckpt(tmp_path, sd, [0.0], [1.0])
    model = ParticleNetLightningLJ(ARGS).load_from_checkpoint(PATH, args=ARGS)
    with pytest.raises(KeyError, match="graph_decoder"):
        model.predict_forces(np.zeros((258, 3)))


# ---------------------------------------------------------------------------------------------------------------
# self_loop_mode
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("skin", [0.0, 1.25])
def test_self_loop_mode_append_lj_against_reference_and_oracle(skin):
    """Mode 1 against the reference module run with an in-place add_self_loop (tests/golden/*selfloop_inplace*), stage
    by stage, and against the oracle; exact build and Verlet-skin filter both append the loops."""
    g, cfg, sd = load_golden("lj258_selfloop_inplace_seed0")
    box, rc, n = float(g["box"]), float(g["cutoff"]), 258
    posw = torch.from_numpy(np.mod(g["pos"], box).astype(np.float32))
    eng = _engine(sd, n, box, rc, keep_stages=True, neighbor_skin=skin, self_loop_mode="append_zero_feature_loops")
    out = eng.forward(posw).cpu().numpy()
    assert eng.counts()[0] == g["edge_idx"].shape[1] + n               # one appended loop per atom
    edges = eng.debug_edges()
    loops = np.stack([np.arange(n), np.arange(n)])
    want = np.concatenate([g["edge_idx"].astype(np.int64), loops], axis=1)
    key = lambda e: np.sort(e[0].astype(np.int64) * n + e[1])
    assert np.array_equal(key(edges), key(want))                       # multiset: (i, i) now appears twice
    row_ptr, col = eng.debug_csr()
    perm_inv = np.arange(n)                                            # the appended loop is the LAST edge of every row
    assert np.array_equal(col[row_ptr[1:] - 1], perm_inv)
    e = eng.debug_e()
    assert np.all(e[row_ptr[1:] - 1] == 0.0) and np.all(np.abs(e[row_ptr[:-1]]).sum(1) > 0)
    for l in range(g["h_layers"].shape[0]):
        assert rel_err(eng.debug_h(l), g["h_layers"][l]) < TOL, f"h_{l}"
    assert rel_err(out, g["out_norm"]) < TOL
    ref = orc.forward(sd, posw, torch.from_numpy(g["edge_idx"]).long(), box, self_loop_mode="append_zero_feature_loops").numpy()
    assert rel_err(out, ref) < TOL
    # a second call (skin: candidates reused) gives the same result
    assert rel_err(eng.forward(posw).cpu().numpy(), g["out_norm"]) < TOL
    eng.close()
    # default mode on the same inputs = the no-op reading = the other golden
    g0 = load_golden("lj258_seed0")[0]
    eng0 = _engine(sd, n, box, rc)
    assert rel_err(eng0.forward(posw).cpu().numpy(), g0["out_norm"]) < TOL
    eng0.close()


@pytest.mark.parametrize("kernel_select", [0, 1])
def test_self_loop_mode_append_dynamic_box(kernel_select):
    """md_module.get_neighbor flavour (no self pairs in the search) + appended loops; also on the generic-width kernels."""
    g, cfg, sd = load_golden("dynbox384_selfloop_inplace_seed4")
    n = g["pos"].shape[0]
    eng = _engine(sd, n, g["box"], float(g["cutoff"]), nbr_flavour="torch", cfg=cfg, kernel_select=kernel_select,
                  self_loop_mode="append_zero_feature_loops")
    out = eng.forward(torch.from_numpy(g["pos"]), box=g["box"], species=g["node_feat"].reshape(-1)).cpu().numpy()
    assert eng.counts()[0] == g["edge_idx"].shape[1] + n
    assert rel_err(out, g["out_norm"]) < TOL
    eng.close()


def test_self_loop_mode_append_model_level_call_and_rejections():
    """pnet_model([pos], [edge_idx]) with the reference's edge list: build_graph appends the loops there too."""
    from gamd_amd.compat import ParticleNetLightningLJ
    from gamd_amd._lib import GamdError
    g, cfg, sd = load_golden("lj258_selfloop_inplace_seed0")
    box = float(g["box"])
    posw = torch.from_numpy(np.mod(g["pos"], box)).float().cuda()
    m = ParticleNetLightningLJ(state_dict=sd, self_loop_mode="append_zero_feature_loops")
    out = m.pnet_model([posw], [torch.from_numpy(g["edge_idx"]).long().cuda()])
    assert rel_err(out.cpu().numpy(), g["out_norm"]) < TOL
    with pytest.raises(ValueError, match="self_loop_mode"):
        _engine(sd, 258, box, 7.5, self_loop_mode="bogus")
    with pytest.raises(GamdError, match="fp32 edge dtype"):
        _engine(sd, 258, box, 7.5, self_loop_mode="append_zero_feature_loops", edge_dtype="f16x3")


# ---------------------------------------------------------------------------------------------------------------
# float node features (nn_module.py:554: x = node_encoder(feat), any float)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kernel_select", [0, 1])
def test_float_node_features_are_carried_not_binarised(kernel_select):
    g, cfg, sd = load_golden("tip3p774_seed3")
    box, rc, n = float(g["box"]), float(g["cutoff"]), 774
    posw = torch.from_numpy(np.mod(g["pos"], box)).float()
    feat = torch.from_numpy(np.random.default_rng(3).normal(0.3, 0.8, (n, 1)).astype(np.float32))
    eng = _engine(sd, n, box, rc, bond=g["bond"], kernel_select=kernel_select)
    out = eng.forward(posw, species=feat).cpu().numpy()
    edges = torch.from_numpy(eng.debug_edges()).long()
    ref = orc.forward(sd, posw, edges, box, feat=feat, bond=g["bond"]).numpy()
    assert rel_err(out, ref) < TOL
    binar = orc.forward(sd, posw, edges, box, feat=(feat != 0).float(), bond=g["bond"]).numpy()
    assert rel_err(out, binar) > 1e-2                                  # it really is a different input
    # integer / bool species keep meaning O = 1 / H = 0, and switching back drops the float features
    out01 = eng.forward(posw, species=(g["node_feat"].reshape(-1) != 0)).cpu().numpy()
    assert rel_err(out01, g["out_norm"]) < TOL
    eng.close()
    from gamd_amd.compat import ParticleNetLightningWater
    w = ParticleNetLightningWater(state_dict=sd)
    fw = w.pnet_model([posw.cuda()], feat.cuda(), [edges.cuda()]).cpu().numpy()
    assert rel_err(fw, ref) < TOL


# ---------------------------------------------------------------------------------------------------------------
# overflow inside an enqueued MD run: freeze on the device, regrow, resume
# ---------------------------------------------------------------------------------------------------------------
def _md_state(n, seed):
    pos, box = workloads.lj_box(n, seed=seed)
    x = torch.from_numpy(pos).float().cuda()
    v = torch.from_numpy(workloads.maxwell_boltzmann(n, seed=5)).float().cuda()
    return x, v, box


@pytest.mark.parametrize("integrator", ["baoab", "nhc"])
def test_md_run_overflow_freezes_state_and_resumes_bit_exactly(integrator):
    """edge_capacity far too small: the overflow is detected by the first force evaluation INSIDE the enqueued run.  The
    integrator kernels must not advance x, v with stale forces; gamd_sync_status regrows, re-evaluates the forces at the
    frozen positions and finishes the run.  Exact rebuild every step -> the result equals the ample-buffer run bit for bit."""
    n, rc, steps = 1500, 7.5, 12
    sd = make_state_dict(ModelConfig(kind="lj"), 2, 5.0, 1.7)
    res = {}
    for tag, cap in (("ample", 0), ("tiny", 4000)):
        x, v, box = _md_state(n, 4)
        eng = _engine(sd, n, box, rc, scaler=SHIPPED_SCALERS["lj"], edge_capacity=0)
        f = eng.forward(x, denormalize=True)
        eng.close()
        eng = _engine(sd, n, box, rc, scaler=SHIPPED_SCALERS["lj"], edge_capacity=cap)
        if integrator == "baoab":
            eng.md_run(x, v, f, steps, seed=11, sync=False)
        else:
            chain = eng.md_run_nhc(x, v, f, steps, sync=False)
        st = eng.sync_status()
        assert st == (1 if cap else 0)
        assert eng.counts()[2] > (cap if cap else 0)
        res[tag] = (x.cpu().numpy(), v.cpu().numpy(), f.cpu().numpy(), chain.cpu().numpy() if integrator == "nhc" else None)
        # the run is complete: one more step works without further regrowth
        eng.md_run(x, v, f, 1, seed=11, first_step=steps) if integrator == "baoab" else eng.md_run_nhc(x, v, f, 1, chain_state=chain)
        assert eng.last_status == 0
        eng.close()
    for a, b in zip(res["ample"], res["tiny"]):
        if a is not None:
            assert np.isfinite(a).all() and np.array_equal(a, b)
    assert not np.array_equal(res["ample"][0], _md_state(n, 4)[0].cpu().numpy())      # it did move


def test_md_run_candidate_overflow_with_skin_reuse_freezes_and_resumes():
    """Verlet-skin mode: the CANDIDATE list overflows (it is larger than the edge list).  Before round 2 that overflow
    never reached the node kernels' check and forces were computed from a truncated list.  The order of a row's edges
    depends on when candidates were rebuilt, so compare to the ample run at fp32 rounding, not bit for bit."""
    n, rc, steps = 1500, 7.5, 10
    sd = make_state_dict(ModelConfig(kind="lj"), 2, 5.0, 1.7)
    res = {}
    for tag, cap in (("ample", 0), ("tiny", 3000)):
        x, v, box = _md_state(n, 4)
        eng = _engine(sd, n, box, rc, scaler=SHIPPED_SCALERS["lj"], edge_capacity=cap, neighbor_skin=rc / 6)
        if cap:
            assert eng.skin_stats()[2] < 60000                          # candidate capacity follows the edge capacity
        e0 = _engine(sd, n, box, rc, scaler=SHIPPED_SCALERS["lj"])
        f = e0.forward(x, denormalize=True)
        e0.close()
        eng.md_run(x, v, f, steps, seed=3, sync=False)
        assert eng.sync_status() == (1 if cap else 0)
        res[tag] = (x.cpu().numpy(), v.cpu().numpy(), f.cpu().numpy())
        eng.close()
    for a, b in zip(res["ample"], res["tiny"]):
        assert np.isfinite(b).all() and rel_err(b, a) < 1e-4


def test_forward_after_overflow_in_md_run_is_clean():
    """The freeze flag is cleared by the regrow: a plain force call afterwards equals a fresh engine's."""
    n, rc = 1200, 7.5
    sd = make_state_dict(ModelConfig(kind="lj"), 2, 5.0, 1.7)
    x, v, box = _md_state(n, 9)
    ref = _engine(sd, n, box, rc)
    eng = _engine(sd, n, box, rc, edge_capacity=2000)
    f = ref.forward(x, denormalize=True)
    eng.md_run(x, v, f, 3, seed=1)
    assert eng.last_status == 1
    assert np.array_equal(eng.forward(x).cpu().numpy(), ref.forward(x).cpu().numpy())
    ref.close(); eng.close()


# ---------------------------------------------------------------------------------------------------------------
# small systems (n <= 1024) in skin mode: 3 neighbour launches per call, integrator halves fused into the first one
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,flavour", [(258, "jaxmd"), (700, "jaxmd"), (1024, "jaxmd"), (384, "torch")])
def test_small_system_skin_path_gives_the_exact_edge_set_every_step(n, flavour):
    """Same contract as the large-system skin path (graph_utils.py:21-25,36-44), on the single-workgroup kernels: along a
    random walk the edge SET equals the rebuild-every-call engine's at every step, forces agree to rounding, candidates
    are rebuilt only now and then, and a jump forces an immediate rebuild."""
    rng = np.random.default_rng(5)
    rc, skin = 7.5, 1.25
    pos, box = workloads.lj_box(n, seed=8)
    sd = make_state_dict(ModelConfig(kind="lj"), 2, 5.0, 1.7)
    exact = _engine(sd, n, box, rc, nbr_flavour=flavour)
    reuse = _engine(sd, n, box, rc, neighbor_skin=skin, nbr_flavour=flavour)
    x = pos.copy()
    rebuilds = []
    for step in range(36):
        if step == 20:
            x[7] += np.array([3.0, -2.0, 0.5])
        p = torch.from_numpy(x).float()
        f0, f1 = exact.forward(p).cpu().numpy(), reuse.forward(p).cpu().numpy()
        assert exact.counts()[0] == reuse.counts()[0], step
        assert np.array_equal(edge_set(exact.debug_edges()), edge_set(reuse.debug_edges())), step
        assert rel_err(f1, f0) < TOL, step
        rebuilds.append(reuse.skin_stats()[0])
        x = x + rng.normal(0.0, 0.06, x.shape)
    assert rebuilds[0] == 1 and rebuilds[20] == rebuilds[19] + 1 and 3 <= rebuilds[-1] < 12
    exact.close(); reuse.close()


@pytest.mark.parametrize("n", [300, 1500])
def test_fused_md_steps_match_the_unfused_path(n):
    """Skin mode: B of step s-1 and B A O A of step s run inside the first neighbour kernel of step s (k_step_small up to
    1024 atoms, k_skin_check above).  Same noise stream (seed, step, atom) and the same arithmetic as the stand-alone
    integrator kernels: against an engine without skin (un-fused launches, exact rebuild) the trajectory agrees to the
    rounding of the row order, and calling md_run step by step equals one call."""
    rc = 7.5
    sd = make_state_dict(ModelConfig(kind="lj"), 2, 5.0, 1.7)
    out = {}
    for tag, kw, chunks in (("unfused", {}, [12]), ("fused", dict(neighbor_skin=rc / 6), [12]),
                            ("fused_split", dict(neighbor_skin=rc / 6), [1, 4, 7])):
        x, v, box = _md_state(n, 4)
        eng = _engine(sd, n, box, rc, scaler=SHIPPED_SCALERS["lj"], **kw)
        f = eng.forward(x, denormalize=True)
        done = 0
        for c in chunks:
            eng.md_run(x, v, f, c, seed=11, first_step=done)
            done += c
        out[tag] = (x.cpu().numpy(), v.cpu().numpy(), f.cpu().numpy())
        eng.close()
    for a, b in zip(out["fused"], out["fused_split"]):
        assert np.array_equal(a, b)
    for a, b in zip(out["unfused"], out["fused"]):
        assert np.isfinite(b).all() and rel_err(b, a) < 2e-5
    assert not np.array_equal(out["fused"][0], _md_state(n, 4)[0].cpu().numpy())


def test_small_system_fused_md_overflow_freezes_and_resumes():
    n, rc, steps = 300, 7.5, 9
    sd = make_state_dict(ModelConfig(kind="lj"), 2, 5.0, 1.7)
    res = {}
    for tag, cap in (("ample", 0), ("tiny", 1500)):
        x, v, box = _md_state(n, 4)
        e0 = _engine(sd, n, box, rc, scaler=SHIPPED_SCALERS["lj"])
        f = e0.forward(x, denormalize=True)
        e0.close()
        eng = _engine(sd, n, box, rc, scaler=SHIPPED_SCALERS["lj"], edge_capacity=cap, neighbor_skin=rc / 6)
        eng.md_run(x, v, f, steps, seed=3, sync=False)
        assert eng.sync_status() == (1 if cap else 0)
        res[tag] = (x.cpu().numpy(), v.cpu().numpy(), f.cpu().numpy())
        eng.md_run(x, v, f, 2, seed=3, first_step=steps)
        assert eng.last_status == 0 and torch.isfinite(x).all()
        eng.close()
    for a, b in zip(res["ample"], res["tiny"]):
        assert np.isfinite(b).all() and rel_err(b, a) < 1e-4


@pytest.mark.parametrize("n", [1025, 16384, 16385])
def test_sizes_at_the_kernel_selection_boundaries(n):
    """1024 / 1025 atoms: single-workgroup fused path vs the launch sequence with fixed-width candidate rows; 16 384 / 16 385:
    one-pass vs multi-pass row scan and one-workgroup vs four-kernel cell build.  Skin engine against the rebuild-every-call
    engine along a short walk with one forced rebuild: same edge set, forces to rounding, and the MD entry point runs."""
    rng = np.random.default_rng(n)
    rc = 6.0
    pos, box = workloads.lj_box(n, seed=5)
    sd = make_state_dict(ModelConfig(kind="lj"), 2, 5.0, 1.7)
    exact = _engine(sd, n, box, rc)
    reuse = _engine(sd, n, box, rc, neighbor_skin=1.0, scaler=SHIPPED_SCALERS["lj"])
    x = pos.copy()
    for step in range(6):
        if step == 3:
            x[n - 1] += np.array([2.0, 1.0, -2.5])
        p = torch.from_numpy(x).float()
        f0, f1 = exact.forward(p).cpu().numpy(), reuse.forward(p).cpu().numpy()
        assert exact.counts()[0] == reuse.counts()[0], step
        assert np.array_equal(edge_set(exact.debug_edges()), edge_set(reuse.debug_edges())), step
        assert rel_err(f1, f0) < TOL, step
        x = x + rng.normal(0.0, 0.05, x.shape)
    assert reuse.skin_stats()[0] >= 2
    xd = torch.from_numpy(x).float().cuda()
    vd = torch.zeros_like(xd)
    fd = reuse.forward(xd, denormalize=True).clone()
    reuse.md_run(xd, vd, fd, 5, temperature_k=100.0, seed=2)
    assert reuse.last_status == 0 and torch.isfinite(xd).all() and torch.isfinite(fd).all()
    exact.close(); reuse.close()


def test_more_than_16384_atoms_take_the_multi_pass_row_scan():
    """Up to 16 384 atoms the row scan (row_ptr, piece numbering) is one pass over registers; above that the generic
    multi-pass scan.  20 000 atoms: the edge set against an independent periodic KD-tree, the CSR invariants, and the forces
    against the CPU oracle on the same edges (the piece metadata decides which partial sums belong to which atom)."""
    from scipy.spatial import cKDTree
    n, rc = 20000, 6.0
    pos, box = workloads.lj_box(n, seed=77)
    sd = make_state_dict(ModelConfig(kind="lj"), 1, 5.0, 1.7)
    p = torch.from_numpy(pos).float()
    for kw in ({}, dict(neighbor_skin=rc / 6)):
        eng = _engine(sd, n, box, rc, **kw)
        out = eng.forward(p).cpu().numpy()
        edges = eng.debug_edges()
        row_ptr, col = eng.debug_csr()
        assert row_ptr[0] == 0 and row_ptr[-1] == edges.shape[1] == eng.counts()[0] and np.all(np.diff(row_ptr) > 0)
        w = np.mod(p.numpy().astype(np.float64), box)
        w[w >= box] = 0.0
        tree = cKDTree(w, boxsize=box)
        loops = np.tile(np.arange(n), (2, 1))

        def directed(r):
            pr = tree.query_pairs(r, output_type="ndarray").T
            return set(map(int, edge_set(np.concatenate([pr, pr[::-1], loops], axis=1))))
        # fp32 positions on the GPU, fp64 here: pairs within 1e-3 of the cutoff may fall on either side
        got = set(map(int, edge_set(edges)))
        inner, outer = directed(rc - 1e-3), directed(rc + 1e-3)
        assert inner <= got <= outer and len(outer) - len(inner) < 2000
        assert rel_err(out, orc.forward(sd, p, torch.from_numpy(edges).long(), box).numpy()) < TOL
        eng.close()


# ---------------------------------------------------------------------------------------------------------------
# argument checks of the MD entry points (they used to skip what gamd_forces_async checks)
# ---------------------------------------------------------------------------------------------------------------
def test_md_entry_points_check_species_bonds_and_rigid_layout():
    from gamd_amd._lib import GamdError
    pos, box, species, bonds = workloads.water_box(64, jitter=0.0, wrap=False)
    sd = make_state_dict(ModelConfig(kind="water", use_bond=True), 3, 2.9, 1.1)
    n = pos.shape[0]
    eng = _engine(sd, n, box, 4.2, bond=bonds, scaler=SHIPPED_SCALERS["tip3p"])
    x = torch.from_numpy(pos).float().cuda()
    v = torch.zeros_like(x)
    f = eng.forward(x, species=species, denormalize=True)
    kw = dict(mass_amu=workloads.MASS_O, mass_h_amu=workloads.MASS_H, rigid_water=True, r_oh=workloads.TIP3P_R_OH,
              r_hh=workloads.TIP3P_R_HH, dt_ps=0.0005)
    with pytest.raises(GamdError, match="species"):
        eng.md_run(x, v, f, 1, species=None, **kw)                     # water model without species
    with pytest.raises(GamdError, match="species"):
        eng.md_run_nhc(x, v, f, 1, species=None, **kw)
    bad = species.copy()
    bad[4] = 1                                                          # O,H,H layout broken
    with pytest.raises(GamdError, match="O,H,H"):
        eng.md_run(x, v, f, 1, species=bad, **kw)
    x0 = x.clone()
    eng.md_run(x, v, f, 2, species=species, **kw)                       # the valid call still runs
    assert torch.isfinite(x).all() and not torch.equal(x, x0)
    eng.close()
    eng2 = _engine(sd, n, box, 4.2, bond=bonds)
    eng2._lib.gamd_set_bonds(eng2._h, None, 0)                          # use_bond model with the bond table dropped
    with pytest.raises(GamdError, match="bonds"):
        eng2.md_run(x, v, f, 1, species=species, **kw)
    eng2.close()


def test_nhc_default_degrees_of_freedom_follow_the_reference(monkeypatch):
    """hack_integrator.py:226-235: 3 per particle - constraints - 3 with a CMMotionRemover (WaterBox has one)."""
    import gamd_amd.engine as E
    seen = {}

    class Recorded(E.GamdNhcParams):                      # the parameter block handed to gamd_md_run_nhc
        def __init__(self, *a):
            super().__init__(*a)
            seen["ndf"] = self.ndf

    monkeypatch.setattr(E, "GamdNhcParams", Recorded)
    pos, box, species, bonds = workloads.water_box(64, jitter=0.0, wrap=False)
    sd = make_state_dict(ModelConfig(kind="water", use_bond=True), 3, 2.9, 1.1)
    eng = _engine(sd, 192, box, 4.2, bond=bonds, scaler=SHIPPED_SCALERS["tip3p"])
    x = torch.from_numpy(pos).float().cuda()
    v = torch.zeros_like(x)
    f = eng.forward(x, species=species, denormalize=True)
    kw = dict(mass_amu=workloads.MASS_O, mass_h_amu=workloads.MASS_H, r_oh=workloads.TIP3P_R_OH, r_hh=workloads.TIP3P_R_HH,
              dt_ps=0.0005, species=species)
    eng.md_run_nhc(x, v, f, 1, rigid_water=True, **kw)
    assert seen["ndf"] == 6 * 64 - 3
    eng.md_run_nhc(x, v, f, 1, rigid_water=True, remove_cm_motion=False, **kw)
    assert seen["ndf"] == 6 * 64
    eng.md_run_nhc(x, v, f, 1, rigid_water=False, **kw)
    assert seen["ndf"] == 3 * 192
    eng.md_run_nhc(x, v, f, 1, rigid_water=False, ndf=17.0, **kw)
    assert seen["ndf"] == 17.0
    eng.close()


# ---------------------------------------------------------------------------------------------------------------
# non-finite forces / operand range of the split-fp16 mode
# ---------------------------------------------------------------------------------------------------------------
def test_nonfinite_flag_and_fp16_operand_range():
    from gamd_amd._lib import GamdError
    g, cfg, sd = load_golden("lj258_seed0")
    box = float(g["box"])
    posw = torch.from_numpy(np.mod(g["pos"], box)).float()
    eng = _engine(sd, 258, box, 7.5)
    eng.forward(posw)
    assert eng.nonfinite_seen() is False
    eng.close()
    # fp32: non-finite output is returned silently like the reference does, but the flag tells (and clears)
    bad = dict(sd)
    bad["graph_decoder.mlp_layer.2.bias"] = torch.tensor([0.0, float("nan"), 0.0])
    eng = _engine(bad, 258, box, 7.5)
    out = eng.forward(posw).cpu().numpy()
    assert np.isnan(out[:, 1]).all() and np.isfinite(out[:, 0]).all()
    assert eng.nonfinite_seen() is True and eng.nonfinite_seen() is False
    eng.close()
    # f16x3: activations beyond the fp16 range (|x| > 65504) cannot be split into hi + lo -> reported, not silent
    big = dict(sd)
    big["graph_conv.conv.0.edge_affine.mlp_layer.0.weight"] = sd["graph_conv.conv.0.edge_affine.mlp_layer.0.weight"] * 3.0e5
    ok32 = _engine(big, 258, box, 7.5)
    assert torch.isfinite(ok32.forward(posw)).all()                     # the fp32 path handles the same weights
    ok32.close()
    eng = _engine(big, 258, box, 7.5, edge_dtype="f16x3")
    with pytest.raises(GamdError, match="fp16 range"):
        eng.forward(posw)
    eng.close()


# ---------------------------------------------------------------------------------------------------------------
# host API details
# ---------------------------------------------------------------------------------------------------------------
def test_forward_returns_fresh_tensors_unless_inplace():
    """The reference's pnet_model(...) returns a new tensor per call; holding f_prev and f_new must work."""
    pos, box = workloads.lj_box(600, seed=2)
    sd = make_state_dict(ModelConfig(kind="lj"), 2, 5.0, 1.7)
    eng = _engine(sd, 600, box, 7.5)
    p1 = torch.from_numpy(pos).float()
    p2 = torch.from_numpy(pos + np.random.default_rng(0).normal(0, 0.1, pos.shape)).float()
    f1 = eng.forward(p1)
    keep = f1.clone()
    f2 = eng.forward(p2)
    assert f1.data_ptr() != f2.data_ptr() and torch.equal(f1, keep) and not torch.equal(f1, f2)
    a = eng.forward(p1, inplace=True)
    b = eng.forward(p2, inplace=True)
    assert a.data_ptr() == b.data_ptr()                                 # the documented fast path aliases
    e1 = eng.forward_edges(p1, torch.from_numpy(eng.debug_edges()))
    assert e1.data_ptr() != a.data_ptr()
    from gamd_amd.compat import ParticleNetLightningLJ
    m = ParticleNetLightningLJ(state_dict=sd, num_atoms=600, box_size=box, cutoff=7.5)
    edges = torch.from_numpy(eng.debug_edges()).long().cuda()
    o1 = m.pnet_model([p1.cuda()], [edges])
    o1c = o1.clone()
    m.pnet_model([p2.cuda()], [edges])
    assert torch.equal(o1, o1c)
    eng.close()


def test_current_device_is_left_alone():
    before = torch.cuda.current_device()
    pos, box = workloads.lj_box(300, seed=2)
    eng = _engine(make_state_dict(ModelConfig(kind="lj"), 2, 5.0, 1.7), 300, box, 7.5, device=0)
    eng.forward(torch.from_numpy(pos).float())
    eng.close()
    assert torch.cuda.current_device() == before
    from gamd_amd._lib import GamdError
    with pytest.raises(GamdError, match="out of range"):
        _engine(make_state_dict(ModelConfig(kind="lj"), 2, 5.0, 1.7), 300, box, 7.5, device=torch.cuda.device_count() + 3)


def test_predict_forces_verbose_prints_the_two_reference_buckets(capsys):
    """train_network_lj.py:134-151: 'Nbr search used time' and 'Force eval used time', same result as the quiet call."""
    from gamd_amd.compat import ParticleNetLightningLJ
    g, cfg, sd = load_golden("lj258_pert_seed1")
    m = ParticleNetLightningLJ(state_dict=sd)
    m.training_mean, m.training_var = g["scaler_mean"], g["scaler_var"]
    quiet = m.predict_forces(g["pos"])
    assert capsys.readouterr().out == ""
    loud = m.predict_forces(g["pos"], verbose=True)
    out = capsys.readouterr().out.splitlines()
    assert out[0].startswith("====") and out[1].startswith("Nbr search used time: ") and out[2].startswith("Force eval used time: ")
    t_nbr, t_force = float(out[1].split(": ")[1]), float(out[2].split(": ")[1])
    assert 0.0 < t_nbr < 0.05 and 0.0 < t_force < 0.05
    assert rel_err(loud, quiet) < 1e-6 and rel_err(loud, g["forces"]) < TOL


def test_non_uniform_rbf_centres_keep_the_exact_form():
    """The encoder evaluates the 40 RBFs by recurrence along the uniform grid linspace(0, 1, 40) (nn_module.py:237-240).
    `edge_expand.centers` is in the state_dict: if someone hands over other centres the kernel must fall back to one
    exponential per centre.  Both paths against the oracle, stage by stage."""
    g, cfg, sd = load_golden("lj258_seed0")
    box, rc = float(g["box"]), float(g["cutoff"])
    posw = torch.from_numpy(np.mod(g["pos"], box)).float()
    for tag in ("uniform", "warped"):
        sd2 = dict(sd)
        if tag == "warped":
            c = sd["edge_expand.centers"].clone()
            sd2["edge_expand.centers"] = (c + 0.01 * torch.sin(7.0 * c)).float()
        for ks in (0, 1):
            eng = _engine(sd2, 258, box, rc, keep_stages=True, kernel_select=ks)
            out = eng.forward(posw).cpu().numpy()
            edges = torch.from_numpy(eng.debug_edges()).long()
            st = {}
            ref = orc.forward(sd2, posw, edges, box, stages=st).numpy()
            assert rel_err(eng.debug_feat(44), st["feat"].numpy()) < TOL, (tag, ks)
            assert rel_err(eng.debug_e(), st["e"].numpy()) < TOL, (tag, ks)
            assert rel_err(out, ref) < TOL, (tag, ks)
            eng.close()


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 5, the other reading: 8 000 NETWORK atoms (SURVEY section 8 flags the ambiguity)
# ---------------------------------------------------------------------------------------------------------------
def test_c5_8001_network_atoms_bf16_within_restated_tolerance():
    """2 667 molecules = 8 001 network atoms (8 000 is not a multiple of 3), bf16 edge-MLP operands, fp32 accumulate:
    same edge set as the fp32 path, forces within the restated 1e-2 of it and of the oracle."""
    pos, box, species, bonds = workloads.water_box(2667, mol_per_20A3=251.0, seed=3456)
    sd = make_state_dict(ModelConfig(kind="water", use_bond=True), 3, 2.9, 1.1)
    p = torch.from_numpy(pos).float()
    e32 = _engine(sd, pos.shape[0], box, 4.2, bond=bonds)
    e16 = _engine(sd, pos.shape[0], box, 4.2, bond=bonds, edge_dtype="bf16")
    a, b = e32.forward(p, species=species).cpu().numpy(), e16.forward(p, species=species).cpu().numpy()
    assert pos.shape[0] == 8001 and np.array_equal(edge_set(e16.debug_edges()), edge_set(e32.debug_edges()))
    assert 1e-5 < rel_err(b, a) < 1e-2
    ref = orc.forward(sd, p, torch.from_numpy(e32.debug_edges()).long(), box,
                      feat=torch.from_numpy(species.astype(np.float32)).view(-1, 1), bond=bonds).numpy()
    assert rel_err(a, ref) < TOL and rel_err(b, ref) < 1e-2
    e32.close(); e16.close()
