"""Round-6 GPU tests (all through the C ABI):

* the generic-width fp32 conv layer on 16-edge work units (wide16.hip, v_mfma_f32_16x16x4_f32) is BIT-IDENTICAL to the 32-edge
  kernel (wide.hip, 32x32x2) — forces, every residual stream h_l, MD trajectories — on the reference-generated wide goldens, the
  DFT-water configuration, batches, edge counts with every kind of last chunk (opt-in through GAMD_KSEL_FORCE_HALF_QUANTUM: its
  half-size work quantum does not pay for its per-phase costs, profiles/r06_experiments.md);
* layer 0's node tables survive between candidate rebuilds inside an enqueued MD run (the first node launch returns at once):
  a run in one gamd_md_run call equals the same run in many calls (every call recomputes them) bit for bit, also across
  rebuilds, for free atoms, rigid water, Nose-Hoover, a batch and the generic-width model.
"""
import numpy as np
import pytest
import torch

from helpers import load_golden, rel_err
from gamd_amd import workloads
from gamd_amd.engine import GamdForce
from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS

pytestmark = pytest.mark.gpu
FORCE_GENERIC, FORCE_HALF, NO_HALF = 1, 2, 0       # include/gamd_hip.h: GAMD_KSEL_* (the 16-edge kernel is opt-in)


def _wide_golden(name, ksel, **kw):
    g, cfg, sd = load_golden(name)
    box, rc, n = g["box"], float(g["cutoff"]), g["pos"].shape[0]
    bond = g["bond"] if "bond" in g else None
    species = (g["node_feat"].reshape(-1) != 0) if "node_feat" in g else None
    flav = "torch" if name.startswith("dynbox") else "jaxmd"
    if "scaler_mean" in g:
        kw["scaler"] = (g["scaler_mean"], g["scaler_var"])
    eng = GamdForce(sd, n, box, rc, bond=bond, cfg=cfg, nbr_flavour=flav, keep_stages=True, kernel_select=ksel, **kw)
    x = torch.from_numpy(np.asarray(g["pos"])).float()
    return g, cfg, eng, x, species


@pytest.mark.parametrize("name", ["lj258_w256_seed9", "tip3p774_w256_seed10", "tip3p774_bn_w256_seed12", "dynbox384_dftcfg_seed5",
                                  "dynbox384_h256_e128_seed7", "dynbox384_h128_e256_noexpand_seed8"])
def test_16_edge_conv_kernel_is_bit_identical_to_the_32_edge_kernel_on_the_wide_goldens(name):
    out = {}
    for ksel in (NO_HALF, FORCE_HALF):
        g, cfg, eng, x, species = _wide_golden(name, ksel)
        f = eng.forward(x, species=species).cpu().numpy().copy()
        hs = [eng.debug_h(l).copy() for l in range(cfg.conv_layer + 1)]
        out[ksel] = (f, hs, eng.counts()[0])
        eng.close()
    (fa, ha, ea), (fb, hb, eb) = out[NO_HALF], out[FORCE_HALF]
    assert ea == eb and np.array_equal(fa, fb)
    for l, (p, q) in enumerate(zip(ha, hb)):
        assert np.array_equal(p, q), l
    ref = g["out_norm"]
    assert np.abs(fb - ref).max() / np.abs(ref).max() < 1e-5


def test_16_edge_conv_kernel_on_a_128_wide_model_with_every_kind_of_last_chunk():
    """128 / 128 / 128 forced onto the generic-width kernels (EHT = HT = 1): edge counts 32 k + r for many r (a last tile whose
    second chunk is empty, partial, full), isolated atoms (rows of one self edge): bit-identical to the 32-edge generic kernel, and
    within fp32 rounding of the specialised 128-wide kernels (whose node side sums in another order)."""
    sd = make_state_dict(ModelConfig(kind="lj"), 4, 5.0, 1.7)
    rng = np.random.default_rng(5)
    for n in (37, 64, 131, 200, 333):
        box = 6.0 * n ** (1.0 / 3.0)
        pos = rng.uniform(0, box, (n, 3))
        pos[: n // 8] += 1000.0 * np.arange(n // 8)[:, None]                  # a few far-away images: wrapped by the search
        res = {}
        for ksel in (0, FORCE_GENERIC, FORCE_GENERIC | FORCE_HALF):
            eng = GamdForce(sd, n, box, 5.5, scaler=SHIPPED_SCALERS["lj"], kernel_select=ksel)
            res[ksel] = (eng.forward(torch.from_numpy(pos).float()).cpu().numpy().copy(), eng.counts()[0])
            eng.close()
        base, g32, g16 = res[0], res[FORCE_GENERIC], res[FORCE_GENERIC | FORCE_HALF]
        assert g32[1] == g16[1] == base[1] and np.array_equal(g32[0], g16[0]), (n, base[1])
        assert np.abs(g16[0] - base[0]).max() / np.abs(base[0]).max() < 1e-5


def test_16_edge_conv_kernel_dft_configuration_md_run():
    """The 774-atom DFT-water configuration (256 / 128 / 256 x 5 layers, bohr): 1 470 tiles on 1 024 SIMDs, the size the 16-edge
    kernel was built for.  100 Langevin steps with skin reuse: 16-edge == 32-edge bit for bit (timings printed)."""
    import time
    from gamd_amd.compat import HARTREE_PER_BOHR_TO_KJ_PER_MOL_NM as CONV
    bohr = workloads.BOHR_PER_NM / 10.0
    pos, box, species, bonds = workloads.water_box(258, seed=4567, jitter=0.0, wrap=False)
    pos, box = pos * bohr, box * bohr
    cfg = ModelConfig(kind="dynbox", encoding_size=256, hidden_dim=128, edge_embedding_dim=256, conv_layer=5)
    sd = make_state_dict(cfg, 5, 3.1 * bohr, 1.2 * bohr)
    mean, var = SHIPPED_SCALERS["dft"]
    md = dict(dt_ps=0.0005, mass_amu=workloads.MASS_O, mass_h_amu=workloads.MASS_H, temperature_k=300.0, length_per_nm=workloads.BOHR_PER_NM,
              rigid_water=True, r_oh=workloads.TIP3P_R_OH * bohr, r_hh=workloads.TIP3P_R_HH * bohr, species=species)
    vel = workloads.maxwell_boltzmann(pos.shape[0], mass_amu=workloads.MASS_O, seed=9) * bohr
    res, ms = {}, {}
    for ksel in (NO_HALF, FORCE_HALF):
        eng = GamdForce(sd, pos.shape[0], box, 9.5, nbr_flavour="torch", cfg=cfg, neighbor_skin=9.5 / 6.0, scaler=(mean * CONV, var * CONV ** 2),
                        kernel_select=ksel)
        x, v = torch.from_numpy(pos).float().cuda(), torch.from_numpy(vel).float().cuda()
        f = eng.forward(x, species=species, denormalize=True).clone()
        eng.md_run(x, v, f, 40, seed=3, **md)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.md_run(x, v, f, 60, seed=3, first_step=40, **md)
        torch.cuda.synchronize()
        ms[ksel] = (time.perf_counter() - t0) / 60 * 1e3
        res[ksel] = (x.cpu(), v.cpu(), f.cpu(), eng.counts()[0])
        eng.close()
    assert res[FORCE_HALF][3] == res[NO_HALF][3]
    for a, b in zip(res[FORCE_HALF][:3], res[NO_HALF][:3]):
        assert torch.equal(a, b)
    print("dft ms/step: 32-edge %.3f, 16-edge %.3f" % (ms[NO_HALF], ms[FORCE_HALF]))


def test_16_edge_conv_kernel_in_a_batch_of_boxes():
    g, cfg, sd = load_golden("tip3p774_w256_seed10")
    box, rc, n = float(g["box"]), float(g["cutoff"]), g["pos"].shape[0]
    species = (g["node_feat"].reshape(-1) != 0)
    rng = np.random.default_rng(1)
    pos = np.concatenate([np.mod(g["pos"] + rng.normal(0, 0.05 * b, g["pos"].shape), box) for b in range(3)])
    out = {}
    for ksel in (NO_HALF, FORCE_HALF):
        eng = GamdForce(sd, n, box, rc, bond=g["bond"], scaler=(g["scaler_mean"], g["scaler_var"]), cfg=cfg, n_boxes=3, kernel_select=ksel)
        out[ksel] = eng.forward(torch.from_numpy(pos).float(), species=np.tile(species, 3)).cpu().numpy().copy()
        eng.close()
    assert np.array_equal(out[NO_HALF], out[FORCE_HALF])


# ---- layer-0 node tables inside an enqueued run ---------------------------------------------------------------------------
def _run_split(eng, x0, v0, species, md, nhc, chunks):
    x, v = x0.clone(), v0.clone()
    f = eng.forward(x, species=species, denormalize=True).clone()
    state, done = None, 0
    for c in chunks:
        if nhc:
            state = eng.md_run_nhc(x, v, f, c, chain_state=state, **md)
        else:
            eng.md_run(x, v, f, c, first_step=done, **md)
        done += c
    return x.cpu(), v.cpu(), f.cpu(), eng.skin_stats()[0]


@pytest.mark.parametrize("case", ["lj", "water_rigid", "water_nhc", "batch", "wide"])
def test_layer0_tables_reused_inside_a_run_give_the_bits_of_a_run_in_single_steps(case):
    """gamd_md_run(n) recomputes layer 0's node tables on its first step and after every candidate rebuild only; n calls of one
    step recompute them every step.  Same trajectories bit for bit, over enough steps for several rebuilds."""
    nhc, species, kw, n_boxes = False, None, {}, 1
    if case in ("lj", "batch"):
        n = 1200
        n_boxes = 3 if case == "batch" else 1
        boxes = [workloads.lj_box(n, seed=50 + b) for b in range(n_boxes)]
        pos, box = np.concatenate([p for p, _ in boxes]), boxes[0][1]
        rc = 3.0 * workloads.LJ_SIGMA
        sd = make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2)
        scaler = SHIPPED_SCALERS["lj"]
        vel = np.concatenate([workloads.maxwell_boltzmann(n, temperature_k=400.0, seed=5 + b) for b in range(n_boxes)])
        md = dict(dt_ps=0.004, mass_amu=39.9, temperature_k=400.0, gamma_per_ps=5.0, seed=11)
        steps = 60
    else:
        wide = case == "wide"
        g, cfg, sd = load_golden("tip3p774_w256_seed10" if wide else "tip3p774_seed3")
        pos, box, species, bonds = workloads.water_box(258, seed=31, jitter=0.0, wrap=False)
        n, rc, scaler = pos.shape[0], 4.2, SHIPPED_SCALERS["tip3p"]
        kw = dict(bond=bonds, cfg=cfg)
        vel = workloads.maxwell_boltzmann(n, mass_amu=workloads.MASS_O, temperature_k=600.0, seed=2)
        md = dict(dt_ps=0.001, mass_amu=workloads.MASS_O, mass_h_amu=workloads.MASS_H, temperature_k=600.0, rigid_water=True,
                  r_oh=workloads.TIP3P_R_OH, r_hh=workloads.TIP3P_R_HH, species=species)
        nhc = case == "water_nhc"
        if not nhc:
            md.update(gamma_per_ps=5.0, seed=4)
        steps = 80
    eng = GamdForce(sd, n, box, rc, scaler=scaler, neighbor_skin=rc / 6.0, n_boxes=n_boxes, **kw)
    sp = None if species is None else species
    x0, v0 = torch.from_numpy(pos).float().cuda(), torch.from_numpy(vel).float().cuda()
    one = _run_split(eng, x0, v0, sp, md, nhc, [steps])
    eng.close()
    eng = GamdForce(sd, n, box, rc, scaler=scaler, neighbor_skin=rc / 6.0, n_boxes=n_boxes, **kw)
    many = _run_split(eng, x0, v0, sp, md, nhc, [1] * steps)
    eng.close()
    assert one[3] >= 3, one[3]                                        # the run really crossed candidate rebuilds
    for a, b in zip(one[:3], many[:3]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_host_buffer_boundary_with_one_synchronisation_gives_the_same_bits():
    """GamdForce.forward_host (pinned staging both ways, one stream synchronisation: what predict_forces runs on) against
    forward(): float64 and float32 host positions, a batch, the regrow-and-replay path, the dynamic-box model with a box per
    call, and predict_forces itself against the three-synchronisation form it replaced."""
    import os
    from gamd_amd.compat import ParticleNetLightningLJ
    g, cfg, sd = load_golden("lj258_seed0")
    box, rc, n = float(g["box"]), float(g["cutoff"]), 258
    posw = np.mod(g["pos"], box)
    eng = GamdForce(sd, n, box, rc, scaler=(g["scaler_mean"], g["scaler_var"]), neighbor_skin=rc / 6.0)
    ref = eng.forward(torch.from_numpy(posw).float()).cpu().numpy()
    assert rel_err(ref, g["out_norm"]) < 1e-5
    for p in (posw.astype(np.float64), posw.astype(np.float32)):
        assert np.array_equal(eng.forward_host(p), ref)
    den = eng.forward(torch.from_numpy(posw).float(), denormalize=True).cpu().numpy()
    assert np.array_equal(eng.forward_host(posw, denormalize=True), den)
    with pytest.raises(ValueError, match="pos must be"):
        eng.forward_host(posw[:-1])
    eng.close()
    # fresh handles driven from a non-blocking side stream from their first call on (the stream contract of include/gamd_hip.h):
    # the staging copies, the kernels and the one synchronisation all belong to that stream
    for k in range(8):
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            e2 = GamdForce(sd, n, box, rc, scaler=(g["scaler_mean"], g["scaler_var"]), neighbor_skin=rc / 6.0 if k & 1 else 0.0)
            assert np.array_equal(e2.forward_host(posw), ref), k
            assert np.array_equal(e2.forward_host(posw, denormalize=True), den), k
            e2.close()
    # a batch; a capacity far too small: detected on the device, regrown, replayed inside the one call
    batch = GamdForce(sd, n, box, rc, n_boxes=2, edge_capacity=64)
    two = np.concatenate([posw, np.mod(posw + 0.3, box)])
    out = batch.forward_host(two).copy()
    assert batch.last_status == 1
    assert np.array_equal(out[:n], ref) and np.array_equal(batch.forward_host(two.reshape(2, n, 3)), out) and batch.last_status == 0
    batch.close()
    # predict_forces: same float64 bits as the form with three synchronisations
    m = ParticleNetLightningLJ(state_dict=sd)
    a = m.predict_forces(g["pos"])
    os.environ["GAMD_PREDICT_LEGACY"] = "1"
    try:
        b = m.predict_forces(g["pos"])
    finally:
        del os.environ["GAMD_PREDICT_LEGACY"]
    assert a.dtype == np.float64 and np.array_equal(a, b) and a is not b
    a2 = m.predict_forces(g["pos"] + 0.01)
    assert not np.array_equal(a, a2) and np.array_equal(a, b)             # the first result does not alias the staging buffer
