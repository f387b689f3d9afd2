"""On-device split BAOAB (hack_integrator.py:141-165,175-178) against the oracle's restatement."""
import numpy as np
import pytest
import torch

import gamd_oracle as orc
from helpers import load_golden, rel_err
from gamd_amd.weights import SHIPPED_SCALERS

pytestmark = pytest.mark.gpu


def test_deterministic_steps_match_oracle():
    """T = 0 K removes the noise term, so x/v after a few steps are comparable exactly (to fp32
    rounding) with the oracle integrator driven by oracle forces."""
    from gamd_amd.engine import GamdForce
    g, cfg, sd = load_golden("lj258_seed0")
    box, rc, n = float(g["box"]), float(g["cutoff"]), 258
    eng = GamdForce(sd, n, box, rc, scaler=SHIPPED_SCALERS["lj"])
    x = torch.from_numpy(np.mod(g["pos"], box)).float().cuda()
    v = torch.from_numpy(np.random.default_rng(1).normal(0, 1.4, (n, 3))).float().cuda()
    f = eng.forward(x, denormalize=True).clone()
    xr, vr, fr = x.cpu().double().numpy(), v.cpu().double().numpy(), f.cpu().double().numpy()
    dt, m, gamma = 0.002, 39.9, 25.0
    steps = 3
    eng.md_run(x, v, f, steps, dt_ps=dt, mass_amu=m, temperature_k=0.0, gamma_per_ps=gamma)
    a = np.exp(-gamma * dt)
    mean, var = SHIPPED_SCALERS["lj"]
    for _ in range(steps):
        xr, vr = orc.baoab_first_half(xr, vr, fr, 10.0 / m, dt, a, 0.0, 0.0)
        xr = np.mod(xr, box)
        fr = orc.predict_forces(sd, xr, box, rc, var=var, mean=mean)
        vr = orc.baoab_second_half(vr, fr, 10.0 / m, dt)
    assert rel_err(x.cpu().numpy(), xr) < 1e-5
    assert rel_err(v.cpu().numpy(), vr) < 1e-4
    assert rel_err(f.cpu().numpy(), fr) < 1e-4
    eng.close()


def test_ou_noise_statistics():
    """gamma*dt >> 1 makes v = b*sigma*xi: check mean 0, variance kT/m, independence of components,
    and reproducibility from (seed, step)."""
    from gamd_amd.engine import GamdForce
    from gamd_amd.weights import ModelConfig, make_state_dict
    from gamd_amd.workloads import lj_box, KB
    n = 4096
    pos, box = lj_box(n)
    eng = GamdForce(make_state_dict(ModelConfig(), 0, 7.0, 2.2), n, box, 7.5)
    T, m = 300.0, 39.9

    def run(seed, first):
        x = torch.from_numpy(pos).float().cuda()
        v = torch.zeros(n, 3, device="cuda")
        f = torch.zeros(n, 3, device="cuda")
        eng.md_run(x, v, f, 1, dt_ps=1e-6, mass_amu=m, temperature_k=T, gamma_per_ps=1e8, seed=seed, first_step=first)
        return v.cpu().numpy()

    v = run(11, 0)
    sigma = 10.0 * np.sqrt(KB * T / m)
    assert abs(v.mean()) < 4 * sigma / np.sqrt(v.size)
    assert abs(v.std() / sigma - 1.0) < 0.02
    assert abs(np.corrcoef(v[:, 0], v[:, 1])[0, 1]) < 0.05 and abs(np.corrcoef(v[:, 0], v[:, 2])[0, 1]) < 0.05
    assert np.array_equal(v, run(11, 0))
    assert not np.array_equal(v, run(11, 1)) and not np.array_equal(v, run(12, 0))
    eng.close()


def test_nose_hoover_chain_matches_oracle():
    """Split NHC step of the reference drivers (hack_integrator.py:182-493, test_nosehoover.py:100-118):
    chain_length 10, 5 multi-time-steps, Yoshida-Suzuki 5; deterministic, so comparable step by step."""
    from gamd_amd.engine import GamdForce
    from gamd_amd.workloads import KB
    g, cfg, sd = load_golden("lj258_seed0")
    box, rc, n = float(g["box"]), float(g["cutoff"]), 258
    eng = GamdForce(sd, n, box, rc, scaler=SHIPPED_SCALERS["lj"])
    x = torch.from_numpy(np.mod(g["pos"], box)).float().cuda()
    v = torch.from_numpy(np.random.default_rng(3).normal(0, 1.44, (n, 3))).float().cuda()
    f = eng.forward(x, denormalize=True).clone()
    xr, vr, fr = x.cpu().double().numpy(), v.cpu().double().numpy(), f.cpu().double().numpy()
    dt, m, freq, T, M = 0.002, 39.9, 25.0, 100.0, 10
    steps = 3
    chain = eng.md_run_nhc(x, v, f, steps, dt_ps=dt, mass_amu=m, temperature_k=T, frequency_per_ps=freq, chain_length=M)
    st = orc.nhc_init(M, freq)
    kT, ndf = KB * T, 3.0 * n
    mean, var = SHIPPED_SCALERS["lj"]
    for _ in range(steps):
        xr, vr = orc.nhc_first_half(st, xr, vr, fr, m, dt, kT, freq, ndf)
        xr = np.mod(xr, box)
        fr = orc.predict_forces(sd, xr, box, rc, var=var, mean=mean)
        vr = orc.nhc_second_half(st, vr, fr, m, dt, kT, freq, ndf)
    assert rel_err(x.cpu().numpy(), xr) < 1e-5
    assert rel_err(v.cpu().numpy(), vr) < 1e-4
    c = chain.cpu().numpy()
    assert rel_err(c[M:2 * M], st["vxi"]) < 1e-4 and rel_err(c[:M], st["xi"]) < 1e-4
    assert abs(c[3 * M] - 1.0) < 0.05                       # last velocity scale stays close to 1
    # continuing with the returned state is the same as one longer run
    x2 = torch.from_numpy(np.mod(g["pos"], box)).float().cuda()
    v2 = torch.from_numpy(np.random.default_rng(3).normal(0, 1.44, (n, 3))).float().cuda()
    f2 = eng.forward(x2, denormalize=True).clone()
    ch = eng.md_run_nhc(x2, v2, f2, 1, dt_ps=dt, mass_amu=m, temperature_k=T, frequency_per_ps=freq, chain_length=M)
    eng.md_run_nhc(x2, v2, f2, 2, chain_state=ch, dt_ps=dt, mass_amu=m, temperature_k=T, frequency_per_ps=freq, chain_length=M)
    assert np.array_equal(x2.cpu().numpy(), x.cpu().numpy()) and np.array_equal(v2.cpu().numpy(), v.cpu().numpy())
    eng.close()


# ---- rigid water (SETTLE) and per-species masses ---------------------------------------------------
def _water_setup(n_mol=64, seed=5):
    from gamd_amd.engine import GamdForce
    from gamd_amd import workloads as wl
    g, cfg, sd = load_golden("tip3p774_seed3")               # water weights with the bond feature
    pos, box, species, bonds = wl.water_box(n_mol, seed=seed, jitter=0.0, wrap=False)
    n = 3 * n_mol
    eng = GamdForce(sd, n, box, 4.2, bond=bonds, scaler=SHIPPED_SCALERS["tip3p"])
    mass = np.where(species == 1, wl.MASS_O, wl.MASS_H).astype(np.float64).reshape(-1, 1)
    pairs, lengths = orc.water_constraints(n, wl.TIP3P_R_OH, wl.TIP3P_R_HH)
    rng = np.random.default_rng(seed + 1)
    v0 = rng.normal(0, 1.0, (n, 3)) * 10.0 * np.sqrt(wl.KB * 300.0 / mass)
    v0 = orc.rattle_velocities(pos, v0, (1.0 / mass).reshape(-1), pairs)       # start on the constraint manifold
    return eng, sd, pos, box, species, bonds, mass, pairs, lengths, v0


def _oracle_water_forces(sd, x, box, species, bonds):
    mean, var = SHIPPED_SCALERS["tip3p"]
    feat = torch.from_numpy(species.astype(np.float32)).view(-1, 1)
    return orc.predict_forces(sd, x, box, 4.2, var=var, mean=mean, feat=feat, bond=bonds)


def _bond_errors(x, v, pairs, lengths):
    r = x[pairs[:, 0]] - x[pairs[:, 1]]
    d = np.linalg.norm(r, axis=1)
    rv = np.sum(r * (v[pairs[:, 0]] - v[pairs[:, 1]]), axis=1) / d
    return np.abs(d / lengths - 1.0).max(), np.abs(rv).max()


def test_rigid_water_langevin_matches_shake_rattle_oracle():
    """The water drivers integrate rigid molecules (OpenMM constraints at hack_integrator.py:145-164,178).
    Device: SETTLE + analytic velocity constraint, one molecule per thread; oracle: SHAKE/RATTLE iterated to
    convergence in f64 (the constrained update is unique).  The device noise of each step is recovered from an
    unconstrained run with the same (seed, step) so the stochastic step is compared too."""
    from gamd_amd import workloads as wl
    eng, sd, pos, box, species, bonds, mass, pairs, lengths, v0 = _water_setup()
    n = pos.shape[0]
    dt, gamma, T, seed = 0.002, 25.0, 300.0, 17
    a = np.exp(-gamma * dt)
    bs = np.sqrt(1.0 - a * a) * 10.0 * np.sqrt(wl.KB * T / mass)
    kw = dict(dt_ps=dt, mass_amu=wl.MASS_O, mass_h_amu=wl.MASS_H, temperature_k=T, gamma_per_ps=gamma, seed=seed,
              species=species)

    def device_noise(step):
        x = torch.from_numpy(pos).float().cuda()
        v = torch.zeros(n, 3, device="cuda")
        f = torch.zeros(n, 3, device="cuda")
        eng.md_run(x, v, f, 1, first_step=step, **kw)                  # v = bs*xi + (dt/2)(10/m) f_new
        return (v.cpu().double().numpy() - 0.5 * dt * 10.0 / mass * f.cpu().double().numpy()) / bs

    x = torch.from_numpy(pos).float().cuda()
    v = torch.from_numpy(v0).float().cuda()
    f = eng.forward(x, species=species, denormalize=True).clone()
    xr, vr, fr = x.cpu().double().numpy(), v.cpu().double().numpy(), f.cpu().double().numpy()
    steps = 2
    eng.md_run(x, v, f, steps, rigid_water=True, r_oh=wl.TIP3P_R_OH, r_hh=wl.TIP3P_R_HH, **kw)
    for s in range(steps):
        vr = orc.remove_cm_motion(vr, mass)                           # the WaterBox System's CMMotionRemover (:142); default on
        xr, vr = orc.baoab_first_half_rigid(xr, vr, fr, 1.0 / mass, dt, a, bs, device_noise(s), pairs, lengths)
        fr = _oracle_water_forces(sd, xr, box, species, bonds)
        vr = orc.baoab_second_half_rigid(xr, vr, fr, 1.0 / mass, dt, pairs)
    xd, vd = x.cpu().double().numpy(), v.cpu().double().numpy()
    # device keeps molecules whole but may translate them by a lattice vector
    shift = np.round((xd - xr) / box) * box
    assert np.abs(shift.reshape(-1, 3, 3) - shift.reshape(-1, 3, 3)[:, :1]).max() == 0.0
    assert np.abs(xd - shift - xr).max() < 2e-5                      # Angstrom (fp32 ulp at 12 A is 1e-6)
    assert rel_err(vd, vr) < 2e-4
    assert rel_err(f.cpu().numpy(), fr) < 1e-4
    eng.close()


def test_rigid_water_stays_rigid_over_many_steps():
    from gamd_amd import workloads as wl
    eng, sd, pos, box, species, bonds, mass, pairs, lengths, v0 = _water_setup(n_mol=125, seed=9)
    x = torch.from_numpy(pos).float().cuda()
    v = torch.from_numpy(v0).float().cuda()
    f = eng.forward(x, species=species, denormalize=True).clone()
    eng.md_run(x, v, f, 200, dt_ps=0.002, mass_amu=wl.MASS_O, mass_h_amu=wl.MASS_H, temperature_k=300.0,
               gamma_per_ps=25.0, seed=3, species=species, rigid_water=True, r_oh=wl.TIP3P_R_OH, r_hh=wl.TIP3P_R_HH)
    xd, vd = x.cpu().double().numpy(), v.cpu().double().numpy()
    assert np.isfinite(xd).all() and np.isfinite(vd).all()
    d_err, rv = _bond_errors(xd, vd, pairs, lengths)
    assert d_err < 2e-5                                               # no drift: SETTLE re-solves from the target lengths
    assert rv < 2e-3 * np.abs(vd).max()
    o = xd[0::3]
    assert (o >= 0).all() and (o < box).all()                         # molecules are wrapped by their oxygen, kept whole
    assert np.abs(xd[1::3] - o).max() < 1.0 and np.abs(xd[2::3] - o).max() < 1.0
    eng.close()


@pytest.mark.parametrize("n_mol", [125, 512])
def test_rigid_water_halves_fused_into_the_skin_check_match_the_separate_kernels(n_mol):
    """Skin mode: the rigid-molecule B / B A O A halves ride in the first neighbour kernel of the step (k_step_small up to
    1024 atoms, k_skin_check above), one thread per molecule, which then runs the displacement check of its three atoms.
    Same device functions and noise stream as k_baoab_*_rigid: the run equals the un-fused engine's (no skin) to the
    rounding of the row order, step-by-step calls equal one call bit for bit, and the molecules stay rigid."""
    from gamd_amd.engine import GamdForce
    from gamd_amd import workloads as wl
    eng0, sd, pos, box, species, bonds, mass, pairs, lengths, v0 = _water_setup(n_mol=n_mol, seed=9)
    eng0.close()
    n = 3 * n_mol
    kw = dict(dt_ps=0.002, mass_amu=wl.MASS_O, mass_h_amu=wl.MASS_H, temperature_k=300.0, gamma_per_ps=25.0, seed=3,
              species=species, rigid_water=True, r_oh=wl.TIP3P_R_OH, r_hh=wl.TIP3P_R_HH)
    out = {}
    for tag, ekw, chunks in (("unfused", {}, [8]), ("fused", dict(neighbor_skin=0.7), [8]),
                             ("fused_split", dict(neighbor_skin=0.7), [1, 3, 4])):
        eng = GamdForce(sd, n, box, 4.2, bond=bonds, scaler=SHIPPED_SCALERS["tip3p"], **ekw)
        x = torch.from_numpy(pos).float().cuda()
        v = torch.from_numpy(v0).float().cuda()
        f = eng.forward(x, species=species, denormalize=True).clone()
        done = 0
        for c in chunks:
            eng.md_run(x, v, f, c, first_step=done, **kw)
            done += c
        out[tag] = (x.cpu().numpy(), v.cpu().numpy(), f.cpu().numpy())
        eng.close()
    for a, b in zip(out["fused"], out["fused_split"]):
        assert np.array_equal(a, b)
    xu, xf = out["unfused"][0].astype(np.float64), out["fused"][0].astype(np.float64)
    d = xf - xu
    d -= box * np.round(d / box)
    # rounding of the row order (1e-6 of the forces), amplified by 8 steps of a stiff random-weight force field
    assert np.abs(d).max() < 2e-4                                     # Angstrom
    assert rel_err(out["fused"][1], out["unfused"][1]) < 5e-4
    d_err, rv = _bond_errors(xf, out["fused"][1].astype(np.float64), pairs, lengths)
    assert d_err < 2e-5 and rv < 2e-3 * np.abs(out["fused"][1]).max()
    assert np.abs(xf - pos).max() > 1e-3                              # it moved


@pytest.mark.parametrize("remove_com", [False, True])
def test_rigid_water_nose_hoover_matches_oracle(remove_com):
    """HackNoseHooverIntegrator / HackHalfNoseHooverIntegrator with constraints (hack_integrator.py:274-280,
    427-430), ndf = 3N - N constraints (:226-235), minus 3 and with the COM motion removed at the top of every step (:272)
    when the System carries a CMMotionRemover (the default for rigid water)."""
    from gamd_amd import workloads as wl
    eng, sd, pos, box, species, bonds, mass, pairs, lengths, v0 = _water_setup()
    n = pos.shape[0]
    dt, freq, T, M = 0.002, 25.0, 300.0, 10
    x = torch.from_numpy(pos).float().cuda()
    v = torch.from_numpy(v0).float().cuda()
    f = eng.forward(x, species=species, denormalize=True).clone()
    xr, vr, fr = x.cpu().double().numpy(), v.cpu().double().numpy(), f.cpu().double().numpy()
    steps = 3
    chain = eng.md_run_nhc(x, v, f, steps, dt_ps=dt, mass_amu=wl.MASS_O, mass_h_amu=wl.MASS_H, temperature_k=T,
                           frequency_per_ps=freq, chain_length=M, species=species, rigid_water=True,
                           r_oh=wl.TIP3P_R_OH, r_hh=wl.TIP3P_R_HH, **({} if remove_com else dict(remove_cm_motion=False)))
    st = orc.nhc_init(M, freq)
    kT, ndf = wl.KB * T, 2.0 * n - (3.0 if remove_com else 0.0)
    for _ in range(steps):
        xr, vr = orc.nhc_first_half_rigid(st, xr, vr, fr, mass, dt, kT, freq, ndf, pairs, lengths, remove_com=remove_com)
        fr = _oracle_water_forces(sd, xr, box, species, bonds)
        vr = orc.nhc_second_half_rigid(st, xr, vr, fr, mass, dt, kT, freq, ndf, pairs)
    xd, vd = x.cpu().double().numpy(), v.cpu().double().numpy()
    shift = np.round((xd - xr) / box) * box
    assert np.abs(xd - shift - xr).max() < 2e-5
    assert rel_err(vd, vr) < 2e-4
    c = chain.cpu().numpy()
    assert rel_err(c[M:2 * M], st["vxi"]) < 1e-3 and rel_err(c[:M], st["xi"]) < 1e-3
    eng.close()


def test_species_masses_and_bohr_units_with_the_dft_model():
    """DFT-water rollouts (water/test_script/test_nosehoover_hb.py:106-123) hand the network positions in bohr and
    convert its output with 2625.5/0.0529177 to kJ/mol/nm: on device that is length_per_nm = bohr per nm, the
    conversion folded into the scaler, and O/H masses by species.  T = 0, unconstrained, against the oracle."""
    from gamd_amd.engine import GamdForce
    from gamd_amd import workloads as wl
    from gamd_amd.compat import HARTREE_PER_BOHR_TO_KJ_PER_MOL_NM as CONV
    g, cfg, sd = load_golden("dynbox384_dftcfg_seed5")
    n, box, rc = g["pos"].shape[0], g["box"], float(g["cutoff"])
    mean, var = SHIPPED_SCALERS["dft"]
    eng = GamdForce(sd, n, box, rc, nbr_flavour="torch", cfg=cfg, scaler=(mean * CONV, var * CONV ** 2))
    species = g["node_feat"].reshape(-1) != 0
    mass = np.where(species, wl.MASS_O, wl.MASS_H).astype(np.float64).reshape(-1, 1)
    x = torch.from_numpy(g["pos"]).float().cuda()
    v = torch.from_numpy(np.random.default_rng(2).normal(0, 8.0, (n, 3))).float().cuda()
    f = eng.forward(x, box=box, species=species, denormalize=True).clone()
    xr, vr, fr = x.cpu().double().numpy(), v.cpu().double().numpy(), f.cpu().double().numpy()
    dt, gamma, L = 0.0005, 25.0, wl.BOHR_PER_NM
    steps = 2
    eng.md_run(x, v, f, steps, dt_ps=dt, mass_amu=wl.MASS_O, mass_h_amu=wl.MASS_H, temperature_k=0.0, gamma_per_ps=gamma,
               species=species, box=box, length_per_nm=L)
    a = np.exp(-gamma * dt)
    feat = torch.from_numpy(g["node_feat"])
    for _ in range(steps):
        xr, vr = orc.baoab_first_half(xr, vr, fr, L / mass, dt, a, 0.0, 0.0)
        xr = np.mod(xr, box.astype(np.float64))
        out = orc.forward_dynamic_box(sd, torch.from_numpy(xr).float(), feat, box, rc).numpy().astype(np.float64)
        fr = (out * np.sqrt(var) + mean) * CONV
        vr = orc.baoab_second_half(vr, fr, L / mass, dt)
    assert rel_err(x.cpu().numpy(), xr) < 1e-5
    assert rel_err(v.cpu().numpy(), vr) < 1e-4
    assert rel_err(f.cpu().numpy(), fr) < 1e-4
    eng.close()


# ---- centre-of-mass motion removal (hack_integrator.py:142, :272: addUpdateContextState -> CMMotionRemover) ------------
@pytest.mark.parametrize("n,skin_frac", [(258, 0.0), (258, 1.0 / 6.0), (2000, 0.0), (2000, 1.0 / 6.0)])
def test_com_motion_removal_langevin_free_atoms(n, skin_frac):
    """remove_cm_motion on the LJ system: T = 0 run against the oracle (COM removed at the top of every step), on the
    standalone integrator kernels (no skin), inside k_step_small (n <= 1024, skin) and inside k_skin_check (n > 1024, skin).
    """
    from gamd_amd.engine import GamdForce
    from gamd_amd.weights import ModelConfig, make_state_dict
    from gamd_amd.workloads import lj_box
    rc = 7.5
    pos, box = lj_box(n, seed=3)
    sd = make_state_dict(ModelConfig(), 0, 5.3, 1.6)
    eng = GamdForce(sd, n, box, rc, scaler=SHIPPED_SCALERS["lj"], neighbor_skin=skin_frac * rc)
    x = torch.from_numpy(pos).float().cuda()
    v = torch.from_numpy(np.random.default_rng(1).normal(0, 1.4, (n, 3)) + np.array([0.7, -0.4, 0.2])).float().cuda()
    f = eng.forward(x, denormalize=True).clone()
    xr, vr, fr = x.cpu().double().numpy(), v.cpu().double().numpy(), f.cpu().double().numpy()
    dt, m, gamma, steps = 0.002, 39.9, 25.0, 3
    eng.md_run(x, v, f, steps, dt_ps=dt, mass_amu=m, temperature_k=0.0, gamma_per_ps=gamma, remove_cm_motion=True)
    a = np.exp(-gamma * dt)
    mean, var = SHIPPED_SCALERS["lj"]
    for _ in range(steps):
        vr = orc.remove_cm_motion(vr, m)
        xr, vr = orc.baoab_first_half(xr, vr, fr, 10.0 / m, dt, a, 0.0, 0.0)
        xr = np.mod(xr, box)
        fr = orc.predict_forces(sd, xr, box, rc, var=var, mean=mean)
        vr = orc.baoab_second_half(vr, fr, 10.0 / m, dt)
    assert rel_err(x.cpu().numpy(), xr) < 1e-5
    assert rel_err(v.cpu().numpy(), vr) < 1e-4
    vcom = v.cpu().double().numpy().mean(axis=0)
    # the initial drift (0.7, -0.4, 0.2) is gone; what is left is what the last step's two half-kicks add (GNN forces do not
    # sum to zero), as in the oracle
    assert np.abs(vcom - vr.mean(axis=0)).max() < 1e-5 and np.abs(vcom).max() < 0.05
    # without the remover the drift stays
    x2 = torch.from_numpy(pos).float().cuda()
    v2 = torch.from_numpy(np.random.default_rng(1).normal(0, 1.4, (n, 3)) + np.array([0.7, -0.4, 0.2])).float().cuda()
    f2 = eng.forward(x2, denormalize=True).clone()
    eng.md_run(x2, v2, f2, steps, dt_ps=dt, mass_amu=m, temperature_k=0.0, gamma_per_ps=gamma)
    assert np.abs(v2.cpu().numpy().mean(axis=0)).max() > 0.1
    eng.close()


def test_com_motion_removal_is_mass_weighted_and_per_box():
    """Two species (O / H masses, unconstrained) in a batch of three boxes with different drifts: after one step with
    gamma dt >> 1 and T = 0 the velocities are the second half-kick only, so sum m v per box = (dt/2) sum f per box; and a
    Nose-Hoover batch removes each box's own drift too."""
    from gamd_amd.engine import GamdForce
    from gamd_amd import workloads as wl
    g, cfg, sd = load_golden("tip3p774_seed3")
    nb, n = 3, 774
    box, rc = float(g["box"]), float(g["cutoff"])
    species = g["node_feat"].reshape(-1) != 0
    mass = np.where(species, wl.MASS_O, wl.MASS_H).astype(np.float64).reshape(-1, 1)
    eng = GamdForce(sd, n, box, rc, bond=g["bond"], scaler=SHIPPED_SCALERS["tip3p"], n_boxes=nb)
    rng = np.random.default_rng(4)
    pos = np.concatenate([np.mod(g["pos"], box) + rng.normal(0, 0.02, (n, 3)) for _ in range(nb)])
    drift = np.array([[3.0, 0.0, -1.0], [0.0, -2.0, 0.5], [-1.5, 1.0, 2.0]])
    v0 = np.concatenate([rng.normal(0, 2.0, (n, 3)) + drift[b] for b in range(nb)])
    sp = np.tile(species, nb)
    kw = dict(dt_ps=0.0005, mass_amu=wl.MASS_O, mass_h_amu=wl.MASS_H, species=sp)
    x = torch.from_numpy(pos).float().cuda(); v = torch.from_numpy(v0).float().cuda()
    f = eng.forward(x, species=sp, denormalize=True).clone()
    # NHC, one step: propagateNHC on the velocities as they are (the drift counts as kinetic energy, hack_integrator.py:271),
    # then each box's COM velocity is removed (:272); compare with the oracle per box
    xn, vn, fn = x.clone(), v.clone(), f.clone()
    chain = eng.md_run_nhc(xn, vn, fn, 1, temperature_k=300.0, remove_cm_motion=True, ndf=3.0 * n - 3.0, **kw)
    for b in range(nb):
        sl = slice(b * n, (b + 1) * n)
        st = orc.nhc_init(10, 25.0)
        xr, vr = orc.nhc_first_half(st, pos[sl], v0[sl], f[sl].cpu().double().numpy(), mass, 0.0005, wl.KB * 300.0, 25.0,
                                    3.0 * n - 3.0, remove_com=True)
        fr = fn[sl].cpu().double().numpy()
        vr = orc.nhc_second_half(st, vr, fr, mass, 0.0005, wl.KB * 300.0, 25.0, 3.0 * n - 3.0)
        assert rel_err(vn[sl].cpu().numpy(), vr) < 1e-4, b
        assert rel_err(chain[b].cpu().numpy()[10:20], st["vxi"]) < 1e-3, b
    # Langevin: per-box momentum after a step with the remover
    eng.md_run(x, v, f, 1, temperature_k=0.0, gamma_per_ps=0.0, remove_cm_motion=True, **kw)
    vd, fd = v.cpu().double().numpy(), f.cpu().double().numpy()
    f_old = eng.forward(torch.from_numpy(pos).float(), species=sp, denormalize=True).cpu().double().numpy()
    for b in range(nb):
        sl = slice(b * n, (b + 1) * n)
        p = (mass * vd[sl]).sum(axis=0)
        expect = 0.5 * 0.0005 * 10.0 * (f_old[sl].sum(axis=0) + fd[sl].sum(axis=0))     # the two half-kicks; the drift is gone
        assert np.abs(p - expect).max() < 2e-3 * np.abs(mass * drift[b]).sum(), (b, p, expect)
    eng.close()
