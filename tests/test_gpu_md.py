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
