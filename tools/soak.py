#!/usr/bin/env python3
"""Long MD runs on the GPU box (not part of pytest): stability of the device loop and determinism of the kernels.

  1. C2 LJ box at 300 K, 4 000 steps with Verlet-skin reuse, run TWICE from the same state: the two trajectories must be
     bit-identical (the conv kernel's counted waits and register hand-offs are what a race would break), finite, and the
     skin engine's forces must match an exact-rebuild engine on the final positions;
  2. C3 rigid water, 3 000 steps: finite, bond lengths kept by SETTLE;
  3. C5 bf16, 3 000 steps: finite, forces within the restated tolerance of the fp32 path on the final positions;
  4. C2 split-fp16 (f16x3), 2 000 steps run TWICE: bit-identical (its weight copies are issued from inline assembly and
     waited for by hand: a missing wait shows as a run-to-run difference), forces within 1e-5 of the fp32 path at the end;
  5. a batch of 38 x 258-atom boxes, 2 000 steps run TWICE: bit-identical, every box finite, COM drift removed when asked;
  6. the DFT-water widths (256 / 128 / 256 x 5) in split-fp16 on rigid water, 1 000 steps run TWICE: bit-identical, forces within
     1e-5 of the fp32 path at the end."""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gamd_amd.engine import GamdForce
from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS
from gamd_amd import workloads as wk


def sha(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]


pos, box = wk.lj_box(10000, seed=77)
sd = make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2)
runs = []
for rep in range(2):
    eng = GamdForce(sd, 10000, box, 10.2, scaler=SHIPPED_SCALERS["lj"], neighbor_skin=1.7)
    x = torch.from_numpy(pos).float().cuda()
    v = torch.from_numpy(wk.maxwell_boltzmann(10000, 300.0)).float().cuda()
    f = eng.forward(x, denormalize=True).clone()
    t0 = time.perf_counter()
    for chunk in range(8):
        eng.md_run(x, v, f, 500, temperature_k=300.0, first_step=chunk * 500)
    dt = time.perf_counter() - t0
    runs.append((sha(x), sha(v), sha(f)))
    if rep == 0:
        exact = GamdForce(sd, 10000, box, 10.2, scaler=SHIPPED_SCALERS["lj"])
        f0 = exact.forward(x, denormalize=True)
        err = float((f0 - eng.forward(x, denormalize=True)).abs().max() / f0.abs().max())
        T = float((39.9 * (v.double() / 10) ** 2).sum() / (3 * 10000 * wk.KB))
        print(f"LJ f32: 4000 steps {dt:.1f} s ({dt / 4:.3f} ms/step) finite={bool(torch.isfinite(x).all() and torch.isfinite(v).all())} "
              f"T={T:.1f} K rebuilds={eng.skin_stats()[0]} E={eng.counts()[0]} skin-vs-exact force err={err:.2e}")
        exact.close()
    eng.close()
print("LJ f32: two runs bit-identical:", runs[0] == runs[1], runs[0])
assert runs[0] == runs[1]

posw, boxw, species, bonds = wk.water_box(1390, seed=5, jitter=0.0, wrap=False)
sdw = make_state_dict(ModelConfig(kind="water", use_bond=True), 3, 2.9, 1.1)
eng = GamdForce(sdw, posw.shape[0], boxw, 4.2, bond=bonds, scaler=SHIPPED_SCALERS["tip3p"], neighbor_skin=0.7)
x = torch.from_numpy(posw).float().cuda()
v = torch.zeros_like(x)
f = eng.forward(x, species=species, denormalize=True).clone()
for chunk in range(6):
    eng.md_run(x, v, f, 500, dt_ps=0.0005, mass_amu=wk.MASS_O, mass_h_amu=wk.MASS_H, temperature_k=300.0, species=species,
               rigid_water=True, r_oh=wk.TIP3P_R_OH, r_hh=wk.TIP3P_R_HH, first_step=chunk * 500)
xd = x.cpu().double().numpy()
d1 = np.linalg.norm(xd[1::3] - xd[0::3], axis=1)
d2 = np.linalg.norm(xd[2::3] - xd[1::3], axis=1)
print(f"water f32: 3000 steps finite={np.isfinite(xd).all()} max|d_OH/r-1|={np.abs(d1 / wk.TIP3P_R_OH - 1).max():.2e} "
      f"max|d_HH/r-1|={np.abs(d2 / wk.TIP3P_R_HH - 1).max():.2e} rebuilds={eng.skin_stats()[0]}")
assert np.isfinite(xd).all() and np.abs(d1 / wk.TIP3P_R_OH - 1).max() < 1e-4
eng.close()

posw, boxw, species, bonds = wk.water_box(2000, mol_per_20A3=251.0, seed=6, jitter=0.0, wrap=False)
engb = GamdForce(sdw, posw.shape[0], boxw, 4.2, bond=bonds, scaler=SHIPPED_SCALERS["tip4p"], neighbor_skin=0.7, edge_dtype="bf16")
x = torch.from_numpy(posw).float().cuda()
v = torch.zeros_like(x)
f = engb.forward(x, species=species, denormalize=True).clone()
for chunk in range(6):
    engb.md_run(x, v, f, 500, dt_ps=0.0005, mass_amu=wk.MASS_O, mass_h_amu=wk.MASS_H, temperature_k=300.0, species=species,
                rigid_water=True, r_oh=wk.TIP3P_R_OH, r_hh=wk.TIP3P_R_HH, first_step=chunk * 500)
ref = GamdForce(sdw, posw.shape[0], boxw, 4.2, bond=bonds, scaler=SHIPPED_SCALERS["tip4p"])
f32 = ref.forward(x, species=species, denormalize=True)
fb = engb.forward(x, species=species, denormalize=True)
err = float((f32 - fb).abs().max() / f32.abs().max())
print(f"water bf16: 3000 steps finite={bool(torch.isfinite(x).all())} bf16-vs-f32 force err on the final positions={err:.2e}")
assert torch.isfinite(x).all() and err < 1e-2


# ---- 4. split-fp16 determinism ---------------------------------------------------------------------------------------
runs = []
for rep in range(2):
    eng = GamdForce(sd, 10000, box, 10.2, scaler=SHIPPED_SCALERS["lj"], neighbor_skin=1.7, edge_dtype="f16x3")
    x = torch.from_numpy(pos).float().cuda()
    v = torch.from_numpy(wk.maxwell_boltzmann(10000, 300.0)).float().cuda()
    f = eng.forward(x, denormalize=True).clone()
    for chunk in range(4):
        eng.md_run(x, v, f, 500, temperature_k=300.0, first_step=chunk * 500)
    runs.append((sha(x), sha(v), sha(f)))
    if rep == 0:
        ref = GamdForce(sd, 10000, box, 10.2, scaler=SHIPPED_SCALERS["lj"])
        f32 = ref.forward(x, denormalize=True)
        err = float((f32 - eng.forward(x, denormalize=True)).abs().max() / f32.abs().max())
        ref.close()
    eng.close()
print(f"LJ f16x3: two 2000-step runs bit-identical: {runs[0] == runs[1]}; f16x3-vs-f32 force err on the final positions={err:.2e}")
assert runs[0] == runs[1] and err < 1e-5

# ---- 5. batched boxes --------------------------------------------------------------------------------------------------
nb = 38
snap = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lj258_seed0.npz"))["pos"]
rng = np.random.default_rng(3)
posb = np.concatenate([np.mod(snap + (rng.normal(0, 0.17, snap.shape) if b else 0.0), 27.27) for b in range(nb)])
runs = []
for rep in range(2):
    eng = GamdForce(sd, 258, 27.27, 7.5, scaler=SHIPPED_SCALERS["lj"], neighbor_skin=1.25, n_boxes=nb)
    x = torch.from_numpy(posb).float().cuda()
    v = torch.from_numpy(wk.maxwell_boltzmann(nb * 258, 300.0)).float().cuda()
    f = eng.forward(x, denormalize=True).clone()
    for chunk in range(4):
        eng.md_run(x, v, f, 500, temperature_k=300.0, first_step=chunk * 500, seed=9, remove_cm_motion=True)
    runs.append((sha(x), sha(v), sha(f)))
    vcom = v.view(nb, 258, 3).mean(dim=1).abs().max().item()
    eng.close()
print(f"LJ batch {nb} x 258: two 2000-step runs bit-identical: {runs[0] == runs[1]}; finite={bool(torch.isfinite(x).all())}; "
      f"largest per-box COM speed {vcom:.3e} A/ps")
assert runs[0] == runs[1] and torch.isfinite(x).all() and vcom < 0.5      # what the last two half-kicks add; the drift itself is removed every step

# ---- 6. generic-width split-fp16 (wide_lp.hip): the DFT-water widths on rigid water, run TWICE ---------------------------
cfgw = ModelConfig(kind="dynbox", encoding_size=256, hidden_dim=128, edge_embedding_dim=256, conv_layer=5)
sdd = make_state_dict(cfgw, 5, 3.1, 1.2)
posd, boxd, speciesd, _ = wk.water_box(258, seed=11, jitter=0.0, wrap=False)
runs = []
for rep in range(2):
    eng = GamdForce(sdd, posd.shape[0], boxd, 5.0, nbr_flavour="torch", cfg=cfgw, neighbor_skin=5.0 / 6, edge_dtype="f16x3", scaler=(0.0, 25.0))
    x = torch.from_numpy(posd).float().cuda()
    v = torch.zeros_like(x)
    f = eng.forward(x, species=speciesd, denormalize=True).clone()
    for chunk in range(4):
        eng.md_run(x, v, f, 250, dt_ps=0.0005, mass_amu=wk.MASS_O, mass_h_amu=wk.MASS_H, temperature_k=300.0, species=speciesd,
                   rigid_water=True, r_oh=wk.TIP3P_R_OH, r_hh=wk.TIP3P_R_HH, first_step=chunk * 250, seed=4)
    runs.append((sha(x), sha(v), sha(f)))
    if rep == 0:
        ref = GamdForce(sdd, posd.shape[0], boxd, 5.0, nbr_flavour="torch", cfg=cfgw, scaler=(0.0, 25.0))
        f32 = ref.forward(x, species=speciesd, denormalize=True)
        errw = float((f32 - eng.forward(x, species=speciesd, denormalize=True)).abs().max() / f32.abs().max())
        ref.close()
    eng.close()
print(f"DFT widths f16x3: two 1000-step rigid-water runs bit-identical: {runs[0] == runs[1]}; finite={bool(torch.isfinite(x).all())}; "
      f"f16x3-vs-f32 force err on the final positions={errw:.2e}")
assert runs[0] == runs[1] and torch.isfinite(x).all() and errw < 1e-5
