"""Zero-edit drop-in (gamd_amd/dropin/): the reference drivers' exact call sequence — import by the reference's module name,
`SimpleNamespace` args as written in the drivers, a Lightning-shaped checkpoint file and scaler.npz — against outputs of
the reference's own modules (tests/golden/).  LJ/test_script/test_langevin.py:56-77,91,108;
water/test_script/test_nosehoover.py:63-89,106,123; water/test_script/test_nosehoover_hb.py:64-92,112,130."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

import gamd_oracle as orc
from helpers import load_golden, rel_err
from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS
from gamd_amd import workloads
from test_dropin import ARGS_LJ, ARGS_TIP, ARGS_DFT, DROPIN, _ckpt

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module", autouse=True)
def dropin_on_path():
    sys.path.insert(0, DROPIN)                      # what a user of the reference does instead of sys.path.append('../')
    yield
    sys.path.remove(DROPIN)


def _write(tmp_path, sd, mean, var):
    PATH, SCALER_CKPT = _ckpt(tmp_path, sd)
    np.savez(SCALER_CKPT, mean=np.asarray(mean), var=np.asarray(var))
    return PATH, SCALER_CKPT


def test_lj_langevin_driver_lines(tmp_path):
    g, cfg, sd = load_golden("lj258_seed0")
    PATH, SCALER_CKPT = _write(tmp_path, sd, g["scaler_mean"], g["scaler_var"])
    # --- LJ/test_script/test_langevin.py:56-77 ---
    from types import SimpleNamespace
    from train_network_lj import ParticleNetLightning
    args = SimpleNamespace(use_layer_norm=True,
                           encoding_size=128,
                           hidden_dim=128,
                           edge_embedding_dim=128,
                           drop_edge=False,
                           conv_layer=4,
                           rotate_aug=False,
                           update_edge=False,
                           use_part=False,
                           data_dir='',
                           loss='mae')
    model = ParticleNetLightning(args).load_from_checkpoint(PATH, args=args)
    model.load_training_stats(SCALER_CKPT)
    model.cuda()
    model.eval()
    # --- :91, :108 (pos: what OpenMM hands back, np f64 Angstrom) ---
    pos = g["pos"].astype(np.float64)
    force = model.predict_forces(pos)
    assert isinstance(force, np.ndarray) and force.dtype == np.float64 and force.shape == (258, 3)
    assert rel_err(force, g["forces"]) < TOL
    force2 = model.predict_forces(pos + 27.27)                       # any periodic image; a fresh array every call
    assert force2 is not force and rel_err(force2, g["forces"]) < TOL
    # the model-level objects of the reference module: pnet_model([pos], [edge_idx]) and forward(...) = its denormalised form
    posw = torch.from_numpy(np.mod(pos, 27.27)).float().cuda()
    edge = torch.from_numpy(g["edge_idx"]).long().cuda()
    out = model.pnet_model([posw], [edge])
    assert rel_err(out.cpu().numpy(), g["out_norm"]) < TOL
    assert rel_err(model.forward([posw], None, [edge]).cpu().numpy(), g["forces"]) < TOL


def test_tip3p_nosehoover_driver_lines(tmp_path):
    g, cfg, sd = load_golden("tip3p774_seed3")
    PATH, SCALER_CKPT = _write(tmp_path, sd, g["scaler_mean"], g["scaler_var"])
    # --- water/test_script/test_nosehoover.py:63-89 ---
    from types import SimpleNamespace
    from train_network_tip3p import ParticleNetLightning
    NUM_OF_ATOMS = g["pos"].shape[0]                  # (258*3)
    args = SimpleNamespace(use_layer_norm=True,
                           encoding_size=128,
                           hidden_dim=128,
                           edge_embedding_dim=128,
                           drop_edge=False,
                           rotate_aug=False,
                           data_dir='',
                           loss='mae')
    model = ParticleNetLightning(args).load_from_checkpoint(PATH, args=args)
    model.load_training_stats(SCALER_CKPT)
    model.cuda()
    model.eval()
    particle_type = []
    for i in range(NUM_OF_ATOMS):
        particle_type.append(1 if i % 3 == 0 else 0)   # O: 1, H: 0
    particle_type = np.array(particle_type).astype(np.int64).reshape(-1, 1)
    particle_type_one_hot = np.zeros((particle_type.size, 1), dtype=np.float32)
    particle_type_one_hot[particle_type.reshape(-1) == 1] = 1
    feat = torch.from_numpy(particle_type_one_hot).float().cuda()
    # --- :106, :123 ---
    force = model.predict_forces(feat, g["pos"].astype(np.float64))
    assert force.dtype == np.float64 and rel_err(force, g["forces"]) < TOL


def test_dft_nosehoover_hb_driver_lines(tmp_path):
    g, cfg, sd = load_golden("dynbox384_dftcfg_seed5")
    mean, var = SHIPPED_SCALERS["dft"]
    PATH, SCALER_CKPT = _write(tmp_path, sd, mean, var)
    # --- water/test_script/test_nosehoover_hb.py:64-92 ---
    from types import SimpleNamespace
    from train_network_real_large import ParticleNetLightning
    NUM_OF_ATOMS = g["pos"].shape[0]
    args = SimpleNamespace(use_layer_norm=True,
                           encoding_size=256,
                           hidden_dim=128,
                           edge_embedding_dim=256,
                           conv_layer=5,
                           drop_edge=False,
                           cutoff=float(g["cutoff"]),            # 9.5 in the driver; the golden's box is smaller
                           rotate_aug=False,
                           update_edge=False,
                           use_part=False,
                           expand_edge=True,
                           data_dir='',
                           loss='mse')
    model = ParticleNetLightning(args).load_from_checkpoint(PATH, args=args)
    model.load_training_stats(SCALER_CKPT)
    model.cuda()
    model.eval()
    feat = torch.zeros((NUM_OF_ATOMS, 1))
    feat[::3] = 1.0
    feat = feat.float().cuda()
    box_size = g["box"]
    # --- :112, :130: the atom count and the box arrive with the call ---
    force = model.predict_forces(feat, g["pos"].astype(np.float64), box_size)
    ref = g["out_norm"].astype(np.float64) * np.sqrt(var) + mean
    assert force.dtype == np.float64 and rel_err(force, ref) < TOL
    assert model.num_atoms == NUM_OF_ATOMS
    # another system size through the same wrapper (md_module.get_neighbor takes whatever it is handed)
    sub, fsub = g["pos"][:300].astype(np.float64), feat[:300]
    f300 = model.predict_forces(fsub, sub, box_size)
    o300 = orc.forward_dynamic_box(sd, torch.from_numpy(np.mod(sub, box_size)).float(), fsub.cpu(), box_size, float(g["cutoff"])).numpy()
    assert rel_err(f300, o300.astype(np.float64) * np.sqrt(var) + mean) < TOL and model.num_atoms == 300


def test_tip4p_module_runs_the_251_molecule_system(tmp_path):
    """water/train_network_tip4p.py: NUM_OF_ATOMS = 251 * 3.  No reference-generated golden of that size exists; checked
    against the oracle (itself pinned by the goldens) on a 251-molecule box."""
    from train_network_tip4p import ParticleNetLightning, NUM_OF_ATOMS, BOX_SIZE, CUTOFF_RADIUS, create_water_bond
    sd = make_state_dict(ModelConfig(kind="water", use_bond=True), 3, 2.9, 1.1)
    mean, var = SHIPPED_SCALERS["tip4p"]
    PATH, SCALER_CKPT = _write(tmp_path, sd, mean, var)
    model = ParticleNetLightning(ARGS_TIP).load_from_checkpoint(PATH, args=ARGS_TIP)
    model.load_training_stats(SCALER_CKPT)
    model.cuda()
    model.eval()
    pos, box, species, bonds = workloads.water_box(251, mol_per_20A3=251.0, seed=77)
    assert abs(box - BOX_SIZE) < 1e-9 and pos.shape[0] == NUM_OF_ATOMS and np.array_equal(bonds, create_water_bond(NUM_OF_ATOMS))
    feat = torch.from_numpy(species.astype(np.float32)).view(-1, 1).cuda()
    force = model.predict_forces(feat, pos)
    pw = torch.from_numpy(np.mod(pos, BOX_SIZE)).float()
    edges = orc.neighbor_edges(pw, BOX_SIZE, CUTOFF_RADIUS, "jaxmd")
    ref = orc.forward(sd, pw, edges, BOX_SIZE, feat=feat.cpu(), bond=bonds).numpy().astype(np.float64) * np.sqrt(var) + mean
    assert rel_err(force, ref) < TOL
