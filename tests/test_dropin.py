"""gamd_amd/dropin/: modules named like the reference's train_network_*.py, importable with only that directory on
sys.path, carrying the reference's module constants and constructor signature (CPU part: no GPU is touched before
predict_forces)."""
import importlib
import inspect
import os
import re
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from gamd_amd.weights import ModelConfig, make_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "gamd_amd", "dropin")
REF = "/root/reference/code"
MODULES = {"train_network_lj": "LJ", "train_network_tip3p": "water", "train_network_tip4p": "water",
           "train_network_real_large": "water"}
# the drivers' args, verbatim: LJ/test_script/test_langevin.py:63-73, water/test_script/test_nosehoover.py:69-76,
# water/test_script/test_nosehoover_hb.py:69-81
ARGS_LJ = SimpleNamespace(use_layer_norm=True, encoding_size=128, hidden_dim=128, edge_embedding_dim=128, drop_edge=False,
                          conv_layer=4, rotate_aug=False, update_edge=False, use_part=False, data_dir='', loss='mae')
ARGS_TIP = SimpleNamespace(use_layer_norm=True, encoding_size=128, hidden_dim=128, edge_embedding_dim=128, drop_edge=False,
                           rotate_aug=False, data_dir='', loss='mae')
ARGS_DFT = SimpleNamespace(use_layer_norm=True, encoding_size=256, hidden_dim=128, edge_embedding_dim=256, conv_layer=5,
                           drop_edge=False, cutoff=9.5, rotate_aug=False, update_edge=False, use_part=False, expand_edge=True,
                           data_dir='', loss='mse')


@pytest.fixture(scope="module")
def dropin():
    sys.path.insert(0, DROPIN)
    mods = {m: importlib.import_module(m) for m in MODULES}
    yield mods
    sys.path.remove(DROPIN)


def test_module_constants_are_the_references(dropin):
    lj, t3, t4, dft = (dropin[m] for m in MODULES)
    assert (lj.NUM_OF_ATOMS, lj.BOX_SIZE, lj.CUTOFF_RADIUS) == (258, 27.27, 7.5)
    assert (t3.NUM_OF_ATOMS, t3.BOX_SIZE, t3.CUTOFF_RADIUS) == (258 * 3, 20.0, 4.2)
    assert (t4.NUM_OF_ATOMS, t4.BOX_SIZE, t4.CUTOFF_RADIUS) == (251 * 3, 20.0, 4.2)
    assert not hasattr(dft, "NUM_OF_ATOMS") and not hasattr(dft, "BOX_SIZE")       # commented out in the reference
    for m in (t3, t4, dft):
        assert m.create_water_bond(9).tolist() == [[0, 1], [0, 2], [3, 4], [3, 5], [6, 7], [6, 8]]
    for m in dropin.values():
        assert callable(m.build_model) and inspect.isclass(m.ParticleNetLightning)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is not present on this machine")
def test_constants_and_signatures_against_the_reference_sources(dropin):
    """Read (never import: jax / dgl / lightning are absent) the reference modules' top-level assignments and the
    ParticleNetLightning / predict_forces signatures and compare."""
    for name, sub in MODULES.items():
        src = open(os.path.join(REF, sub, name + ".py")).read()
        mod = dropin[name]
        env = {}
        for const in ("CUTOFF_RADIUS", "left_bound", "right_bound", "BOX_SIZE", "NUM_OF_ATOMS", "LAMBDA1", "LAMBDA2"):
            m = re.search(rf"^{const}\s*=\s*(.+?)\s*(#.*)?$", src, flags=re.M)
            if m:
                env[const] = eval(m.group(1), {}, dict(env))
                assert getattr(mod, const) == env[const], (name, const)
            else:
                assert not hasattr(mod, const), (name, const)
        init = re.search(r"class ParticleNetLightning\(pl\.LightningModule\):\s*def __init__\(self, (.*?)\):", src, flags=re.S).group(1)
        want = [p.split("=")[0].strip() for p in init.replace("\n", " ").split(",")]
        got = [p for p in inspect.signature(mod.ParticleNetLightning.__init__).parameters if p not in ("self", "engine_kw")]
        assert got == want, (name, got, want)
        pf = re.search(r"def predict_forces\(self, (.*?)\):", src).group(1)
        want_pf = [p.split(":")[0].split("=")[0].strip() for p in pf.split(",")]
        assert list(inspect.signature(mod.ParticleNetLightning.predict_forces).parameters)[1:] == want_pf, name


def _ckpt(tmp_path, sd, lightning=True):
    p = tmp_path / "checkpoint.ckpt"
    torch.save({"state_dict": {"pnet_model." + k: v for k, v in sd.items()}, "epoch": 1} if lightning else dict(sd), p)
    np.savez(tmp_path / "scaler.npz", mean=np.array([0.25]), var=np.array([9.0]))
    return str(p), str(tmp_path / "scaler.npz")


def test_driver_sequence_up_to_the_first_gpu_call(dropin, tmp_path):
    """`ParticleNetLightning(args).load_from_checkpoint(PATH, args=args)` returns a NEW wrapper (Lightning's classmethod
    semantics), sized by the module constants; load_training_stats / cuda / eval chain; no GPU is needed until
    predict_forces, which fails loudly without one."""
    PL = dropin["train_network_lj"].ParticleNetLightning
    sd = make_state_dict(ModelConfig(kind="lj"), 0)
    PATH, SCALER_CKPT = _ckpt(tmp_path, sd)
    first = PL(ARGS_LJ)
    model = first.load_from_checkpoint(PATH, args=ARGS_LJ)
    assert isinstance(model, PL) and model is not first and first._sd is None
    assert (model.num_atoms, model.box_size, model.cutoff) == (258, 27.27, 7.5) and model.loss_fn == "mae"
    assert set(model.state_dict()) == set(sd)
    model.load_training_stats(SCALER_CKPT)
    assert model.cuda() is model and model.eval() is model
    assert float(model.training_var[0]) == 9.0 and model.training_mean.dtype == np.float64
    also = PL.load_from_checkpoint(PATH, args=ARGS_LJ)                 # the classmethod form
    assert isinstance(also, PL) and torch.equal(also.state_dict()["node_emb"], sd["node_emb"])
    if not torch.cuda.is_available():
        with pytest.raises(Exception, match="no CPU fallback|HIP"):
            model.predict_forces(np.zeros((258, 3)))
    with pytest.raises(NotImplementedError, match="force-inference"):
        model.training_step(None, 0)


def test_module_constants_are_read_when_a_wrapper_is_built(dropin, tmp_path, monkeypatch):
    """The reference is re-targeted by editing NUM_OF_ATOMS / BOX_SIZE / CUTOFF_RADIUS in the module
    (water/train_network_tip3p.py:31-32 keeps the TIP4P values as comments): same here."""
    t3 = dropin["train_network_tip3p"]
    monkeypatch.setattr(t3, "NUM_OF_ATOMS", 251 * 3)
    monkeypatch.setattr(t3, "CUTOFF_RADIUS", 3.4)
    m = t3.ParticleNetLightning(ARGS_TIP)
    assert (m.num_atoms, m.cutoff) == (753, 3.4) and m.bond.shape == (502, 2)
    m4 = dropin["train_network_tip4p"].ParticleNetLightning(ARGS_TIP)
    assert (m4.num_atoms, m4.box_size, m4.cutoff) == (753, 20.0, 4.2) and m4.bond.shape == (502, 2)
    d = dropin["train_network_real_large"].ParticleNetLightning(ARGS_DFT)
    assert d.cutoff == 9.5 and d.num_atoms is None and d.use_part is False      # the count arrives with the first call


def test_a_checkpoint_of_other_widths_than_args_is_refused(dropin, tmp_path):
    """build_model(args) + strict load_state_dict in the reference: size mismatch.  conv_layer is 4 in the LJ / TIP
    wrappers whatever args says (LJ/train_network_lj.py:75), args.conv_layer in the DFT one (:80)."""
    wide = make_state_dict(ModelConfig(kind="lj", encoding_size=256, hidden_dim=128, edge_embedding_dim=256), 1)
    PATH, _ = _ckpt(tmp_path, wide)
    PL = dropin["train_network_lj"].ParticleNetLightning
    with pytest.raises(RuntimeError, match="size mismatch.*encoding_size"):
        PL(ARGS_LJ).load_from_checkpoint(PATH, args=ARGS_LJ)
    two = make_state_dict(ModelConfig(kind="lj", conv_layer=2), 1)
    PATH2, _ = _ckpt(tmp_path, two, lightning=False)
    with pytest.raises(RuntimeError, match="conv_layer"):
        PL(ARGS_LJ, model_weights_ckpt=PATH2)
    dsd = make_state_dict(ModelConfig(kind="dynbox", encoding_size=256, hidden_dim=128, edge_embedding_dim=256, conv_layer=5), 2)
    PATH3, _ = _ckpt(tmp_path, dsd)
    D = dropin["train_network_real_large"].ParticleNetLightning
    assert D(ARGS_DFT).load_from_checkpoint(PATH3, args=ARGS_DFT).state_dict().keys() == dsd.keys()
    with pytest.raises(RuntimeError, match="conv_layer"):
        D(ARGS_DFT).load_from_checkpoint(PATH3, args=SimpleNamespace(**{**vars(ARGS_DFT), "conv_layer": 4}))


def test_only_the_dropin_directory_on_pythonpath_is_enough(tmp_path):
    """What INTEGRATION.md section 1 tells a maintainer to do: PYTHONPATH=<gamd_amd/dropin> in front of the reference's own
    sys.path.append('../') — from another working directory, without the repository root on the path (the drop-in modules find
    the package themselves)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = DROPIN
    code = ("import sys; sys.path.append('../'); sys.path.append('../../')\n"      # the drivers' own lines (test_langevin.py:24-25)
            "from types import SimpleNamespace\n"
            "from train_network_lj import ParticleNetLightning, NUM_OF_ATOMS, BOX_SIZE, CUTOFF_RADIUS\n"
            "from train_network_tip3p import create_water_bond\n"
            "import train_network_tip4p, train_network_real_large\n"
            "m = ParticleNetLightning(SimpleNamespace(use_layer_norm=True, encoding_size=128, hidden_dim=128, edge_embedding_dim=128,\n"
            "    drop_edge=False, conv_layer=4, rotate_aug=False, update_edge=False, use_part=False, data_dir='', loss='mae'))\n"
            "print('OK', NUM_OF_ATOMS, BOX_SIZE, CUTOFF_RADIUS, m.num_atoms, type(m).__module__)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "OK 258 27.27 7.5 258 train_network_lj" in r.stdout
