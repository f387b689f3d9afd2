"""Host-side logic that needs no GPU: weights contract, loaders, workloads, loud failure."""
import os

import numpy as np
import pytest
import torch

from gamd_amd.weights import (ModelConfig, state_dict_spec, make_state_dict, infer_config, validate_state_dict,
                              load_checkpoint, load_scaler, SHIPPED_SCALERS)
from gamd_amd import workloads


def test_spec_and_infer_roundtrip():
    for cfg in (ModelConfig(kind="lj"), ModelConfig(kind="water", use_bond=True), ModelConfig(kind="water")):
        sd = make_state_dict(cfg, 3)
        validate_state_dict(sd, cfg)
        got = infer_config(sd)
        assert (got.kind, got.use_bond, got.conv_layer, got.encoding_size) == (cfg.kind, cfg.use_bond, 4, 128)


def test_make_state_dict_is_deterministic_and_nontrivial():
    a, b = make_state_dict(ModelConfig(), 5), make_state_dict(ModelConfig(), 5)
    assert all(torch.equal(a[k], b[k]) for k in a)
    c = make_state_dict(ModelConfig(), 6)
    assert not torch.equal(a["node_emb"], c["node_emb"])
    assert not torch.allclose(a["edge_layer_norm.weight"], torch.ones(128))
    assert a["edge_expand.centers"].shape == (40,) and abs(float(a["edge_expand.centers"][1]) - 1 / 39) < 1e-7


def test_validate_reports_mismatch():
    sd = make_state_dict(ModelConfig(), 0)
    sd.pop("node_emb")
    with pytest.raises(KeyError, match="node_emb"):
        validate_state_dict(sd, ModelConfig())


def test_lightning_checkpoint_and_scaler(tmp_path):
    sd = make_state_dict(ModelConfig(), 1)
    ck = {"state_dict": {"pnet_model." + k: v for k, v in sd.items()}, "epoch": 3}
    p = tmp_path / "checkpoint.ckpt"
    torch.save(ck, p)
    got = load_checkpoint(str(p))
    assert set(got) == set(sd) and torch.equal(got["node_emb"], sd["node_emb"])
    torch.save(dict(sd), tmp_path / "plain.ckpt")
    assert set(load_checkpoint(str(tmp_path / "plain.ckpt"))) == set(sd)
    np.savez(tmp_path / "scaler.npz", mean=np.array([0.5]), var=np.array([4.0]))
    m, v = load_scaler(str(tmp_path / "scaler.npz"))
    assert m.dtype == np.float64 and float(v[0]) == 4.0
    assert abs(SHIPPED_SCALERS["lj"][1][0] - 1010.00278026) < 1e-6


def test_workloads_shapes_and_density():
    pos, box = workloads.lj_box(10000)
    assert pos.shape == (10000, 3) and abs(box - 92.29) < 0.01
    assert pos.min() >= 0 and pos.max() < box
    pos2, _ = workloads.lj_box(10000)
    assert np.array_equal(pos, pos2)
    assert not np.array_equal(pos, workloads.lj_box(10000, seed=1235)[0])
    w, wb, sp, bonds = workloads.water_box(1390)
    assert w.shape == (4170, 3) and abs(wb - 35.06) < 0.02 and sp[:3].tolist() == [1, 0, 0]
    assert bonds.shape == (2780, 2) and bonds[1].tolist() == [0, 2]
    d = np.linalg.norm(w[1] - w[0] - wb * np.round((w[1] - w[0]) / wb))
    assert abs(d - 0.9572) < 0.1
    v = workloads.maxwell_boltzmann(20000)
    assert abs(v.std() - 10 * np.sqrt(workloads.KB * 100 / 39.9)) < 0.02


def test_engine_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gamd_amd.engine import GamdForce
    from gamd_amd._lib import GamdError
    with pytest.raises(GamdError, match="no CPU fallback"):
        GamdForce(make_state_dict(ModelConfig(), 0), 64, 12.0, 3.0)


def test_product_package_never_imports_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gamd_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "gamd_oracle" not in txt and "ref_stubs" not in txt, f


def test_compat_water_bond_table():
    from gamd_amd.compat import create_water_bond
    b = create_water_bond(9)
    assert b.tolist() == [[0, 1], [0, 2], [3, 4], [3, 5], [6, 7], [6, 8]]
