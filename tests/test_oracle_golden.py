"""Pin the CPU oracle (oracle/gamd_oracle.py) against outputs of the reference's
own modules (tests/golden/*.npz, produced by oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import gamd_oracle as orc
from helpers import load_golden, rel_err, edge_set

# the *_w256_* cases: the trainers' default widths 256 / 128 / 256 (LJ/train_network_lj.py:394-396) on the fixed-box models
# the *_bn_* cases: use_layer_norm=False, the constructors' / trainers' default (BatchNorm1d between the conv layers, eval mode)
FIXED = ["lj258_seed0", "lj258_pert_seed1", "lj64_h32", "tip3p774_seed3", "lj258_w256_seed9", "tip3p774_w256_seed10",
         "lj258_bn_seed11", "tip3p774_bn_w256_seed12", "lj258_d192_seed15", "tip3p774_d256_w256_seed16"]
# the *_d192_* / *_d256_* cases: hidden_dim above 128 (128 / 192 / 128 on LJ, 256 / 256 / 256 on water)


@pytest.mark.parametrize("name", FIXED)
def test_forward_matches_reference(name):
    g, cfg, sd = load_golden(name)
    box, rc = float(g["box"]), float(g["cutoff"])
    posw = torch.from_numpy(np.mod(g["pos"], box)).float()
    edge_idx = torch.from_numpy(g["edge_idx"]).long()
    feat = torch.from_numpy(g["node_feat"]) if "node_feat" in g else None
    bond = g["bond"] if "bond" in g else None
    st = {}
    out = orc.forward(sd, posw, edge_idx, box, feat=feat, bond=bond, stages=st).numpy()
    s = int(g["edge_stride"])
    # same ops in the same order -> (near) bit-equal; thresholds leave room for BLAS blocking
    assert rel_err(st["feat"].numpy()[::s], g["feat_rows"]) < 1e-6
    assert rel_err(st["e"].numpy()[::s], g["e_rows"]) < 5e-6
    if "h_layers" in g:
        hs = int(g["h_stride"]) if "h_stride" in g else 1
        for l, h in enumerate(st["h"]):
            assert rel_err(h.numpy()[::hs], g["h_layers"][l]) < 5e-6, f"layer {l}"
    assert rel_err(out, g["out_norm"]) < 5e-6
    forces = orc.denormalize(out, g["scaler_var"], g["scaler_mean"])
    assert forces.dtype == np.float64
    assert rel_err(forces, g["forces"]) < 5e-6


@pytest.mark.parametrize("name", FIXED)
def test_predict_forces_front_end(name):
    """np.mod / f32 cast / neighbour search / denormalise (train_network_lj.py:133-157)."""
    g, cfg, sd = load_golden(name)
    feat = torch.from_numpy(g["node_feat"]) if "node_feat" in g else None
    bond = g["bond"] if "bond" in g else None
    f = orc.predict_forces(sd, g["pos"], float(g["box"]), float(g["cutoff"]),
                           var=g["scaler_var"], mean=g["scaler_mean"], feat=feat, bond=bond)
    assert f.shape == g["forces"].shape and f.dtype == np.float64
    assert rel_err(f, g["forces"]) < 5e-6


def test_neighbor_semantics_jaxmd():
    """self pair kept, strict '<' (graph_utils.py:25,59)."""
    pos = torch.tensor([[0.0, 0, 0], [3.0, 0, 0], [9.5, 0, 0], [5.0, 5.0, 5.0]])
    e = orc.neighbor_edges(pos, 10.0, 3.0, "jaxmd").numpy()
    pairs = set(map(tuple, e.T))
    assert {(i, i) for i in range(4)} <= pairs          # self edges
    assert (0, 1) not in pairs                          # |r| == rc is excluded by '<'
    assert (0, 2) in pairs and (2, 0) in pairs          # periodic image, distance 0.5
    assert len(pairs) == 6


@pytest.mark.parametrize("name", ["dynbox384_seed4", "dynbox384_dftcfg_seed5", "dynbox384_noexpand_seed6",
                                  "dynbox384_h256_e128_seed7", "dynbox384_h128_e256_noexpand_seed8",
                                  "dynbox384_update_dftcfg_seed13", "dynbox384_update_seed14"])      # update_edge=True
def test_dynamic_box_matches_reference(name):
    """md_module.get_neighbor executed as-is (<=, no self, per-axis box) and
    WaterMDDynamicBoxNet.forward."""
    g, cfg, sd = load_golden(name)
    pos = torch.from_numpy(g["pos"])
    st = {}
    out = orc.forward_dynamic_box(sd, pos, torch.from_numpy(g["node_feat"]), g["box"],
                                  float(g["cutoff"]), stages=st).numpy()
    assert np.array_equal(st["edge_idx"].numpy(), g["edge_idx"].astype(np.int64))   # same order too
    assert not np.any(g["edge_idx"][0] == g["edge_idx"][1])
    assert rel_err(out, g["out_norm"]) < 5e-6


def test_state_dict_spec_counts():
    from gamd_amd.weights import ModelConfig, state_dict_spec
    n = sum(int(np.prod(s)) for s in state_dict_spec(ModelConfig(kind="lj")).values())
    assert n == 651565          # SURVEY.md §8a parameter inventory (incl. buffers-as-params)


# ---- self_loop_mode: the one reference semantic that cannot be executed here (DGL absent) ----
def _selfloop_lj():
    g, cfg, sd = load_golden("lj258_selfloop_inplace_seed0")
    box = float(g["box"])
    posw = torch.from_numpy(np.mod(g["pos"], box)).float()
    return g, sd, posw, torch.from_numpy(g["edge_idx"]).long(), box


def test_self_loop_mode_append_matches_reference_with_inplace_add_self_loop():
    """mode "append_zero_feature_loops" = the reference module run on a stub graph whose add_self_loop() mutates
    the receiver (DGL < 0.5 semantics; oracle/ref_stubs.py INPLACE_SELF_LOOP)."""
    g, sd, posw, edge_idx, box = _selfloop_lj()
    assert int(g["self_loop_inplace"]) == 1
    st = {}
    out = orc.forward(sd, posw, edge_idx, box, stages=st, self_loop_mode="append_zero_feature_loops").numpy()
    for l, h in enumerate(st["h"]):
        assert rel_err(h.numpy(), g["h_layers"][l]) < 5e-6, f"layer {l}"
    assert rel_err(out, g["out_norm"]) < 5e-6


def test_self_loop_mode_default_is_the_dgl07_noop_and_differs():
    """Same inputs and weights as lj258_seed0: the default mode reproduces THAT golden (functional add_self_loop,
    result discarded), and the two readings really give different forces (so the switch is observable)."""
    g1, sd, posw, edge_idx, box = _selfloop_lj()
    g0, _, sd0 = load_golden("lj258_seed0")
    assert np.array_equal(g0["edge_idx"], g1["edge_idx"]) and np.array_equal(g0["pos"], g1["pos"])
    out = orc.forward(sd, posw, edge_idx, box).numpy()
    assert rel_err(out, g0["out_norm"]) < 5e-6
    assert rel_err(g1["out_norm"], g0["out_norm"]) > 1e-3
    with pytest.raises(ValueError):
        orc.forward(sd, posw, edge_idx, box, self_loop_mode="bogus")


def test_self_loop_mode_append_dynamic_box():
    g, cfg, sd = load_golden("dynbox384_selfloop_inplace_seed4")
    g0 = load_golden("dynbox384_seed4")[0]
    pos = torch.from_numpy(g["pos"])
    out = orc.forward_dynamic_box(sd, pos, torch.from_numpy(g["node_feat"]), g["box"], float(g["cutoff"]),
                                  self_loop_mode="append_zero_feature_loops").numpy()
    assert rel_err(out, g["out_norm"]) < 5e-6
    assert rel_err(g["out_norm"], g0["out_norm"]) > 1e-3


def test_batched_model_call_golden_is_the_graphs_one_by_one():
    """The reference's model-level call with two graphs (build_graph_batches + dgl.batch, nn_module.py:655-661,676-679):
    the golden output equals the oracle applied to each graph separately, concatenated in list order."""
    g = dict(np.load(__import__("os").path.join(__import__("helpers").GOLDEN, "lj258_batch2_seed0.npz"), allow_pickle=False))
    from gamd_amd.weights import ModelConfig, make_state_dict
    kind, H, D, Eh, L, bond = [str(x) for x in g["cfg"]][:6]
    cfg = ModelConfig(kind=kind, encoding_size=int(H), hidden_dim=int(D), edge_embedding_dim=int(Eh), conv_layer=int(L))
    sd = make_state_dict(cfg, int(g["seed"]), float(g["length_mean"]), float(g["length_std"]))
    outs = []
    for i in range(int(g["n_graphs"])):
        outs.append(orc.forward(sd, torch.from_numpy(g[f"pos{i}"]), torch.from_numpy(g[f"edge_idx{i}"]).long(), float(g["box"])).numpy())
    assert np.concatenate(outs).shape == g["out_norm"].shape
    assert rel_err(np.concatenate(outs), g["out_norm"]) < 5e-6


def test_committed_fixtures_are_what_the_generator_writes_today():
    """Fixtures vs generator drift (round-4 review): wherever the reference is present, oracle/make_golden.py --check runs
    the reference's own modules again into a temporary directory and compares every array of every fixture with the
    committed one — same files, same keys, dtypes and shapes, integer arrays identical, floating-point arrays within 2e-6 of
    their largest element (bit for bit on the build container; torch's CPU GEMM blocking may differ with the host's thread
    count).  Skipped on the GPU box, where /root/reference does not exist."""
    import os
    import subprocess
    import sys
    if not os.path.isdir("/root/reference/code"):
        pytest.skip("the reference is not present on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "oracle", "make_golden.py"), "--check", "--float-rtol", "2e-6"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "regenerates" in r.stdout
