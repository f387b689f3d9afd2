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


class _Payload:                                     # stands for arbitrary code a pickle can run on load
    def __reduce__(self):
        return (os.getenv, ("GAMD_PICKLE_CANARY",))


def test_checkpoints_are_read_with_the_restricted_unpickler(tmp_path):
    """weights_only=True first: a Lightning checkpoint of the reference (tensors + an argparse Namespace of hyper
    parameters) loads; one that needs the full unpickler is refused unless the caller opts in with allow_pickle=True."""
    import argparse
    from types import SimpleNamespace
    sd = make_state_dict(ModelConfig(), 2)
    ck = {"state_dict": {"pnet_model." + k: v for k, v in sd.items()}, "epoch": 30, "global_step": 12345,
          "hparams_name": "args", "hyper_parameters": {"args": argparse.Namespace(encoding_size=128, lr=3e-4, loss="mae")},
          "extra": SimpleNamespace(a=1)}
    torch.save(ck, tmp_path / "lightning.ckpt")
    assert set(load_checkpoint(str(tmp_path / "lightning.ckpt"))) == set(sd)
    bad = dict(ck, callbacks=_Payload())
    torch.save(bad, tmp_path / "needs_pickle.ckpt")
    with pytest.raises(RuntimeError, match="allow_pickle=True"):
        load_checkpoint(str(tmp_path / "needs_pickle.ckpt"))
    got = load_checkpoint(str(tmp_path / "needs_pickle.ckpt"), allow_pickle=True)
    assert torch.equal(got["node_emb"], sd["node_emb"])


def test_batched_model_call_checks_its_list_lengths():
    """compat._ModelLevel._batched: the per-graph list must match pos_lst (checked before any GPU work)."""
    from gamd_amd.compat import ParticleNetLightningLJ
    m = ParticleNetLightningLJ(state_dict=make_state_dict(ModelConfig(), 0), num_atoms=8)
    pos = [torch.zeros(8, 3), torch.zeros(8, 3)]
    e = torch.zeros(2, 4, dtype=torch.long)
    with pytest.raises(ValueError, match="edge_lst has 3 entries for 2 graphs"):
        m.pnet_model(pos, [e, e, e])
    with pytest.raises(ValueError, match="edge_lst has 1 entries for 2 graphs"):
        m.pnet_model(pos, [e])
    with pytest.raises(TypeError):
        m.pnet_model(pos, 1, 2, 3, 4)


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


def test_wide_and_unexpanded_specs_roundtrip():
    """DFT-water widths (256/256/128 x 5) and expand_edge=False: spec, seeded weights and infer_config agree."""
    from gamd_amd.weights import ModelConfig, make_state_dict, infer_config, state_dict_spec, validate_state_dict
    for cfg in (ModelConfig(kind="dynbox", encoding_size=256, hidden_dim=128, edge_embedding_dim=256, conv_layer=5),
                ModelConfig(kind="dynbox", n_rbf=0), ModelConfig(kind="water", use_bond=True, n_rbf=0)):
        sd = make_state_dict(cfg, 4)
        spec = state_dict_spec(cfg)
        assert list(sd) == list(spec)
        assert ("edge_expand.centers" in sd) == (cfg.n_rbf > 0)
        assert sd["edge_encoder.mlp_layer.0.weight"].shape[1] == cfg.edge_in == 3 + 1 + cfg.n_rbf + int(cfg.use_bond)
        got = infer_config(sd)
        assert (got.encoding_size, got.hidden_dim, got.edge_embedding_dim, got.conv_layer, got.n_rbf, got.use_bond) == \
               (cfg.encoding_size, cfg.hidden_dim, cfg.edge_embedding_dim, cfg.conv_layer, cfg.n_rbf, cfg.use_bond)
        validate_state_dict(sd, cfg)
    # edge_affine's inner width is MLP's default 128 whatever hidden_dim is (nn_module.py:25,95)
    wide = state_dict_spec(ModelConfig(kind="dynbox", encoding_size=256, hidden_dim=128, edge_embedding_dim=256))
    assert wide["graph_conv.conv.0.edge_affine.mlp_layer.0.weight"] == (128, 256)
    assert wide["graph_conv.conv.0.theta_edge.mlp_layer.3.weight"] == (256, 128)


def test_batchnorm_checkpoints_spec_oracle_and_loader(tmp_path):
    """use_layer_norm=False (the constructors' default, nn_module.py:171-196): BatchNorm1d buffers in the spec in torch's
    registration order, recognised by infer_config, kept through the checkpoint loader (int64 step counter included), and
    the oracle's eval-mode branch is the running-statistics affine map."""
    from oracle import gamd_oracle as orc
    cfg = ModelConfig(kind="lj", use_layer_norm=False, conv_layer=2)
    sd = make_state_dict(cfg, 5)
    keys = [k for k in sd if k.startswith("graph_conv.norm_layers.0.")]
    assert [k.rsplit(".", 1)[1] for k in keys] == ["weight", "bias", "running_mean", "running_var", "num_batches_tracked"]
    assert sd["graph_conv.norm_layers.0.num_batches_tracked"].dtype == torch.int64
    assert not infer_config(sd).use_layer_norm and infer_config(make_state_dict(ModelConfig(), 5)).use_layer_norm
    validate_state_dict(sd, cfg)
    bn = torch.nn.BatchNorm1d(cfg.encoding_size)
    bn.load_state_dict({k.rsplit(".", 1)[1]: sd[k] for k in keys}, strict=True)      # the keys torch itself expects
    bn.eval()
    x = torch.randn(7, cfg.encoding_size)
    assert torch.equal(orc.node_norm(sd, "graph_conv.norm_layers.0", x), bn(x))
    torch.save({"state_dict": {"pnet_model." + k: v for k, v in sd.items()}}, tmp_path / "bn.ckpt")
    got = load_checkpoint(str(tmp_path / "bn.ckpt"))
    assert list(got) == list(sd) and got["graph_conv.norm_layers.1.num_batches_tracked"].dtype == torch.int64
    assert torch.equal(got["graph_conv.norm_layers.1.running_var"], sd["graph_conv.norm_layers.1.running_var"])


def test_update_edge_spec_roundtrip():
    """update_edge=True: every conv layer registers its edge_layer_norm FIRST (nn_module.py:91-92); infer_config sees it."""
    cfg = ModelConfig(kind="dynbox", update_edge=True, encoding_size=256, hidden_dim=128, edge_embedding_dim=256, conv_layer=2)
    sd = make_state_dict(cfg, 2)
    layer0 = [k for k in sd if k.startswith("graph_conv.conv.0.")]
    assert layer0[:3] == ["graph_conv.conv.0.edge_layer_norm.weight", "graph_conv.conv.0.edge_layer_norm.bias",
                          "graph_conv.conv.0.edge_affine.mlp_layer.0.weight"]
    assert sd["graph_conv.conv.1.edge_layer_norm.weight"].shape == (256,)
    got = infer_config(sd)
    assert got.update_edge and not infer_config(make_state_dict(ModelConfig(kind="dynbox"), 2)).update_edge
    validate_state_dict(sd, cfg)


def test_compat_wrappers_are_lazy_and_mirror_the_reference_signatures():
    """Constructing the Lightning-shaped wrappers needs no GPU; the engine is created on first use."""
    import inspect
    from types import SimpleNamespace
    from gamd_amd.compat import ParticleNetLightningLJ, ParticleNetLightningWater, ParticleNetLightningDFT
    lj = ParticleNetLightningLJ(SimpleNamespace())
    assert list(inspect.signature(lj.predict_forces).parameters) == ["pos", "verbose"]          # train_network_lj.py:133
    w = ParticleNetLightningWater(SimpleNamespace())
    assert list(inspect.signature(w.predict_forces).parameters) == ["feat", "pos"]              # train_network_tip3p.py:142
    d = ParticleNetLightningDFT(SimpleNamespace(cutoff=9.5))
    assert list(inspect.signature(d.predict_forces).parameters) == ["feat", "pos", "box_size"]  # train_network_real_large.py:148
    assert d.cutoff == 9.5 and d._nbr_flavour == "torch" and d._skin == 0.0
    assert lj._skin == lj.cutoff / 6.0                                                          # graph_utils.py:24
    assert lj.cuda() is lj and lj.eval() is lj
    with pytest.raises(RuntimeError, match="no weights"):
        lj._get_engine()


def _bench(*flags, env=None):
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "GAMD_BENCH_SHARE_GPU")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *flags], capture_output=True, text=True,
                          timeout=300, cwd=root, env=e)


def test_bench_refuses_more_ranks_than_devices():
    """`bench.py --gpus N` must never degrade silently to fewer ranks: without N visible devices it exits non-zero
    before anything touches a GPU (round-1 verdict: it used to print n_gpus = 1)."""
    if torch.cuda.device_count() >= 3:
        pytest.skip("three devices present")
    p = _bench("--gpus", "3", "--steps", "1", "--warmup", "0")
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert "--gpus 3" in p.stderr and "HIP device" in p.stderr


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    p = _bench("--gpus", "3", "--steps", "1", "--warmup", "0", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and p.stdout.strip() == "" and "WORLD_SIZE=2" in p.stderr
    p = _bench("--gpus", "1", "--steps", "1", "--warmup", "0", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_pmc_record_is_tied_to_kernel_sources():
    """profiles/pmc_conv_edge.json carries the hash of the kernel sources it was measured on; bench.py prints
    traffic = null (plus a note) when they no longer match."""
    import json, importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    rec = json.load(open(os.path.join(root, "profiles", "pmc_conv_edge.json")))
    assert "kernel_source_sha256_16" in rec and len(rec["kernel_source_sha256_16"]) == 16


def test_bench_result_line_stays_under_the_drivers_capture():
    """Round 5's bench line was 21 KB; the driver keeps an 8 081-character tail of stdout and parsed nothing.  bench.compact_line
    turns the full record into the contract line: fed with that very record (profiles/r05_final_bench_default.json), once as it
    is and once blown up to eight ranks, the line stays under bench.LINE_LIMIT and keeps the contract's key order."""
    import json, importlib.util, types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(root, "profiles", "r05_final_bench_default.json")))
    assert len(json.dumps(full)) > 20000
    args = types.SimpleNamespace(detail=os.path.join(root, "bench_detail.json"))
    for ranks in (1, 8):
        rec = json.loads(json.dumps(full))
        rec["ensemble"]["per_rank"] = [dict(rec["ensemble"]["per_rank"][0], rank=r) for r in range(ranks)]
        rec["ensemble"]["boxes"] = rec["n_gpus"] = rec["config"]["boxes"] = ranks
        line = bench.compact_line(rec, args)
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
        assert len(text) < bench.LINE_LIMIT <= 6000, len(text)
        assert list(line)[:14] == ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                   "vs_baseline", "dtype", "data", "config", "roofline"]
        assert list(line["roofline"])[:7] == ["kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"]
        assert list(line["cpu_baseline"])[:5] == ["value", "unit", "cores", "kind", "sample"]
        assert abs(line["value"] / full["value"] - 1) < 1e-5 and abs(line["roofline"]["frac"] / full["roofline"]["frac"] - 1) < 1e-5
        assert set(line["secondary"]) == set(bench.SECONDARY_COMPACT) and len(line["ensemble"]["rank_seconds"]) == ranks
        print(ranks, len(text))


def test_self_loop_mode_names_agree_between_oracle_and_engine():
    import gamd_oracle as orc
    from gamd_amd.engine import SELF_LOOP
    assert tuple(SELF_LOOP) == orc.SELF_LOOP_MODES and SELF_LOOP["dgl07_noop"] == 0


def test_gelu_tail_polynomial_in_the_kernel_header_is_accurate():
    """gamd_common.h evaluates GELU(x) = max(x,0) - |x| 2^Q(|x|) with a degree-6 fit Q of log2 Phi(-a) on [0, 6].  Re-evaluate it
    in fp32 with the coefficients as written in the header and compare with the exact erf-GELU of nn.GELU()
    (nn_module.py:41-42): within half an ulp of the result plus 1.2e-7 everywhere on [-10, 10]."""
    import re
    from scipy.special import ndtr
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "gamd_amd", "csrc", "gamd_common.h")).read()
    q = [np.float32(re.search(rf"#define GAMD_GELU_Q{i} (\S+)f", src).group(1)) for i in range(7)]
    f = np.float32
    x = np.linspace(-10, 10, 2000001).astype(f)
    a = np.minimum(np.abs(x), f(6.0))
    p = np.full_like(a, q[6])
    for c in q[5::-1]:
        p = (p * a + c).astype(f)
    out = (np.maximum(x, f(0)) - (a * np.exp2(p).astype(f)).astype(f)).astype(f)
    exact = x.astype(np.float64) * ndtr(x.astype(np.float64))
    err = np.abs(out.astype(np.float64) - exact)
    ulp = np.spacing(np.abs(exact).astype(f)).astype(np.float64)
    assert (err / (0.5 * ulp + 1.2e-7)).max() < 1.6 and err.max() < 7e-7
    assert err[np.abs(x) < 2.5].max() < 3.5e-7


def test_box_argument_of_a_three_box_batch_is_not_ambiguous():
    """A 1-D box of three numbers is ONE orthorhombic box for every box of the batch — also when the batch has exactly three
    boxes (round-4 advisor: it used to be read as three cubic boxes, silently wrong min-image forces); per-box cubic edges
    are spelled [n_boxes, 1], per-box orthorhombic boxes [n_boxes, 3]."""
    from gamd_amd.engine import _boxes, _box3
    one = np.array([20.0, 30.0, 40.0], dtype=np.float32)
    for nb in (1, 2, 3, 5):
        assert np.array_equal(_boxes(_box3(one), nb), np.tile(one, (nb, 1)))
        assert np.array_equal(_boxes(27.27, nb), np.full((nb, 3), np.float32(27.27)))
    assert np.array_equal(_boxes(np.array([[20.0], [30.0], [40.0]]), 3), np.repeat(one[:, None], 3, axis=1))
    assert np.array_equal(_boxes([20.0, 30.0], 2), np.array([[20.0] * 3, [30.0] * 3], dtype=np.float32))
    per_box = np.arange(9, dtype=np.float32).reshape(3, 3) + 10
    assert np.array_equal(_boxes(per_box, 3), per_box)
    with pytest.raises(ValueError, match="box must be"):
        _boxes(np.ones((2, 3)), 3)
    with pytest.raises(ValueError, match="box must be"):
        _boxes([1.0, 2.0, 3.0, 4.0], 3)
