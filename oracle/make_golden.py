#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own
modules (`/root/reference/code/nn_module.py`, `md_module.py`) on CPU in the
build container, through the stub dgl/jax modules of `oracle/ref_stubs.py`.

TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists (never on the
GPU box).  The reference source is imported in place, never copied.  What is
committed is data: inputs (positions that are not reproducible from a seed),
seeds, and the reference's outputs.

Weights are not stored: every case loads `gamd_amd.weights.make_state_dict(cfg,
seed)` into the reference module with `load_state_dict(strict=True)` — which
also proves our key names / shapes equal the reference's — so tests rebuild the
identical weights from the seed.

Usage:  python oracle/make_golden.py                     writes tests/golden/*.npz
        python oracle/make_golden.py --only-wide         (or --only-bn / --only-update / --only-batch / --only-selfloop)
        python oracle/make_golden.py --out DIR           writes DIR/*.npz instead
        python oracle/make_golden.py --check             regenerates every fixture into a temporary directory and compares it
                                                         with tests/golden/ array by array, BIT FOR BIT (keys, dtypes, shapes,
                                                         bytes); exit status 1 and a list of differences when anything drifted.
        python oracle/make_golden.py --check --float-rtol 2e-6
                                                         the same, but floating-point arrays may differ by that much relative to
                                                         their largest element (torch's CPU GEMM blocking depends on the thread
                                                         count of the host); keys, dtypes, shapes, integer arrays stay exact.
                                                         tests/test_oracle_golden.py runs this wherever /root/reference exists.
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_stubs  # noqa: E402
from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS  # noqa: E402
import gamd_oracle as orc  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def water_bond(n):
    """create_water_bond, water/train_network_tip3p.py:38-42 (restated)."""
    return np.array([[i, i + k] for i in range(0, n, 3) for k in (1, 2)])


def jaxmd_edge_set(pos32: torch.Tensor, box: float, cutoff: float) -> torch.Tensor:
    """Edge set of the jax-md path, restated from the call-site arguments
    (graph_utils.py:21-25,51-61): no executable reference exists here, so the
    same restatement as the oracle is used and only the *model* outputs on that
    edge set come from the reference module."""
    return orc.neighbor_edges(pos32, box, cutoff, "jaxmd")


def margin_to_cutoff(pos32, box, cutoff):
    boxt = orc._box_tensor(box)
    d = orc._min_image(pos32[:, None, :] - pos32[None, :, :], boxt).norm(dim=-1)
    return float((d - cutoff).abs().min())


def run_fixed_box(nn_module, name, cfg, seed, pos, box, cutoff, scaler, feat=None,
                  bond=None, lmean=4.0, lstd=1.5, keep_h=True, edge_stride=1, inplace_self_loop=False, h_stride=1):
    torch.manual_seed(1234)
    ref_stubs.INPLACE_SELF_LOOP = bool(inplace_self_loop)
    sd = make_state_dict(cfg, seed, lmean, lstd)
    if cfg.kind == "lj":
        m = nn_module.SimpleMDNetNew(encoding_size=cfg.encoding_size, out_feats=3, box_size=box,
                                     hidden_dim=cfg.hidden_dim, conv_layer=cfg.conv_layer,
                                     edge_embedding_dim=cfg.edge_embedding_dim, drop_edge=False,
                                     use_layer_norm=cfg.use_layer_norm)
    else:
        m = nn_module.WaterMDNetNew(in_feats=1, encoding_size=cfg.encoding_size, out_feats=3,
                                    box_size=box, bond=torch.as_tensor(bond) if bond is not None else None,
                                    hidden_dim=cfg.hidden_dim, conv_layer=cfg.conv_layer,
                                    edge_embedding_dim=cfg.edge_embedding_dim, drop_edge=False,
                                    use_layer_norm=cfg.use_layer_norm)
    m.load_state_dict(sd, strict=True)
    m.eval()
    # predict_forces front end (LJ/train_network_lj.py:135-142), restated
    pos64 = np.asarray(pos, dtype=np.float64)
    pos32 = torch.from_numpy(pos64).float()
    edge_idx = jaxmd_edge_set(torch.remainder(pos32, orc._box_tensor(box)), box, cutoff)
    posw = torch.from_numpy(np.mod(pos64, np.array(box))).float()
    margin = margin_to_cutoff(posw, box, cutoff)
    # fp32 distance error is a few ulp (~2e-6 at 10 A); 2e-5 keeps edge membership unambiguous
    assert margin > 2e-5, f"{name}: a pair sits {margin} from the cutoff"

    # capture per-stage tensors from the real module with forward hooks
    cap = {}
    m.edge_layer_norm.register_forward_hook(lambda mod, i, o: cap.__setitem__("e", o.detach().clone()))
    m.edge_encoder.register_forward_hook(lambda mod, i, o: cap.__setitem__("feat", i[0].detach().clone()))
    # residual stream h at the start of every layer (input of norm_layers[l]; the block
    # calls conv.forward() directly so conv hooks never fire) and after the last one
    hs = []
    for norm in m.graph_conv.norm_layers:
        norm.register_forward_hook(lambda mod, i, o: hs.append(i[0].detach().clone()))
    m.graph_decoder.register_forward_hook(lambda mod, i, o: hs.append(i[0].detach().clone()))
    with torch.no_grad():
        if cfg.kind == "lj":
            out = m([posw], [edge_idx])
        else:
            out = m([posw], feat, [edge_idx])
    ref_stubs.INPLACE_SELF_LOOP = False
    out = out.numpy()
    mean, var = scaler
    forces = out * np.sqrt(var) + mean          # denormalize, train_network_lj.py:128-131
    rec = dict(pos=pos64, box=np.float64(box), cutoff=np.float64(cutoff), seed=np.int64(seed),
               length_mean=np.float64(lmean), length_std=np.float64(lstd),
               edge_idx=edge_idx.numpy().astype(np.int32),
               edge_stride=np.int64(edge_stride),
               feat_rows=cap["feat"].numpy()[::edge_stride],
               e_rows=cap["e"].numpy()[::edge_stride],
               out_norm=out, forces=forces, scaler_mean=mean, scaler_var=var,
               margin=np.float64(margin),
               cfg=np.array([cfg.kind, str(cfg.encoding_size), str(cfg.hidden_dim),
                             str(cfg.edge_embedding_dim), str(cfg.conv_layer), str(int(cfg.use_bond))]))
    if h_stride > 1:
        rec["h_stride"] = np.int64(h_stride)
    if not cfg.use_layer_norm:
        rec["use_layer_norm"] = np.int64(0)
    if inplace_self_loop:
        rec["self_loop_inplace"] = np.int64(1)
    if keep_h:
        # h_0 .. h_L — enough to localise a diff to one layer
        rec["h_layers"] = np.stack([h.numpy()[::h_stride] for h in hs])
    if feat is not None:
        rec["node_feat"] = feat.numpy()
    if bond is not None:
        rec["bond"] = np.asarray(bond, dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: N={pos64.shape[0]} E={edge_idx.shape[1]} margin={margin:.2e} "
          f"|F|max={np.abs(forces).max():.4g}")


def run_dynbox(nn_module, md_module, name, cfg, seed, pos, box, cutoff, lmean, lstd, inplace_self_loop=False):
    ref_stubs.INPLACE_SELF_LOOP = bool(inplace_self_loop)
    sd = make_state_dict(cfg, seed, lmean, lstd)
    m = nn_module.WaterMDDynamicBoxNet(in_feats=1, encoding_size=cfg.encoding_size, out_feats=3,
                                       bond=None, hidden_dim=cfg.hidden_dim, conv_layer=cfg.conv_layer,
                                       edge_embedding_dim=cfg.edge_embedding_dim, drop_edge=False,
                                       use_layer_norm=True, update_edge=cfg.update_edge, expand_edge=cfg.n_rbf > 0)
    m.load_state_dict(sd, strict=True)
    m.eval()
    pos32 = torch.from_numpy(np.asarray(pos)).float()
    n = pos32.shape[0]
    feat = torch.zeros(n, 1)
    feat[::3] = 1.0
    boxa = np.asarray(box, dtype=np.float32)
    # the reference's own O(N^2) search (md_module.py:93-126), executed as is
    edge_idx, dist, dist_norm, _ = md_module.get_neighbor(pos32, cutoff, torch.from_numpy(boxa))
    with torch.no_grad():
        out = m([pos32], feat, [boxa], cutoff).numpy()
    ref_stubs.INPLACE_SELF_LOOP = False
    np.savez_compressed(os.path.join(OUT, name + ".npz"), self_loop_inplace=np.int64(1 if inplace_self_loop else 0),
                        pos=pos32.numpy(), box=boxa, cutoff=np.float64(cutoff), seed=np.int64(seed),
                        length_mean=np.float64(lmean), length_std=np.float64(lstd),
                        edge_idx=edge_idx.numpy().astype(np.int32), dist_norm=dist_norm.numpy(),
                        node_feat=feat.numpy(), out_norm=out, **({"update_edge": np.int64(1)} if cfg.update_edge else {}),
                        cfg=np.array([cfg.kind, str(cfg.encoding_size), str(cfg.hidden_dim),
                                      str(cfg.edge_embedding_dim), str(cfg.conv_layer), "0", str(cfg.n_rbf)]))
    print(f"{name}: N={n} E={edge_idx.shape[1]}")


def run_batched_lj(nn_module, name, cfg, seed, pos_list, box, cutoff, lmean, lstd):
    """The model-level call with SEVERAL graphs, SimpleMDNetNew.forward(pos_lst, edge_idx_lst) with len(pos_lst) > 1:
    build_graph_batches + dgl.batch (nn_module.py:655-661, :676-679), output [sum N, 3] in list order.  Edge indices are
    local to each graph.  Pins gamd_amd.compat._ModelLevel._batched."""
    torch.manual_seed(1234)
    sd = make_state_dict(cfg, seed, lmean, lstd)
    m = nn_module.SimpleMDNetNew(encoding_size=cfg.encoding_size, out_feats=3, box_size=box, hidden_dim=cfg.hidden_dim,
                                 conv_layer=cfg.conv_layer, edge_embedding_dim=cfg.edge_embedding_dim, drop_edge=False,
                                 use_layer_norm=True)
    m.load_state_dict(sd, strict=True)
    m.eval()
    pos_w, edges = [], []
    for pos in pos_list:
        pos64 = np.asarray(pos, dtype=np.float64)
        posw = torch.from_numpy(np.mod(pos64, np.array(box))).float()
        e = jaxmd_edge_set(posw, box, cutoff)
        assert margin_to_cutoff(posw, box, cutoff) > 2e-5
        pos_w.append(posw)
        edges.append(e)
    with torch.no_grad():
        out = m(pos_w, edges).numpy()
        singles = [m([p], [e]).numpy() for p, e in zip(pos_w, edges)]
    # dgl.batch of independent graphs == the graphs one by one (no batch statistics in the shipped configuration)
    assert np.abs(out - np.concatenate(singles)).max() <= 2e-6 * np.abs(out).max()
    rec = dict(box=np.float64(box), cutoff=np.float64(cutoff), seed=np.int64(seed), length_mean=np.float64(lmean),
               length_std=np.float64(lstd), out_norm=out, n_graphs=np.int64(len(pos_list)),
               cfg=np.array([cfg.kind, str(cfg.encoding_size), str(cfg.hidden_dim), str(cfg.edge_embedding_dim),
                             str(cfg.conv_layer), str(int(cfg.use_bond))]))
    for i, (p, e) in enumerate(zip(pos_w, edges)):
        rec[f"pos{i}"] = p.numpy()
        rec[f"edge_idx{i}"] = e.numpy().astype(np.int32)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: graphs={len(pos_list)} N={[p.shape[0] for p in pos_w]} E={[e.shape[1] for e in edges]} |out|max={np.abs(out).max():.4g}")


def compare_dirs(new_dir, old_dir, float_rtol=0.0):
    """Differences between the fixtures of two directories, as a list of strings (empty = identical bit for bit, or — with
    float_rtol > 0 — floating-point arrays within that relative distance and everything else identical)."""
    diffs = []
    new, old = (sorted(f for f in os.listdir(d) if f.endswith(".npz")) for d in (new_dir, old_dir))
    for f in sorted(set(new) ^ set(old)):
        diffs.append(f"{f}: only in {'the regenerated set' if f in new else old_dir}")
    for f in sorted(set(new) & set(old)):
        a, b = np.load(os.path.join(new_dir, f), allow_pickle=False), np.load(os.path.join(old_dir, f), allow_pickle=False)
        for k in sorted(set(a.files) ^ set(b.files)):
            diffs.append(f"{f}[{k}]: only in {'the regenerated file' if k in a.files else 'the committed file'}")
        for k in sorted(set(a.files) & set(b.files)):
            x, y = a[k], b[k]
            if x.dtype != y.dtype or x.shape != y.shape:
                diffs.append(f"{f}[{k}]: {x.dtype}{x.shape} regenerated vs {y.dtype}{y.shape} committed")
            elif x.tobytes() != y.tobytes():
                worst = ""
                if x.dtype.kind == "f":
                    rel = float(np.abs(x.astype(np.float64) - y.astype(np.float64)).max() / max(float(np.abs(y).max()), 1e-30))
                    if rel <= float_rtol:
                        continue
                    worst = f", worst relative difference {rel:.3g}"
                diffs.append(f"{f}[{k}]: contents differ{worst}")
    return diffs


def main():
    global OUT
    argv = list(sys.argv[1:])
    check = "--check" in argv
    if check:
        argv.remove("--check")
        rtol = float(argv[argv.index("--float-rtol") + 1]) if "--float-rtol" in argv else 0.0
        committed, tmp = OUT, tempfile.TemporaryDirectory(prefix="gamd_golden_")
        OUT = tmp.name
        generate([])
        diffs = compare_dirs(OUT, committed, rtol)
        exact = rtol == 0.0 or not compare_dirs(OUT, committed)
        tmp.cleanup()
        if diffs:
            print(f"make_golden --check: {len(diffs)} difference(s) between the reference's outputs here and tests/golden/:")
            for d in diffs:
                print("  " + d)
            sys.exit(1)
        print("make_golden --check: every fixture under tests/golden/ regenerates "
              + ("bit for bit" if exact else f"within {rtol:g} (floats), exactly otherwise (not bit for bit on this host)")
              + " from the reference")
        return
    if "--out" in argv:
        i = argv.index("--out")
        OUT = os.path.abspath(argv[i + 1])
        del argv[i:i + 2]
    generate(argv)


def generate(argv):
    """argv: at most one --only-* selector."""
    everything = len(argv) == 0
    os.makedirs(OUT, exist_ok=True)
    nn_module, md_module = ref_stubs.import_reference(REF)
    lj_pos = np.load(os.path.join(REF, "code/LJ/init_pos.npy"))          # [258,3] f32, in [0, 27.22]
    w_pos = np.load(os.path.join(REF, "code/water/init_pos.npy"))        # [774,3] f64, centred
    full = dict(encoding_size=128, hidden_dim=128, edge_embedding_dim=128, conv_layer=4)

    # The trainers' DEFAULT widths (LJ/train_network_lj.py:394-396, water/train_network_tip3p.py:404-406: encoding_size 256,
    # hidden_dim 128, edge_embedding_dim 256) on the fixed-box models: jax-md neighbour flavour (self edges), bond feature,
    # 4 layers — the configuration the generic-width kernels (wide.hip) serve outside the dynamic-box flavour.
    # `--only-wide` writes just these two.
    wide = dict(encoding_size=256, hidden_dim=128, edge_embedding_dim=256, conv_layer=4)
    if "--only-wide" in argv or everything:
        run_fixed_box(nn_module, "lj258_w256_seed9", ModelConfig(kind="lj", **wide), 9, lj_pos, 27.27, 7.5,
                      SHIPPED_SCALERS["lj"], lmean=5.3, lstd=1.6, edge_stride=61, h_stride=3)
        nw = w_pos.shape[0]
        featw = torch.zeros(nw, 1)
        featw[::3] = 1.0
        run_fixed_box(nn_module, "tip3p774_w256_seed10", ModelConfig(kind="water", use_bond=True, **wide), 10,
                      w_pos, 20.0, 4.2, SHIPPED_SCALERS["tip3p"], feat=featw, bond=water_bond(nw),
                      lmean=2.9, lstd=1.1, edge_stride=211, h_stride=9)
    if "--only-wide" in argv:
        return

    # hidden_dim above 128 (the MLPs' inner width, nn_module.py:24-51 — any value is legal in the reference): 192 (not a
    # multiple of the 128-wide block the kernels pad to) on LJ and 256 with the trainers' other default widths on water.
    # `--only-d256` writes just these two.
    if "--only-d256" in argv or everything:
        run_fixed_box(nn_module, "lj258_d192_seed15",
                      ModelConfig(kind="lj", encoding_size=128, hidden_dim=192, edge_embedding_dim=128, conv_layer=4), 15,
                      lj_pos, 27.27, 7.5, SHIPPED_SCALERS["lj"], lmean=5.3, lstd=1.6, edge_stride=61, h_stride=3)
        nd_ = w_pos.shape[0]
        featd = torch.zeros(nd_, 1)
        featd[::3] = 1.0
        run_fixed_box(nn_module, "tip3p774_d256_w256_seed16",
                      ModelConfig(kind="water", use_bond=True, encoding_size=256, hidden_dim=256, edge_embedding_dim=256,
                                  conv_layer=4), 16,
                      w_pos, 20.0, 4.2, SHIPPED_SCALERS["tip3p"], feat=featd, bond=water_bond(nd_),
                      lmean=2.9, lstd=1.1, edge_stride=211, h_stride=9)
    if "--only-d256" in argv:
        return

    # use_layer_norm=False: the constructors' and the trainers' DEFAULT (--use_layer_norm is a store_true flag,
    # LJ/train_network_lj.py:398): nn.BatchNorm1d between the conv layers (nn_module.py:171-196, :579), in eval mode with
    # non-trivial running statistics.  `--only-bn` writes just these two.
    if "--only-bn" in argv or everything:
        run_fixed_box(nn_module, "lj258_bn_seed11", ModelConfig(kind="lj", use_layer_norm=False, **full), 11, lj_pos, 27.27, 7.5,
                      SHIPPED_SCALERS["lj"], lmean=5.3, lstd=1.6, edge_stride=211, h_stride=3)
        nb_ = w_pos.shape[0]
        featb = torch.zeros(nb_, 1)
        featb[::3] = 1.0
        run_fixed_box(nn_module, "tip3p774_bn_w256_seed12",
                      ModelConfig(kind="water", use_bond=True, use_layer_norm=False, encoding_size=256, hidden_dim=128,
                                  edge_embedding_dim=256, conv_layer=4), 12,
                      w_pos, 20.0, 4.2, SHIPPED_SCALERS["tip3p"], feat=featb, bond=water_bond(nb_),
                      lmean=2.9, lstd=1.1, edge_stride=211, h_stride=9)
    if "--only-bn" in argv:
        return

    # update_edge=True (--update_edge, water/train_network_real_large.py:83,362 -> SmoothConvLayerNew.update_edge_emb,
    # nn_module.py:91-92, :140-146): every conv layer hands LayerNorm(e_emb) to the layers after it as their edge embedding.
    # The DFT-water widths and a 128-wide 3-layer model.  `--only-update` writes just these two.
    if "--only-update" in argv or everything:
        subu = np.mod(w_pos[:384], 20.0)
        run_dynbox(nn_module, md_module, "dynbox384_update_dftcfg_seed13",
                   ModelConfig(kind="dynbox", encoding_size=256, hidden_dim=128, edge_embedding_dim=256, conv_layer=5,
                               update_edge=True), 13, subu, [20.0, 21.0, 22.5], 4.6, 3.1, 1.2)
        run_dynbox(nn_module, md_module, "dynbox384_update_seed14",
                   ModelConfig(kind="dynbox", update_edge=True, encoding_size=128, hidden_dim=128, edge_embedding_dim=128,
                               conv_layer=3), 14, subu, [20.0, 21.0, 22.5], 4.6, 3.1, 1.2)
    if "--only-update" in argv:
        return

    # model-level call with two graphs (`--only-batch` writes just this one)
    rngb = np.random.default_rng(11)
    run_batched_lj(nn_module, "lj258_batch2_seed0", ModelConfig(kind="lj", **full), 0,
                   [lj_pos, lj_pos.astype(np.float64) + rngb.normal(0, 0.3, lj_pos.shape)], 27.27, 7.5, 5.3, 1.6)
    if "--only-batch" in argv:
        return

    # The other reading of `fluid_graph.add_self_loop()` (result discarded, nn_module.py:650-652, :364): an IN-PLACE
    # add_self_loop as in DGL < 0.5.  The reference module is executed with a stub graph that mutates itself; these
    # cases pin the build's self_loop_mode = 1 ("append_zero_feature_loops").  `--only-selfloop` writes just these.
    run_fixed_box(nn_module, "lj258_selfloop_inplace_seed0", ModelConfig(kind="lj", **full), 0, lj_pos, 27.27, 7.5,
                  SHIPPED_SCALERS["lj"], lmean=5.3, lstd=1.6, edge_stride=211, inplace_self_loop=True)
    run_dynbox(nn_module, md_module, "dynbox384_selfloop_inplace_seed4", ModelConfig(kind="dynbox", **full), 4,
               np.mod(w_pos[:384], 20.0), [20.0, 21.0, 22.5], 4.6, 3.1, 1.2, inplace_self_loop=True)
    if "--only-selfloop" in argv:
        return

    # C1: the reference's own LJ snapshot, shipped LJ scaler, full-size model
    run_fixed_box(nn_module, "lj258_seed0", ModelConfig(kind="lj", **full), 0, lj_pos, 27.27, 7.5,
                  SHIPPED_SCALERS["lj"], lmean=5.3, lstd=1.6, edge_stride=29)
    # perturbed, partly un-wrapped copy: exercises the np.mod / jnp.mod wraps
    rng = np.random.default_rng(7)
    pert = lj_pos.astype(np.float64) + rng.normal(0, 0.35, lj_pos.shape) + np.array([27.27, -27.27, 0.0])
    run_fixed_box(nn_module, "lj258_pert_seed1", ModelConfig(kind="lj", **full), 1, pert, 27.27, 7.5,
                  SHIPPED_SCALERS["lj"], lmean=5.3, lstd=1.6, keep_h=False, edge_stride=101)
    # reduced-width toy (fast unit tests of the oracle's genericity)
    toy = rng.uniform(0, 12.0, (64, 3))
    run_fixed_box(nn_module, "lj64_h32", ModelConfig(kind="lj", encoding_size=32, hidden_dim=32,
                                                     edge_embedding_dim=32, conv_layer=2),
                  2, toy, 12.0, 3.9, (np.array([0.0]), np.array([1.0])), lmean=2.5, lstd=0.8)
    # TIP3P snapshot: bonds + species feature (water/train_network_tip3p.py, test_nosehoover.py:82-89)
    n = w_pos.shape[0]
    feat = torch.zeros(n, 1)
    feat[::3] = 1.0
    run_fixed_box(nn_module, "tip3p774_seed3", ModelConfig(kind="water", use_bond=True, **full), 3,
                  w_pos, 20.0, 4.2, SHIPPED_SCALERS["tip3p"], feat=feat, bond=water_bond(n),
                  lmean=2.9, lstd=1.1, edge_stride=53)
    # dynamic-box flavour (md_module.get_neighbor: <=, no self; orthorhombic box)
    sub = np.mod(w_pos[:384], 20.0)
    run_dynbox(nn_module, md_module, "dynbox384_seed4", ModelConfig(kind="dynbox", **full), 4,
               sub, [20.0, 21.0, 22.5], 4.6, 3.1, 1.2)
    run_dynbox(nn_module, md_module, "dynbox384_dftcfg_seed5",
               ModelConfig(kind="dynbox", encoding_size=256, hidden_dim=128, edge_embedding_dim=256,
                           conv_layer=5), 5, sub, [20.0, 21.0, 22.5], 4.6, 3.1, 1.2)
    # expand_edge=False (--disable_expand_edge, water/train_network_real_large.py:363) and mixed widths
    run_dynbox(nn_module, md_module, "dynbox384_noexpand_seed6", ModelConfig(kind="dynbox", n_rbf=0, **full), 6,
               sub, [20.0, 21.0, 22.5], 4.6, 3.1, 1.2)
    run_dynbox(nn_module, md_module, "dynbox384_h256_e128_seed7",
               ModelConfig(kind="dynbox", encoding_size=256, hidden_dim=128, edge_embedding_dim=128, conv_layer=3),
               7, sub, [20.0, 21.0, 22.5], 4.6, 3.1, 1.2)
    run_dynbox(nn_module, md_module, "dynbox384_h128_e256_noexpand_seed8",
               ModelConfig(kind="dynbox", encoding_size=128, hidden_dim=128, edge_embedding_dim=256, conv_layer=3,
                           n_rbf=0), 8, sub, [20.0, 21.0, 22.5], 4.6, 3.1, 1.2)


if __name__ == "__main__":
    main()
