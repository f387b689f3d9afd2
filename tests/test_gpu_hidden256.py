"""hidden_dim above 128: the DT = 2 kernels of gamd_amd/csrc/wide_d.hip.

`hidden_dim` is the inner width of the reference's MLPs (nn_module.py:21-60, :95-106, :306-320) and may be anything there;
up to round 5 the library stopped at 128.  Pinned by two goldens generated from the reference itself (oracle/make_golden.py
--only-d256): 128 / 192 / 128 on the LJ model (192 is zero-padded to two 128-blocks) and 256 / 256 / 256 on the water model
with the bond feature; seeded odd widths go against the oracle, which test_oracle_golden.py pins on the same goldens."""
import numpy as np
import pytest
import torch

import gamd_oracle as orc
from gamd_amd import workloads
from gamd_amd._lib import GamdError
from gamd_amd.weights import ModelConfig, make_state_dict
from helpers import load_golden, rel_err, per_atom_err, edge_set
from test_gpu_round4 import _engine, _wide_case, _check_stages, TOL, P99_TOL

pytestmark = pytest.mark.gpu

D256 = [("lj258_d192_seed15", (128, 192, 128)), ("tip3p774_d256_w256_seed16", (256, 256, 256))]


@pytest.mark.parametrize("name,widths", D256)
def test_reference_goldens_stage_by_stage(name, widths):
    """edge features, edge embedding, every layer's node features, normalised and denormalised forces of the reference."""
    _check_stages(*_wide_case(name, widths=widths, keep_stages=True))


@pytest.mark.parametrize("name,widths", D256)
def test_skin_reuse_batches_and_enqueued_md_runs(name, widths):
    """Skin mode along a random walk against the exact rebuild; a batch of boxes against the boxes one by one (bitwise);
    an MD run enqueued in one call against the same steps one call at a time (bitwise, layer-0 table reuse included)."""
    g, cfg, sd, eng, box, rc, n, bond, species = _wide_case(name, widths=widths, skin_frac=1.0 / 6.0)
    scaler = (g["scaler_mean"], g["scaler_var"])
    exact = _engine(sd, n, box, rc, bond=bond, scaler=scaler)
    posw = np.mod(g["pos"], box)
    assert rel_err(eng.forward(torch.from_numpy(posw).float(), species=species).cpu().numpy(), g["out_norm"]) < TOL
    rng = np.random.default_rng(5)
    x = posw.copy()
    for step in range(10):
        x = x + rng.normal(0, 0.04, x.shape)
        a = eng.forward(torch.from_numpy(x).float(), species=species).cpu().numpy()
        b = exact.forward(torch.from_numpy(x).float(), species=species).cpu().numpy()
        assert np.array_equal(edge_set(eng.debug_edges()), edge_set(exact.debug_edges())), step
        assert rel_err(a, b) < TOL
    assert 1 <= eng.skin_stats()[0] < 10
    # batch of 3 boxes
    pos = [posw + (rng.normal(0, 0.1, (n, 3)) if b else 0.0) for b in range(3)]
    batch = _engine(sd, n, box, rc, bond=bond, scaler=scaler, n_boxes=3)
    sp3 = None if species is None else np.concatenate([species] * 3)
    out = batch.forward(torch.from_numpy(np.concatenate(pos)).float(), species=sp3).cpu().numpy()
    assert rel_err(out[:n], g["out_norm"]) < TOL
    for b in range(3):
        one = exact.forward(torch.from_numpy(pos[b]).float(), species=species).cpu().numpy()
        assert np.array_equal(out[b * n:(b + 1) * n], one), b
    batch.close(); exact.close()
    # md_run: 6 steps in one call == 6 calls of one step
    kw = (dict(mass_amu=workloads.MASS_O, mass_h_amu=workloads.MASS_H, species=species) if species is not None else dict(mass_amu=39.9))
    res = []
    for chunks in ((6,), (1,) * 6):
        e = _engine(sd, n, box, rc, bond=bond, scaler=scaler, neighbor_skin=rc / 6.0)
        xg = torch.from_numpy(posw).float().cuda()
        vg = torch.from_numpy(np.random.default_rng(1).normal(0, 1.0, (n, 3))).float().cuda()
        fg = e.forward(xg, species=species, denormalize=True).clone()
        for c in chunks:
            e.md_run(xg, vg, fg, c, dt_ps=0.001, temperature_k=0.0, gamma_per_ps=25.0, **kw)
        res.append((xg.cpu().numpy(), vg.cpu().numpy(), fg.cpu().numpy()))
        e.close()
    for a, b in zip(*res):
        assert np.isfinite(a).all() and np.array_equal(a, b)
    eng.close()


@pytest.mark.parametrize("enc,hid,emb,kind,n_rbf", [(128, 129, 128, "lj", 40), (96, 256, 200, "water", 40), (256, 200, 64, "dynbox", 0),
                                                    (200, 255, 256, "dynbox", 40), (128, 160, 128, "water", 40)])
def test_odd_widths_and_feature_sets_against_the_oracle(enc, hid, emb, kind, n_rbf):
    """Widths that are no multiple of 128 on either side of hidden_dim, with and without the RBF expansion and the bond
    feature, both neighbour flavours (the dynamic-box model searches like torch, <= and no self pairs); also the small-tile and the many-tile regime of the launches."""
    water = kind != "lj"
    cfg = ModelConfig(kind=kind, encoding_size=enc, hidden_dim=hid, edge_embedding_dim=emb, conv_layer=3, use_bond=kind == "water", n_rbf=n_rbf)
    sd = make_state_dict(cfg, 31, 2.9, 1.1)
    for nmol in ((60, 700) if water else (150, 3000)):
        if water:
            pos, box, species, bonds = workloads.water_box(nmol, seed=8)
            feat, rc = torch.from_numpy(species.astype(np.float32)).view(-1, 1), 4.2
            if kind == "dynbox":
                bonds = None
        else:
            pos, box = workloads.lj_box(nmol, seed=8)
            species = bonds = feat = None
            rc = 7.5
        n = pos.shape[0]
        flavour = "torch" if kind == "dynbox" else "jaxmd"
        eng = _engine(sd, n, box, rc, bond=bonds, nbr_flavour=flavour)
        p = torch.remainder(torch.from_numpy(pos).float(), float(box))
        out = eng.forward(p, species=species).cpu().numpy()
        edges = orc.neighbor_edges(p, box, rc, flavour)
        assert np.array_equal(edge_set(eng.debug_edges()), edge_set(edges.numpy()))
        if kind == "dynbox":
            ref = orc.forward_dynamic_box(sd, p, feat, np.full(3, box, dtype=np.float32), rc).numpy()
        else:
            ref = orc.forward(sd, p, edges, box, feat=feat, bond=bonds).numpy()
        assert rel_err(out, ref) < TOL, (nmol, rel_err(out, ref))
        med, p99, worst, cnt = per_atom_err(out, ref)
        assert p99 < P99_TOL, (med, p99, worst)
        eng.close()


@pytest.mark.parametrize("name,widths", D256)
def test_reference_edge_lists_appended_self_loops_and_the_model_classes(name, widths):
    """pnet_model([pos], [edge_idx]) of the drop-in classes with the reference's own edge list, a shuffled and an empty one; the
    other reading of add_self_loop (one zero-embedding loop appended per atom) against the oracle's restatement of it."""
    from gamd_amd.compat import ParticleNetLightningLJ, ParticleNetLightningWater
    g, cfg, sd = load_golden(name)
    assert (cfg.encoding_size, cfg.hidden_dim, cfg.edge_embedding_dim) == widths
    box, rc, n = float(g["box"]), float(g["cutoff"]), g["pos"].shape[0]
    posw = torch.from_numpy(np.mod(g["pos"], box)).float().cuda()
    edge_idx = torch.from_numpy(g["edge_idx"]).long().cuda()
    water = "node_feat" in g
    feat = torch.from_numpy(g["node_feat"]) if water else None
    bond = g["bond"] if "bond" in g else None
    species = (g["node_feat"].reshape(-1) != 0) if water else None
    if water:
        m = ParticleNetLightningWater(state_dict=sd)
        out = m.pnet_model([posw], feat.cuda(), [edge_idx])
    else:
        m = ParticleNetLightningLJ(state_dict=sd)
        out = m.pnet_model([posw], [edge_idx])
    assert rel_err(out.cpu().numpy(), g["out_norm"]) < TOL
    eng = m._get_engine()
    perm = torch.randperm(edge_idx.shape[1], generator=torch.Generator().manual_seed(0)).cuda()
    assert rel_err(eng.forward_edges(posw, edge_idx[:, perm], species=species).cpu().numpy(), g["out_norm"]) < TOL
    none = torch.zeros((2, 0), dtype=torch.long)
    ref0 = orc.forward(sd, posw.cpu(), none, box, feat=feat, bond=bond).numpy()
    assert rel_err(eng.forward_edges(posw, none, species=species).cpu().numpy(), ref0) < TOL
    mode = "append_zero_feature_loops"
    loops = _engine(sd, n, box, rc, bond=bond, self_loop_mode=mode)
    ref = orc.forward(sd, posw.cpu(), edge_idx.cpu(), box, feat=feat, bond=bond, self_loop_mode=mode).numpy()
    assert rel_err(ref, g["out_norm"]) > 1e-3                                   # the two readings do differ
    assert rel_err(loops.forward(posw, species=species).cpu().numpy(), ref) < TOL
    assert rel_err(loops.forward_edges(posw, edge_idx[:, perm], species=species).cpu().numpy(), ref) < TOL
    loops.close()


@pytest.mark.parametrize("width,hid,n_rbf", [(160, 200, 40), (256, 256, 0)])
def test_update_edge_models_with_a_wide_hidden_layer(width, hid, n_rbf):
    """WaterMDDynamicBoxNet(update_edge=True) (nn_module.py:91-92, :140-146): the layers after the first read the previous
    layer's LayerNorm(e_emb); with hidden_dim above 128 the conv kernel of wide_d.hip writes those rows."""
    cfg = ModelConfig(kind="dynbox", encoding_size=width, hidden_dim=hid, edge_embedding_dim=width, conv_layer=3, n_rbf=n_rbf, update_edge=True)
    sd = make_state_dict(cfg, 77, 2.9, 1.1)
    assert "graph_conv.conv.0.edge_layer_norm.weight" in sd
    pos, box, species, _ = workloads.water_box(110, seed=4)
    feat, rc, n = torch.from_numpy(species.astype(np.float32)).view(-1, 1), 4.2, pos.shape[0]
    p = torch.remainder(torch.from_numpy(pos).float(), float(box))
    ref = orc.forward_dynamic_box(sd, p, feat, np.full(3, box, dtype=np.float32), rc).numpy()
    for kw in (dict(), dict(neighbor_skin=rc / 6.0), dict(n_boxes=2)):
        eng = _engine(sd, n, box, rc, nbr_flavour="torch", **kw)
        nb = kw.get("n_boxes", 1)
        out = eng.forward(torch.cat([p] * nb), species=np.tile(species, nb)).cpu().numpy()
        for b in range(nb):
            assert rel_err(out[b * n:(b + 1) * n], ref) < TOL, (kw, b)
        eng.close()


def test_what_is_not_built_is_refused_loudly():
    cfg = ModelConfig(kind="lj", encoding_size=128, hidden_dim=256, edge_embedding_dim=128, conv_layer=2)
    sd = make_state_dict(cfg, 1, 5.0, 1.5)
    for dtype in ("bf16", "f16x3"):
        with pytest.raises(GamdError, match="hidden_dim above 128"):
            _engine(sd, 258, 27.27, 7.5, edge_dtype=dtype)
    with pytest.raises(ValueError, match="hidden_dim"):
        _engine(make_state_dict(ModelConfig(kind="lj", hidden_dim=300, conv_layer=2), 1, 5.0, 1.5), 258, 27.27, 7.5)
