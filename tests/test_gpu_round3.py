"""GPU tests added in round 3: the padding slots of the last 32-edge tile (the throughput conv kernel and the bf16 kernel
let them gather an all-zero row of hn / S / D instead of masking every element), checked over edge counts with different
remainders mod 32 against the latency kernel (which keeps the masks) bit for bit and against the CPU oracle; repeated
calls with different edge counts on ONE engine (the zero row must stay zero); the fused multiply-add message accumulate
stays within the parity tolerance on the water model with bonds.  All through the C ABI."""
import numpy as np
import pytest
import torch

import gamd_oracle as orc
from helpers import load_golden, rel_err
from gamd_amd.weights import ModelConfig, make_state_dict
from gamd_amd import workloads

pytestmark = pytest.mark.gpu
TOL = 1e-5
MAIN_ONLY, SMALL_ONLY = dict(small_tile_limit=-1), dict(small_tile_limit=1000000)


def _engine(*a, **kw):
    from gamd_amd.engine import GamdForce
    return GamdForce(*a, **kw)


def test_padding_slots_of_the_last_tile_over_many_remainders():
    """E mod 32 takes many values over these boxes (asserted below): throughput kernel (zero-row padding) == latency
    kernel (masked padding) bit for bit, and both match the oracle on the GPU's own edge list."""
    sd = make_state_dict(ModelConfig(kind="lj"), 4, 5.0, 1.7)
    rems = set()
    for n in (61, 97, 130, 171, 222, 258, 301):
        pos, box = workloads.lj_box(n, seed=100 + n)
        rc = 0.45 * float(box)                      # keeps a few thousand edges in play at these sizes
        p = torch.from_numpy(pos).float()
        outs = {}
        for tag, kw in (("main", MAIN_ONLY), ("small", SMALL_ONLY)):
            eng = _engine(sd, n, box, rc, **kw)
            outs[tag] = eng.forward(p).cpu().numpy().copy()
            if tag == "main":
                E = eng.counts()[0]
                edges = torch.from_numpy(eng.debug_edges()).long()
            eng.close()
        rems.add(E % 32)
        assert np.array_equal(outs["main"], outs["small"]), (n, E)
        ref = orc.forward(sd, torch.remainder(p, float(box)), edges, box).numpy()
        assert rel_err(outs["main"], ref) < TOL, (n, E, rel_err(outs["main"], ref))
    assert len(rems) >= 5 and any(r != 0 for r in rems), rems


def test_zero_row_survives_calls_with_changing_edge_counts():
    """One engine, positions that change the edge count (and its remainder mod 32) from call to call: every result equals
    that of a fresh engine on the same positions bit for bit — nothing a previous call left in the padded rows leaks in."""
    n = 258
    pos, box = workloads.lj_box(n, seed=7)
    rc = 7.5
    sd = make_state_dict(ModelConfig(kind="lj"), 1, 6.0, 2.0)
    rng = np.random.default_rng(3)
    eng = _engine(sd, n, box, rc, **MAIN_ONLY)
    seen = set()
    x = pos.copy()
    for step in range(6):
        p = torch.from_numpy(x).float()
        out = eng.forward(p).cpu().numpy().copy()
        seen.add(eng.counts()[0] % 32)
        fresh = _engine(sd, n, box, rc, **MAIN_ONLY)
        ref = fresh.forward(p).cpu().numpy().copy()
        fresh.close()
        assert np.array_equal(out, ref), step
        x = x + rng.normal(0.0, 0.35, x.shape)
    eng.close()
    assert len(seen) >= 3, seen


@pytest.mark.parametrize("dtype, tol", [("f32", TOL), ("bf16", 1e-2)])
def test_water_with_bonds_both_kernels_and_bf16_padding(dtype, tol):
    """TIP3P golden (774 atoms, bond feature): fp32 throughput kernel == latency kernel bit for bit and == the reference
    output within 1e-5; the bf16 kernel (same zero-row padding, permuted W4 image) within its restated tolerance."""
    g, cfg, sd = load_golden("tip3p774_seed3")
    box, rc, n = float(g["box"]), float(g["cutoff"]), g["pos"].shape[0]
    posw = torch.from_numpy(np.mod(g["pos"], box).astype(np.float32))
    species = g["node_feat"].reshape(-1) != 0
    if dtype == "f32":
        outs = []
        for kw in (MAIN_ONLY, SMALL_ONLY):
            eng = _engine(sd, n, box, rc, bond=g["bond"], **kw)
            outs.append(eng.forward(posw, species=species).cpu().numpy().copy())
            eng.close()
        assert np.array_equal(outs[0], outs[1])
        assert rel_err(outs[0], g["out_norm"]) < tol
    else:
        eng = _engine(sd, n, box, rc, bond=g["bond"], edge_dtype="bf16")
        out = eng.forward(posw, species=species).cpu().numpy().copy()
        again = eng.forward(posw, species=species).cpu().numpy().copy()
        eng.close()
        assert np.array_equal(out, again)
        err = rel_err(out, g["out_norm"])
        assert 1e-6 < err < tol, err


def test_atom_count_beyond_the_32_bit_row_offsets_is_refused():
    """The conv-layer edge kernels address node-table rows with 32-bit byte offsets (row * 512 B): gamd_create refuses a box
    that would overflow them instead of gathering from wrapped addresses."""
    from gamd_amd._lib import GamdError
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2)
    with pytest.raises(GamdError, match="at most"):
        _engine(sd, (1 << 23) - 1, 1.0e4, 10.2)


def test_model_level_call_with_several_graphs():
    """pnet_model(pos_lst, edge_idx_lst) with len(pos_lst) > 1 (the reference batches the graphs with dgl.batch,
    nn_module.py:655-661): against the reference's own output for two graphs, and for the water wrapper (feat = the
    concatenated node features) against the graphs evaluated one by one."""
    import os
    from helpers import GOLDEN
    from gamd_amd.compat import ParticleNetLightningLJ, ParticleNetLightningWater
    g = dict(np.load(os.path.join(GOLDEN, "lj258_batch2_seed0.npz"), allow_pickle=False))
    kind, H, D, Eh, L, bond = [str(x) for x in g["cfg"]][:6]
    cfg = ModelConfig(kind=kind, encoding_size=int(H), hidden_dim=int(D), edge_embedding_dim=int(Eh), conv_layer=int(L))
    sd = make_state_dict(cfg, int(g["seed"]), float(g["length_mean"]), float(g["length_std"]))
    m = ParticleNetLightningLJ(state_dict=sd, num_atoms=258, box_size=float(g["box"]), cutoff=float(g["cutoff"]))
    pos_lst = [torch.from_numpy(g[f"pos{i}"]).cuda() for i in range(2)]
    edge_lst = [torch.from_numpy(g[f"edge_idx{i}"]).long().cuda() for i in range(2)]
    out = m.pnet_model(pos_lst, edge_lst)
    assert tuple(out.shape) == (516, 3)
    assert rel_err(out.cpu().numpy(), g["out_norm"]) < TOL
    # graphs of different sizes are allowed since round 5 (dgl.batch takes any); an edge list that names atoms its graph does
    # not have is refused by the library
    from gamd_amd._lib import GamdError
    with pytest.raises(GamdError, match="outside"):
        m.pnet_model([pos_lst[0], pos_lst[1][:100]], edge_lst)
    # water: feat rows are split by graph
    gw, cfgw, sdw = load_golden("tip3p774_seed3")
    box = float(gw["box"])
    posw = torch.from_numpy(np.mod(gw["pos"], box)).float().cuda()
    e = torch.from_numpy(gw["edge_idx"]).long().cuda()
    feat = torch.from_numpy(gw["node_feat"]).cuda()
    mw = ParticleNetLightningWater(state_dict=sdw)
    one = mw.pnet_model([posw], feat, [e])
    two = mw.pnet_model([posw, posw], torch.cat([feat, feat]), [e, e])
    assert tuple(two.shape) == (2 * posw.shape[0], 3)
    assert np.array_equal(two[: posw.shape[0]].cpu().numpy(), one.cpu().numpy())
    assert np.array_equal(two[posw.shape[0]:].cpu().numpy(), one.cpu().numpy())
    assert rel_err(one.cpu().numpy(), gw["out_norm"]) < TOL
