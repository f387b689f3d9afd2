"""BASELINE config 4 — the ensemble of eight independent 10 000-atom LJ boxes, one per GPU — as far as ONE GPU can test it,
plus the tests that switch themselves on when the box has a second device.

* the eight rank inputs themselves (`lj_box(10000, seed = 1234 + rank)`, what `bench.py --gpus 8` hands to rank 0 .. 7):
  exact edge set against the oracle's own O(N^2) search and forces against the oracle on ITS edge list, max-norm and
  per-atom p99 < 1e-5 — rounds 1-4 only ever tested seed 1234;
* the same eight boxes as ONE `n_boxes = 8` batch (80 000 atoms, what `secondary.c2_batch8` times): bit-identical to the
  boxes one by one;
* >= 2 devices (skipped — visibly — on the one-GPU box): a handle on device 1, two handles on two devices in one process
  giving the bits of device 0 (the per-device `hipFuncSetAttribute` guards), and `bench.py --gpus 2` on real RCCL.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import gamd_oracle as orc
from helpers import rel_err, edge_set, per_atom_err, edge_set_diff_near_cutoff
from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS
from gamd_amd import workloads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL, P99_TOL, NEAR_CUTOFF = 1e-5, 1e-5, 1e-4
N, RC, RANKS = 10000, 3.0 * workloads.LJ_SIGMA, 8


def _engine(*a, **kw):
    from gamd_amd.engine import GamdForce
    return GamdForce(*a, **kw)


def _sd():
    return make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2)          # bench.py's C2 weights


@pytest.fixture(scope="module")
def rank_boxes():
    boxes = [workloads.lj_box(N, seed=1234 + r) for r in range(RANKS)]
    assert len({b for _, b in boxes}) == 1
    assert len({p.tobytes() for p, _ in boxes}) == RANKS                 # eight different microstates
    return [p for p, _ in boxes], boxes[0][1]


@pytest.mark.parametrize("rank", range(RANKS))
def test_each_ranks_box_matches_the_oracle(rank, rank_boxes):
    """Rank r's input of `bench.py --gpus 8`: edge set == the oracle's own search, forces == the oracle on its own edge list.
    The library evaluates the cutoff test op for op as the oracle's restatement does (neighbor.hip: gamd_mask_d2), so the
    sets are expected to be EQUAL (seed 1237 has a pair 3e-5 from the cutoff that an earlier build put on the other side);
    should a pair within 1e-4 of the cutoff ever flip again, the set check still bounds it and the forces are then compared
    on the GPU's list (a network is not continuous in its edge set)."""
    pos, box = rank_boxes[0][rank], rank_boxes[1]
    sd = _sd()
    eng = _engine(sd, N, box, RC, scaler=SHIPPED_SCALERS["lj"])
    p = torch.from_numpy(pos).float()
    out = eng.forward(p).cpu().numpy()
    pw = torch.remainder(p, float(box))
    ref_edges = orc.neighbor_edges(pw, box, RC, "jaxmd")
    got = eng.debug_edges()
    pairs, dist = edge_set_diff_near_cutoff(got, ref_edges.numpy(), pw.numpy(), box, RC, N)
    assert len(pairs) <= 8 and (len(pairs) == 0 or dist.max() < NEAR_CUTOFF), (rank, len(pairs))
    assert len(np.unique(edge_set(got))) == got.shape[1] and int((got[0] == got[1]).sum()) == N
    if len(pairs):
        ref_edges = torch.from_numpy(got).long()
    ref = orc.forward(sd, pw, ref_edges, box).numpy()
    med, p99, worst, cnt = per_atom_err(out, ref)
    print(f"rank {rank} (seed {1234 + rank}): E={got.shape[1]} max-norm {rel_err(out, ref):.2e}; per atom median {med:.2e} "
          f"p99 {p99:.2e} max {worst:.2e} over {cnt} atoms; {len(pairs)} near-cutoff pairs differ")
    assert rel_err(out, ref) < TOL
    assert cnt > 0.9 * N and p99 < P99_TOL, (med, p99, worst, cnt)
    eng.close()


def test_the_eight_rank_boxes_as_one_batch_are_bit_identical_to_the_ranks(rank_boxes):
    """`secondary.c2_batch8`: n_boxes = 8, 80 000 atoms in one set of launches.  Forces (normalised and denormalised) and the
    per-box edge lists equal the eight single-box evaluations bit for bit; a short BAOAB run (exact rebuild every step, box b
    with seed + b) ends on the single boxes' positions bit for bit."""
    poss, box = rank_boxes
    sd = _sd()
    batch = _engine(sd, N, box, RC, scaler=SHIPPED_SCALERS["lj"], n_boxes=RANKS)
    single = _engine(sd, N, box, RC, scaler=SHIPPED_SCALERS["lj"])
    allpos = torch.from_numpy(np.concatenate(poss)).float().cuda()
    out = batch.forward(allpos).cpu().numpy()
    den = batch.forward(allpos, denormalize=True).cpu().numpy()
    edges = batch.debug_edges()
    assert np.all(edges[0] // N == edges[1] // N)                        # no edge crosses boxes
    row_ptr, _ = batch.debug_csr()
    assert all(row_ptr[b * N] % 16 == 0 for b in range(RANKS))           # every box starts on a chunk boundary
    for b in range(RANKS):
        sl = slice(b * N, (b + 1) * N)
        one = single.forward(allpos[sl]).cpu().numpy()
        assert np.array_equal(out[sl], one), b
        assert np.array_equal(den[sl], single.forward(allpos[sl], denormalize=True).cpu().numpy()), b
        eb = edges[:, edges[0] // N == b] - b * N
        assert np.array_equal(eb, single.debug_edges()), b              # same edges in the same CSR order
    # 3 MD steps, seed 7 + b per box (what rank b's gamd_md_run(seed = 7 + rank) draws)
    md = dict(dt_ps=0.002, mass_amu=39.9, temperature_k=100.0, gamma_per_ps=25.0)
    vel = np.concatenate([workloads.maxwell_boltzmann(N, seed=99 + b) for b in range(RANKS)])
    x, v = allpos.clone(), torch.from_numpy(vel).float().cuda()
    f = batch.forward(x, denormalize=True)
    batch.md_run(x, v, f, 3, seed=7, **md)
    for b in (0, 3, 7):
        sl = slice(b * N, (b + 1) * N)
        xs, vs = allpos[sl].clone(), torch.from_numpy(vel[sl]).float().cuda()
        fs = single.forward(xs, denormalize=True)
        single.md_run(xs, vs, fs, 3, seed=7 + b, **md)
        assert torch.equal(x[sl], xs) and torch.equal(v[sl], vs) and torch.equal(f[sl], fs), b
    batch.close(); single.close()


def test_two_handles_in_one_process_with_runs_in_flight_on_two_streams():
    """The one-GPU analogue of 'one stream per device from one process' (SURVEY 8e): two handles — an LJ box and a bf16 rigid
    water box, so different kernel families, dynamic-LDS sizes and neighbour paths — with MD runs enqueued on two streams at
    the same time give the bits each gives alone (nothing mutable is shared between handles: launch-attribute guards, error
    slots, counters, scratch)."""
    from helpers import load_golden
    pos, box = workloads.lj_box(N, seed=1236)
    sd = _sd()
    _, _, wsd = load_golden("tip3p774_seed3")
    n_mol = 216
    wpos, wbox, species, bonds = workloads.water_box(n_mol, seed=31, jitter=0.0, wrap=False)
    nw = 3 * n_mol
    mass = np.where(species == 1, workloads.MASS_O, workloads.MASS_H).astype(np.float64).reshape(-1, 1)
    pairs, _ = orc.water_constraints(nw, workloads.TIP3P_R_OH, workloads.TIP3P_R_HH)
    wv0 = np.random.default_rng(32).normal(0, 1.0, (nw, 3)) * 10.0 * np.sqrt(workloads.KB * 300.0 / mass)
    wv0 = orc.rattle_velocities(wpos, wv0, (1.0 / mass).reshape(-1), pairs)
    wmd = dict(dt_ps=0.0005, mass_amu=workloads.MASS_O, mass_h_amu=workloads.MASS_H, temperature_k=300.0, rigid_water=True,
               r_oh=workloads.TIP3P_R_OH, r_hh=workloads.TIP3P_R_HH, species=species, seed=5)

    def run(which, stream_a=None, stream_b=None):
        out = {}
        cur = torch.cuda.current_stream()
        sa, sb = (stream_a or cur), (stream_b or cur)
        a = b = None
        if "a" in which:
            a = _engine(sd, N, box, RC, scaler=SHIPPED_SCALERS["lj"], neighbor_skin=RC / 6.0)
            xa = torch.from_numpy(pos).float().cuda()
            va = torch.from_numpy(workloads.maxwell_boltzmann(N, seed=3)).float().cuda()
            fa = a.forward(xa, denormalize=True).clone()
        if "b" in which:
            b = _engine(wsd, nw, wbox, 4.2, bond=bonds, scaler=SHIPPED_SCALERS["tip3p"], edge_dtype="bf16", neighbor_skin=0.7)
            xb = torch.from_numpy(wpos).float().cuda()
            vb = torch.from_numpy(wv0).float().cuda()
            fb = b.forward(xb, species=species, denormalize=True).clone()
        torch.cuda.synchronize()
        for _ in range(3):                                             # interleaved enqueues, nothing synchronised in between
            if a is not None:
                with torch.cuda.stream(sa):
                    a.md_run(xa, va, fa, 10, seed=9, sync=False)
            if b is not None:
                with torch.cuda.stream(sb):
                    b.md_run(xb, vb, fb, 25, sync=False, **wmd)
        if a is not None:
            with torch.cuda.stream(sa):
                assert a.sync_status() == 0
            out["a"] = (xa.cpu(), fa.cpu())
            a.close()
        if b is not None:
            with torch.cuda.stream(sb):
                assert b.sync_status() == 0
            out["b"] = (xb.cpu(), fb.cpu())
            b.close()
        return out

    alone_a, alone_b = run("a")["a"], run("b")["b"]
    both = run("ab", torch.cuda.Stream(), torch.cuda.Stream())
    assert torch.isfinite(both["b"][1]).all() and torch.isfinite(both["a"][1]).all()
    assert torch.equal(both["a"][0], alone_a[0]) and torch.equal(both["a"][1], alone_a[1])
    assert torch.equal(both["b"][0], alone_b[0]) and torch.equal(both["b"][1], alone_b[1])


# ---- >= 2 devices: switch themselves on where a second GPU is visible ---------------------------------------------------
two_devices = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 visible HIP devices (one-GPU box)")


@two_devices
def test_a_handle_on_device_1_gives_the_bits_of_device_0():
    """gamd_config.device = 1 while the caller's current device stays 0: every kernel family that needs more than 64 KiB of
    dynamic LDS (fp32, bf16, split-fp16, the generic-width ones) on a device that is not the first the process touched."""
    from helpers import load_golden
    assert torch.cuda.current_device() == 0
    for name, dtypes in (("lj258_seed0", ("f32", "bf16", "f16x3")), ("tip3p774_w256_seed10", ("f32", "f16x3", "bf16"))):
        g, cfg, sd = load_golden(name)
        box, rc, n = float(g["box"]), float(g["cutoff"]), g["pos"].shape[0]
        posw = np.mod(g["pos"], box).astype(np.float32)
        bond = g["bond"] if "bond" in g else None
        species = (g["node_feat"].reshape(-1) != 0) if "node_feat" in g else None
        for dt in dtypes:
            outs = []
            for dev in (0, 1):
                eng = _engine(sd, n, box, rc, bond=bond, device=dev, edge_dtype=dt, small_tile_limit=-1)
                outs.append(eng.forward(torch.from_numpy(posw).to(f"cuda:{dev}"), species=species).cpu().numpy())
                assert torch.cuda.current_device() == 0               # the caller's device is restored
                eng.close()
            assert np.array_equal(outs[0], outs[1]), (name, dt)
            if dt == "f32":
                assert rel_err(outs[1], g["out_norm"]) < TOL


@two_devices
def test_two_handles_on_two_devices_in_one_process():
    """SURVEY 8e's 'one stream per device from one process': interleaved evaluations and MD runs on two handles give what
    each gives alone; the C2 box on both devices at once (throughput kernels: 132 KiB of dynamic LDS on each)."""
    pos, box = workloads.lj_box(N, seed=1235)
    sd = _sd()
    engs = [_engine(sd, N, box, RC, scaler=SHIPPED_SCALERS["lj"], device=d, neighbor_skin=RC / 6.0) for d in (0, 1)]
    xs = [torch.from_numpy(pos).float().to(f"cuda:{d}") for d in (0, 1)]
    vs = [torch.from_numpy(workloads.maxwell_boltzmann(N, seed=100)).float().to(f"cuda:{d}") for d in (0, 1)]
    fs = [e.forward(x, denormalize=True) for e, x in zip(engs, xs)]
    assert torch.equal(fs[0].cpu(), fs[1].cpu())
    for e, x, v, f in zip(engs, xs, vs, fs):                             # both runs in flight at once
        e.md_run(x, v, f, 20, seed=11, sync=False)
    for e in engs:
        assert e.sync_status() == 0
    assert torch.equal(xs[0].cpu(), xs[1].cpu()) and torch.equal(fs[0].cpu(), fs[1].cpu())
    for e in engs:
        e.close()


@two_devices
def test_bench_gpus_2_on_real_rccl():
    """`bench.py --gpus 2` with backend nccl (= RCCL) on two devices: two distinct GPUs, both ranks saw a group of size 2,
    and each rank's time is within 3 % of a --gpus 1 run (no collective on the step path: weak scaling is flat)."""
    def run(*flags):
        e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                               "TORCHELASTIC_RUN_ID", "GAMD_BENCH_SHARE_GPU", "GAMD_BENCH_BACKEND")}
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            det = os.path.join(td, "detail.json")
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "20", "--no-cpu-baseline",
                                "--no-secondary", "--detail", det, *flags], capture_output=True, text=True, timeout=900, cwd=ROOT, env=e)
            assert p.returncode == 0, p.stderr[-2000:]
            lines = [l for l in p.stdout.splitlines() if l.strip()]
            assert len(lines) == 1 and len(lines[0]) < 6000, p.stdout
            assert json.loads(lines[0])["ensemble"]["boxes"] == len(json.load(open(det))["detail"]["ensemble"]["per_rank"])
            return json.load(open(det))["detail"]                  # the full record: per-rank summaries
    one, two = run("--gpus", "1"), run("--gpus", "2")
    ens = two["ensemble"]
    assert two["n_gpus"] == 2 and ens["distinct_devices"] == 2 and ens["collective_on_step_path"] is False
    assert [r["group_world_size"] for r in ens["per_rank"]] == [2, 2] and [r["device"] for r in ens["per_rank"]] == [0, 1]
    assert [r["box_seed"] for r in ens["per_rank"]] == [1234, 1235]
    t1 = one["ensemble"]["per_rank"][0]["seconds"]
    for r in ens["per_rank"]:
        assert abs(r["seconds"] - t1) / t1 < 0.03, (r["seconds"], t1)
    assert two["value"] > 1.9 * one["value"]
