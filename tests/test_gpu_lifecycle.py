"""Handle life cycle, threads, and the barrier-free multi-workgroup cell build.

`gamd_create` ... `gamd_destroy` gives back what it took:

A force provider in a serving process is created and destroyed many times (one handle per system size / per request class).
Everything a handle owns is allocated by the library itself (`hipMalloc`, `hipHostMalloc`, event pools), not through torch's
caching allocator, so `hipMemGetInfo` before and after a run of create / use / destroy cycles shows a leak directly.
Different handles may be driven from different threads at once (include/gamd_hip.h).  The candidate rebuild's cell build runs on
32 workgroups that never wait for each other (neighbor.hip, k_cells_sliced): checked against the exact build on the same grid.
"""
import gc
import os

import numpy as np
import pytest
import torch

from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS
from gamd_amd import workloads

pytestmark = pytest.mark.gpu
MB = 1 << 20


def _free_bytes():
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return torch.cuda.mem_get_info()[0]


def _rss_bytes():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")


def _cycle(i):
    """One create / use / destroy cycle; the configuration rotates through every kernel family and buffer set."""
    from gamd_amd.engine import GamdForce
    k = i % 6
    if k == 0:      # LJ, exact rebuild every call, timing pools on
        pos, box = workloads.lj_box(2000, seed=10 + i)
        eng = GamdForce(make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2), 2000, box, 10.2, scaler=SHIPPED_SCALERS["lj"])
        eng.timing_enable(True)
        x = torch.from_numpy(pos).float().cuda()
        eng.forward(x)
        eng.timing_read()
    elif k == 1:    # LJ, Verlet skin + on-device Langevin run (candidate buffers, reference positions, event pools of the steps)
        pos, box = workloads.lj_box(3000, seed=10 + i)
        eng = GamdForce(make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2), 3000, box, 10.2, scaler=SHIPPED_SCALERS["lj"],
                        neighbor_skin=1.7)
        x = torch.from_numpy(pos).float().cuda()
        v = torch.from_numpy(workloads.maxwell_boltzmann(3000, seed=i)).float().cuda()
        f = eng.forward(x, denormalize=True).clone()
        eng.md_run(x, v, f, 20, seed=i)
    elif k == 2:    # water, bond feature, bf16 edge MLP, Nose-Hoover chain (ke partials, chain state)
        pos, box, species, bonds = workloads.water_box(300, seed=10 + i)
        eng = GamdForce(make_state_dict(ModelConfig(kind="water", use_bond=True), 1, 2.9, 1.1), 900, box, 4.2, bond=bonds,
                        scaler=SHIPPED_SCALERS["tip3p"], edge_dtype="bf16", neighbor_skin=0.7)
        x = torch.from_numpy(pos).float().cuda()
        v = torch.zeros_like(x)
        f = eng.forward(x, species=species, denormalize=True).clone()
        eng.md_run_nhc(x, v, f, 5, dt_ps=0.0005, mass_amu=workloads.MASS_O, mass_h_amu=workloads.MASS_H, temperature_k=300.0,
                       species=species)
    elif k == 3:    # several boxes per launch (per-box tables, scan scratch)
        boxes = [workloads.lj_box(500, seed=100 + i + b) for b in range(4)]
        eng = GamdForce(make_state_dict(ModelConfig(kind="lj"), 0, 5.3, 1.6), 500, boxes[0][1], 7.5, n_boxes=4,
                        scaler=SHIPPED_SCALERS["lj"])
        x = torch.from_numpy(np.stack([p for p, _ in boxes])).float().cuda()
        eng.forward(x)
    elif k == 4:    # the 256 / 256 / 128 x 5 dynamic-box model (generic-width kernels), split-fp16
        pos, box, species, _ = workloads.water_box(200, seed=10 + i)
        cfg = ModelConfig(kind="dynbox", encoding_size=256, hidden_dim=128, edge_embedding_dim=256, conv_layer=5)
        eng = GamdForce(make_state_dict(cfg, 2, 2.9, 1.1), 600, box, 4.2, edge_dtype="f16x3")
        x = torch.from_numpy(pos).float().cuda()
        eng.forward(x, species=species)
    else:           # the reference's own 258-atom size (single-workgroup neighbour path), skin mode
        pos, box = workloads.lj_box(258, seed=10 + i)
        eng = GamdForce(make_state_dict(ModelConfig(kind="lj"), 0, 5.3, 1.6), 258, box, 7.5, scaler=SHIPPED_SCALERS["lj"],
                        neighbor_skin=1.25)
        x = torch.from_numpy(pos).float().cuda()
        v = torch.from_numpy(workloads.maxwell_boltzmann(258, seed=i)).float().cuda()
        f = eng.forward(x, denormalize=True).clone()
        eng.md_run(x, v, f, 50, seed=i)
    torch.cuda.synchronize()
    eng.close()


def test_create_use_destroy_cycles_give_device_and_host_memory_back():
    from gamd_amd.engine import GamdForce
    for i in range(6):                       # first pass: code objects, torch's context, the library's lazy state
        _cycle(i)
    free0, rss0 = _free_bytes(), _rss_bytes()
    # the instrument sees the library's allocations: a live 10 000-atom handle holds several hundred MiB
    pos, box = workloads.lj_box(10000)
    eng = GamdForce(make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2), 10000, box, 10.2, scaler=SHIPPED_SCALERS["lj"])
    eng.forward(torch.from_numpy(pos).float().cuda())
    held = free0 - _free_bytes()
    eng.close()
    assert held > 300 * MB, held
    assert free0 - _free_bytes() < 8 * MB
    n = 30
    for i in range(6, 6 + n):
        _cycle(i)
    free1, rss1 = _free_bytes(), _rss_bytes()
    print(f"{n} cycles: device free {free0 / MB:.1f} -> {free1 / MB:.1f} MiB, host RSS {rss0 / MB:.1f} -> {rss1 / MB:.1f} MiB")
    assert free0 - free1 < 8 * MB, (free0, free1)            # a leaked 2 000-atom handle alone is > 100 MiB
    assert rss1 - rss0 < 64 * MB, (rss0, rss1)


def test_close_is_idempotent_and_a_closed_engine_fails_loudly():
    from gamd_amd.engine import GamdForce
    from gamd_amd._lib import GamdError
    pos, box = workloads.lj_box(500, seed=3)
    eng = GamdForce(make_state_dict(ModelConfig(kind="lj"), 0, 5.3, 1.6), 500, box, 7.5)
    x = torch.from_numpy(pos).float().cuda()
    eng.forward(x)
    eng.close()
    eng.close()
    with pytest.raises((GamdError, ValueError, RuntimeError)):
        eng.forward(x)


def test_two_threads_each_with_its_own_handle_and_stream():
    """include/gamd_hip.h: a handle is for one caller at a time, different handles may be driven from different threads at once
    (ctypes releases the GIL around every call).  Two threads, two handles of different kernel families, own streams, 40 calls
    each, against the same calls made one after the other."""
    import threading
    from gamd_amd.engine import GamdForce
    pos, box = workloads.lj_box(4000, seed=5)
    wpos, wbox, species, bonds = workloads.water_box(400, seed=6)
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2)
    wsd = make_state_dict(ModelConfig(kind="water", use_bond=True), 1, 2.9, 1.1)
    rng = np.random.default_rng(0)
    xs = [torch.from_numpy(pos + rng.normal(0, 0.05, pos.shape)).float().cuda() for _ in range(8)]
    ws = [torch.from_numpy(wpos + rng.normal(0, 0.01, wpos.shape)).float().cuda() for _ in range(8)]

    def work_a(out, stream):
        eng = GamdForce(sd, 4000, box, 10.2, scaler=SHIPPED_SCALERS["lj"], neighbor_skin=1.7)
        with torch.cuda.stream(stream):
            for k in range(40):
                out.append(eng.forward(xs[k % 8]).clone())
            stream.synchronize()
        eng.close()

    def work_b(out, stream):
        eng = GamdForce(wsd, 1200, wbox, 4.2, bond=bonds, scaler=SHIPPED_SCALERS["tip3p"], edge_dtype="f16x3")
        with torch.cuda.stream(stream):
            for k in range(40):
                out.append(eng.forward(ws[k % 8], species=species).clone())
            stream.synchronize()
        eng.close()

    ref_a, ref_b, got_a, got_b = [], [], [], []
    work_a(ref_a, torch.cuda.Stream())
    work_b(ref_b, torch.cuda.Stream())
    errs = []

    def guarded(fn, *a):
        try:
            fn(*a)
        except Exception as e:                                         # surfaces in the main thread below
            errs.append(e)

    ta = threading.Thread(target=guarded, args=(work_a, got_a, torch.cuda.Stream()))
    tb = threading.Thread(target=guarded, args=(work_b, got_b, torch.cuda.Stream()))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errs, errs
    assert len(got_a) == 40 and len(got_b) == 40
    assert all(torch.equal(g, r) for g, r in zip(got_a, ref_a))
    assert all(torch.equal(g, r) for g, r in zip(got_b, ref_b))
    assert torch.isfinite(got_a[-1]).all() and torch.isfinite(got_b[-1]).all()


@pytest.mark.parametrize("case", ["water6000", "lj10000", "batch38x258", "dilute_many_cells", "overfull_cells", "tiny_box_one_cell"])
def test_candidate_rebuild_by_sliced_workgroups_sorts_like_the_exact_build(case):
    """neighbor.hip k_cells_sliced: 32 workgroups that each histogram all atoms and then sort their own share of the cells, with no
    barrier between them.  The sorted order (perm) of a skin-mode handle after a candidate rebuild equals that of an exact build
    on the same cell grid (a handle with cutoff = rc + skin), its edge set and forces those of the handle that rebuilds exactly
    every call — for the BASELINE water / LJ sizes, a batch of boxes
    (cells numbered box after box), a grid of more than 4 096 cells (the one-workgroup form still serves those), cells with more
    than 64 atoms (serial sort fallback inside a slice) and a box that is one cell."""
    from gamd_amd.engine import GamdForce
    kw, fkw = dict(scaler=SHIPPED_SCALERS["lj"]), {}
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2)
    if case == "water6000":
        pos, box, species, bonds = workloads.water_box(2000, mol_per_20A3=251.0, seed=3456)
        sd = make_state_dict(ModelConfig(kind="water", use_bond=True), 1, 2.9, 1.1)
        n, rc, skin = 6000, 4.2, 0.7
        kw, fkw = dict(bond=bonds, scaler=SHIPPED_SCALERS["tip3p"]), dict(species=species)
    elif case == "lj10000":
        pos, box = workloads.lj_box(10000)
        n, rc, skin = 10000, 10.2, 1.7
    elif case == "batch38x258":
        boxes = [workloads.lj_box(258, seed=50 + b) for b in range(38)]
        pos, box = np.concatenate([p for p, _ in boxes]), boxes[0][1]
        n, rc, skin = 258, 7.5, 1.25
        kw["n_boxes"] = 38
    elif case == "dilute_many_cells":
        rng = np.random.default_rng(7)
        box = 120.0
        pos = rng.uniform(0, box, (5000, 3))
        n, rc, skin = 5000, 5.0, 1.0                     # 20^3 = 8 000 cells > 4 096: k_cells_one_wg
    elif case == "overfull_cells":
        rng = np.random.default_rng(8)
        box = 24.0
        pos = rng.uniform(0, box, (1500, 3))             # 2^3 cells of ~190 atoms
        n, rc, skin = 1500, 9.0, 1.5
    else:
        rng = np.random.default_rng(9)
        box = 9.0
        pos = rng.uniform(0, box, (1100, 3))
        n, rc, skin = 1100, 7.0, 1.0
    from helpers import edge_set, rel_err
    x = torch.from_numpy(np.ascontiguousarray(pos)).float().cuda()
    exact = GamdForce(sd, n, box, rc, **kw)
    grid = GamdForce(sd, n, box, rc + skin, **kw)          # exact build on the candidate pass's cell grid (cells >= rc + skin)
    skinned = GamdForce(sd, n, box, rc, neighbor_skin=skin, **kw)
    for step, xs in enumerate((x, x + torch.from_numpy(np.random.default_rng(1).normal(0, 0.6 * skin, pos.shape)).float().cuda())):
        # (second round: the atoms have moved by more than half the skin: another rebuild, from a used state)
        f0 = exact.forward(xs, **fkw)
        grid.forward(xs, **fkw)
        f1 = skinned.forward(xs, **fkw)
        assert skinned.skin_stats()[0] == step + 1                        # the call rebuilt the candidate list
        assert np.array_equal(grid.debug_perm(), skinned.debug_perm())    # same cells, same order inside every cell
        e0, e1 = edge_set(exact.debug_edges()), edge_set(skinned.debug_edges())
        assert len(e0) == len(e1) and np.array_equal(np.sort(e0), np.sort(e1))
        assert torch.isfinite(f0).all() and rel_err(f1.cpu().numpy(), f0.cpu().numpy()) < 1e-5
    exact.close(); grid.close(); skinned.close()


@pytest.mark.parametrize("n,box,rc", [(3000, 100.0, 6.0), (1030, 60.0, 5.0), (16384, 150.0, 6.0), (5000, 40.0, 6.0)])
def test_fill_pass_with_its_own_row_scan_on_rows_of_every_kind(n, box, rc):
    """neighbor.hip k_filter_fill_scan (one box, 1 024 < n <= 16 384): the fill workgroups compute their own row pointers.  The
    `md_module.get_neighbor` flavour has no self edges, so a dilute gas has EMPTY rows, rows that start on and off 16-edge chunk
    boundaries and chunks spanning many rows; sizes just above 1 024, at 16 384 (the last size on this path) and a dense box.
    Edge set, per-row degrees and forces against the handle that rebuilds exactly (k_count | k_scan_deg | k_fill | k_chunk_meta),
    over three calls with moving atoms."""
    from gamd_amd.engine import GamdForce
    from helpers import edge_set, rel_err
    rng = np.random.default_rng(n)
    pos = rng.uniform(0, box, (n, 3))
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 5.0, 1.5)
    exact = GamdForce(sd, n, box, rc, nbr_flavour="torch")
    skinned = GamdForce(sd, n, box, rc, nbr_flavour="torch", neighbor_skin=rc / 6.0)
    for step in range(3):
        x = torch.from_numpy(pos + step * rng.normal(0, 0.1, pos.shape)).float().cuda()
        f0, f1 = exact.forward(x), skinned.forward(x)
        e0, e1 = exact.debug_edges(), skinned.debug_edges()
        assert e0.shape == e1.shape and np.array_equal(edge_set(e0), edge_set(e1))
        assert np.array_equal(np.bincount(e0[0], minlength=n), np.bincount(e1[0], minlength=n))
        assert (np.bincount(e0[0], minlength=n) == 0).any() or box < 50.0        # the dilute cases do have empty rows
        assert torch.isfinite(f1).all() and rel_err(f1.cpu().numpy(), f0.cpu().numpy()) < 1e-5
        assert exact.counts()[:2] != (0, 0) and skinned.counts()[0] == exact.counts()[0]
    exact.close(); skinned.close()


def test_host_species_are_uploaded_once_and_outlive_an_asynchronous_run():
    """engine._dev_species: the device copy of a host-side species array belongs to the engine (kernels of an md_run(sync=False)
    still read it after the call has returned) and is reused while the content is the same — same pointer, so the library's
    O,H,H layout check (a device -> host copy + stream synchronisation) runs once, not once per call.  Changed content or another
    stream gets its own copy."""
    from gamd_amd.engine import GamdForce
    pos, box, species, bonds = workloads.water_box(300, seed=12, jitter=0.0, wrap=False)
    eng = GamdForce(make_state_dict(ModelConfig(kind="water", use_bond=True), 1, 2.9, 1.1), 900, box, 4.2, bond=bonds,
                    scaler=SHIPPED_SCALERS["tip3p"], neighbor_skin=0.7)
    x = torch.from_numpy(pos).float().cuda()
    v = torch.zeros_like(x)
    f = eng.forward(x, species=species, denormalize=True).clone()
    p0 = eng._species_cache[1].data_ptr()
    kw = dict(dt_ps=0.0005, mass_amu=workloads.MASS_O, mass_h_amu=workloads.MASS_H, temperature_k=300.0, rigid_water=True,
              r_oh=workloads.TIP3P_R_OH, r_hh=workloads.TIP3P_R_HH, species=species, seed=1)
    for _ in range(3):
        eng.md_run(x, v, f, 5, sync=False, **kw)
        assert eng._species_cache[1].data_ptr() == p0
    assert eng.sync_status() == 0 and torch.isfinite(x).all()
    eng.forward(x, species=species.copy())                                 # another array, same content: same copy
    assert eng._species_cache[1].data_ptr() == p0
    flipped = species.copy(); flipped[0], flipped[1] = flipped[1], flipped[0]
    eng.forward(x, species=flipped)                                        # other content: a new copy
    assert not torch.equal(eng._species_cache[1].cpu(), torch.from_numpy((species != 0).astype(np.uint8)))
    with torch.cuda.stream(torch.cuda.Stream()):
        eng.forward(x, species=flipped)                                    # other stream: not the copy another stream made
    eng.close()


@pytest.mark.parametrize("seed", range(24))
def test_skin_path_against_exact_path_on_random_boxes(seed):
    """Seeded sweep over what the large-system skin path (k_skin_check | k_cells_sliced | k_filter_count | k_filter_fill_scan, or
    the five-launch form above 16 384 atoms) can meet: atom counts that are not multiples of the 8 rows of a fill workgroup,
    orthorhombic boxes, both search flavours, densities from a dilute gas to cells with > 64 atoms, skins from 2 % to 40 % of the
    cutoff.  Four calls with growing displacements (so reuse steps and rebuild steps both occur); every call: edge set and per-row
    degrees equal the exact path's, forces within 1e-5."""
    from gamd_amd.engine import GamdForce
    from helpers import edge_set, rel_err
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1025, 1031, 2048, 3001, 4099, 7777, 12345, 16384, 16391, 20011]))
    flavour = "jaxmd" if seed % 2 == 0 else "torch"
    rc = float(rng.uniform(4.0, 9.0))
    per_atom = float(rng.choice([1.5, 6.0, 25.0, 60.0]))                  # neighbours within the cutoff
    vol = n * (4.0 / 3.0) * np.pi * rc ** 3 / per_atom
    shape = rng.uniform(0.7, 1.4, 3)
    box = (vol / shape.prod()) ** (1.0 / 3.0) * shape
    box = np.maximum(box, 2.05 * rc)                                       # minimum image
    skin = float(rc * rng.choice([0.02, 1.0 / 6.0, 0.4]))
    pos = rng.uniform(0, 1, (n, 3)) * box
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 0.6 * rc, 0.2 * rc)
    exact = GamdForce(sd, n, box, rc, nbr_flavour=flavour)
    skinned = GamdForce(sd, n, box, rc, nbr_flavour=flavour, neighbor_skin=skin)
    step_sigma = [0.0, 0.1 * skin, 0.25 * skin, 0.6 * skin]
    rebuilds = []
    for k, sg in enumerate(step_sigma):
        pos = pos + rng.normal(0, sg, pos.shape) if sg > 0 else pos
        x = torch.from_numpy(pos).float().cuda()
        f0, f1 = exact.forward(x), skinned.forward(x)
        e0, e1 = exact.debug_edges(), skinned.debug_edges()
        assert e0.shape == e1.shape, (k, e0.shape, e1.shape)
        assert np.array_equal(edge_set(e0), edge_set(e1)), k
        assert np.array_equal(np.bincount(e0[0], minlength=n), np.bincount(e1[0], minlength=n))
        assert torch.isfinite(f1).all() and rel_err(f1.cpu().numpy(), f0.cpu().numpy()) < 1e-5
        rebuilds.append(skinned.skin_stats()[0])
    # (a dense first build can outgrow the initial candidate capacity: regrown and rebuilt, 2 on the first call)
    assert rebuilds[0] >= 1 and rebuilds[-1] > rebuilds[0]                 # the last displacement is beyond half the skin
    exact.close(); skinned.close()


# ---- a fresh handle driven from a non-blocking side stream from its first call on ---------------------------------------
# (what caught the hipMemset race of DESIGN.md section 7: buffers that a call allocates and zeroes on the NULL stream while the
# caller's stream is not ordered behind it — the momentum sums of a run's first step were wiped in 1 of ~300 runs)
def _sc_lj(stream):
    pos, box = workloads.lj_box(3000, seed=11)
    from gamd_amd.engine import GamdForce
    e = GamdForce(make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2), 3000, box, 10.2, scaler=SHIPPED_SCALERS["lj"], neighbor_skin=1.7)
    x = torch.from_numpy(pos).float().cuda(); v = torch.from_numpy(workloads.maxwell_boltzmann(3000, seed=1)).float().cuda()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        f = e.forward(x, denormalize=True).clone()
        e.md_run(x, v, f, 15, seed=2, sync=False, remove_cm_motion=True)
        e.sync_status()
    out = (x.cpu(), v.cpu(), f.cpu()); e.close(); return out

def _sc_water_nhc(stream):
    pos, box, species, bonds = workloads.water_box(300, seed=12, jitter=0.0, wrap=False)
    from gamd_amd.engine import GamdForce
    e = GamdForce(make_state_dict(ModelConfig(kind="water", use_bond=True), 1, 2.9, 1.1), 900, box, 4.2, bond=bonds,
                  scaler=SHIPPED_SCALERS["tip3p"], neighbor_skin=0.7)
    x = torch.from_numpy(pos).float().cuda(); v = torch.zeros_like(x)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        f = e.forward(x, species=species, denormalize=True).clone()
        ch = e.md_run_nhc(x, v, f, 10, dt_ps=0.0005, mass_amu=workloads.MASS_O, mass_h_amu=workloads.MASS_H, temperature_k=300.0,
                          species=species, rigid_water=True, r_oh=workloads.TIP3P_R_OH, r_hh=workloads.TIP3P_R_HH, sync=False)
        e.sync_status()
    out = (x.cpu(), v.cpu(), f.cpu(), ch.cpu()); e.close(); return out

def _sc_batch(stream):
    boxes = [workloads.lj_box(500, seed=100 + b) for b in range(4)]
    from gamd_amd.engine import GamdForce
    e = GamdForce(make_state_dict(ModelConfig(kind="lj"), 0, 5.3, 1.6), 500, boxes[0][1], 7.5, n_boxes=4, scaler=SHIPPED_SCALERS["lj"],
                  neighbor_skin=1.25)
    x = torch.from_numpy(np.concatenate([p for p, _ in boxes])).float().cuda()
    v = torch.from_numpy(workloads.maxwell_boltzmann(2000, seed=4)).float().cuda()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        f = e.forward(x, denormalize=True).clone()
        e.md_run(x, v, f, 12, seed=3, sync=False, remove_cm_motion=True)
        e.sync_status()
    out = (x.cpu(), v.cpu(), f.cpu()); e.close(); return out

def _sc_small(stream):
    pos, box = workloads.lj_box(258, seed=13)
    from gamd_amd.engine import GamdForce
    e = GamdForce(make_state_dict(ModelConfig(kind="lj"), 0, 5.3, 1.6), 258, box, 7.5, scaler=SHIPPED_SCALERS["lj"], neighbor_skin=1.25)
    x = torch.from_numpy(pos).float().cuda(); v = torch.from_numpy(workloads.maxwell_boltzmann(258, seed=5)).float().cuda()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        f = e.forward(x, denormalize=True).clone()
        e.md_run(x, v, f, 60, seed=6, sync=False, remove_cm_motion=True)
        e.sync_status()
    out = (x.cpu(), v.cpu(), f.cpu()); e.close(); return out



@pytest.mark.parametrize("scenario", ["lj", "water_nhc", "batch", "small"])
def test_fresh_handles_on_side_streams_match_the_default_stream(scenario):
    fn = {"lj": _sc_lj, "water_nhc": _sc_water_nhc, "batch": _sc_batch, "small": _sc_small}[scenario]
    ref = fn(torch.cuda.current_stream())
    assert all(torch.isfinite(r).all() for r in ref)
    for k in range(25):
        got = fn(torch.cuda.Stream())
        assert all(torch.equal(g, r) for g, r in zip(got, ref)), (scenario, k, [int(torch.isnan(g).sum()) for g in got])
