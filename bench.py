#!/usr/bin/env python3
"""bench.py — atom-steps/sec of the GAMD force path on MI355X (BASELINE.json metric).

One "step" = one MD step of one 10 000-atom LJ box (config C2): BAOAB first half ->
neighbour build + full GNN force evaluation -> BAOAB second half, all on device, inputs
resident in HBM.  N GPUs = N independent boxes (ensemble, weak scaling, no collective on
the step path).

`python bench.py --gpus N ...` launches the N ranks itself (one child process per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before this process touches any GPU) and
relays rank 0's line; under `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...` it joins the ranks torchrun started.  Either way it fails loudly when the
world size and --gpus disagree or when fewer than N devices are visible.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      live HIP-event timing of the dominant kernel (conv-layer edge kernel) over the timed
                region, plus a `kernels` list (edge encoder, node kernel, neighbour stage) from
                event-timed replays after it
  cpu_baseline  the CPU oracle (port of the reference's PyTorch path) timed on this host on the
                SAME 10 000-atom inputs and weights (N=1 only)
  secondary     short runs of the other single-GPU BASELINE configs (C1, C3, C5) (N=1 only)
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ATOMS = 10000
FLOP_PER_EDGE_CONV = 8 * 128 * 128          # 4 GEMMs 128x128 per edge per conv-edge launch
PEAK_FP32_MFMA_TFLOPS = 157.3               # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E spec peak
CPU_SAMPLE_ATOMS = 2000
C1_BATCH_BOXES = 38                         # 38 x 258 = 9 804 atoms: a C2-sized set of launches
SECONDARY_COMPACT = ("c1", "c3", "c5", "c5b", "dft")    # the default line's secondary triples; --secondary full runs twelve
LINE_LIMIT = 6000                           # characters of the ONE stdout line (the driver keeps an 8 081-character tail)
C2_BATCH_BOXES = 8                          # config 4's eight rank boxes (seeds 1234 .. 1241) as ONE batch on one GPU
# SURVEY.md 8d algorithmic FLOPs of one force evaluation (F = 44, L = 4, H = 128), src/dst Linears on node rows
FLOP_PER_EDGE_STEP = 2 * (44 * 128 + 2 * 128 * 128) + 4 * (8 * 128 * 128 + 4 * 128)      # 603 136
FLOP_PER_NODE_STEP = 4 * (10 * 128 * 128) + 2 * (128 * 128 + 3 * 128)                    # 688 896
# the files whose contents decide k_conv_edge's memory traffic: profiles/pmc_conv_edge.json is stamped with their hash
PMC_SOURCES = ("gamd_amd/csrc/conv_edge.hip", "gamd_amd/csrc/gamd_common.h", "gamd_amd/csrc/gamd_internal.h",
               "gamd_amd/csrc/neighbor.hip")


# ... and the conv-layer edge kernels whose gather traffic profiles/gather_hbm.json records at 10^5 / 10^6 atoms
GATHER_SOURCES = ("gamd_amd/csrc/conv_edge.hip", "gamd_amd/csrc/conv_edge_bf16.hip", "gamd_amd/csrc/gamd_common.h",
                  "gamd_amd/csrc/gamd_bf16.h", "gamd_amd/csrc/gamd_internal.h")


def _source_hash(files):
    h = hashlib.sha256()
    for rel in files:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def kernel_source_hash():
    return _source_hash(PMC_SOURCES)


def gather_source_hash():
    return _source_hash(GATHER_SOURCES)


def gather_hbm_block():
    """roofline.neighbour_gather_hbm: the conv-layer edge kernel's memory-side traffic (rocprofv3 counters) / kernel time /
    8 TB/s where the node tables it gathers from no longer fit the L2s / the Infinity Cache (10^5 - 10^6 atoms; at the
    BASELINE sizes they are L2-resident and the SURVEY's gather figure is algorithmic).  Measured by tools/gpu_pmc_gather.sh,
    kept in profiles/gather_hbm.json, reported only for the kernel sources it was taken on."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "gather_hbm.json")))
    except Exception as exc:
        return {"note": f"no record: {exc}"}
    if rec.get("kernel_source_sha256_16") != gather_source_hash():
        return {"note": f"profiles/gather_hbm.json was taken on other kernel sources ({rec.get('kernel_source_sha256_16')} != "
                        f"{gather_source_hash()}): not reported"}
    out = {k: {"atoms": v["n_atoms"], "edges": v["edges"], "kernel": v["kernel"], "kernel_ms": v["conv_ms_per_launch_live"],
               "counter_bytes_per_launch": v["hbm_bytes_per_launch"], "GB_per_s": v["hbm_GB_per_s"],
               "frac_of_hbm_peak": v["hbm_frac_of_8TBs"], "traffic_over_mandatory": v["hbm_over_mandatory"],
               "l2_hit_rate": v["l2_hit_rate"], "algorithmic_gather_frac": v["algorithmic_gather_frac_of_8TBs"]}
           for k, v in rec["records"].items()}
    out["source"] = rec["source"]
    out["bound"] = ("fp32: matrix-bound at every size (0.09 of the HBM peak); bf16: bound by each wave's chain of GEMM / SiLU / "
                    "gather-issue segments, per edge faster at 10^6 atoms than at the L2-resident C5 size (profiles/r05_gather_hbm.md)")
    return out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="same as --secondary none")
    ap.add_argument("--secondary", default="compact", choices=["none", "compact", "full"],
                    help="c2, N = 1 only.  compact (default): 20-step runs of the other single-GPU BASELINE configs (c1, c3, c5, "
                         "c5b) and of the DFT-water configuration, one [ms_per_step, atom-steps/s, conv-kernel frac] triple each in the "
                         "result line; full: all twelve secondary workloads with their per-step distributions, in the detail file only "
                         "(the line still carries the compact triples); none: skip them and the per-kernel replays")
    ap.add_argument("--line", default="compact", choices=["compact", "full"],
                    help="compact (default; the driver's contract): the ONE stdout line is the compact record.  full: print the "
                         "full record instead (tools/ only: the per-kernel list and per-step distributions on stdout)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full record goes (per-step distributions, per-kernel list, per-rank summaries, prose notes, "
                         "secondary workloads): the ONE line on stdout is the compact contract record, <= 6 000 characters; "
                         "'-' = do not write a detail file")
    ap.add_argument("--workload", default="c2", choices=["c1", "c1_batch", "c2", "c2_batch8", "c3", "c5", "c5b", "dft"],
                    help="c2 (default, the headline metric): 10k-atom LJ fp32; c1: the reference's own 258-atom LJ snapshot "
                         "(code/LJ/init_pos.npy); c1_batch: 38 such boxes in one set of launches (gamd_config.n_boxes); c2_batch8: the eight 10k-atom boxes the ranks of "
                         "BASELINE config 4 own (seeds 1234 .. 1241), as one n_boxes = 8 batch on ONE GPU; c3: 4 170-atom TIP3P fp32; c5: TIP4P-Ew-sized box of 2 000 molecules = 6 000 network "
                         "atoms, bf16 edge-MLP; c5b: the other reading of BASELINE config 5, 2 667 molecules = 8 001 network "
                         "atoms; dft: the 774-atom DFT-water configuration (256/256/128 x 5 layers, bohr, cutoff 9.5)")
    ap.add_argument("--skin", type=float, default=1.0 / 6.0,
                    help="Verlet-skin reuse of the neighbour candidates, in units of the cutoff (the reference's jax-md "
                         "list uses 1/6, graph_utils.py:24, and so does the default here); 0 = exact cell-list rebuild "
                         "every step.  The edge set is identical either way (not used by the dft workload, whose reference "
                         "searches from scratch every call)")
    ap.add_argument("--edge-dtype", default="f32", choices=["f32", "f16x3", "bf16"],
                    help="c2 / c3. f32 (default, the headline): fp32 MFMA, bit-exact fp32 FMAs.  f16x3: the same GEMMs on the "
                         "fp16 matrix pipe with every operand split into hi + lo fp16 (3 MFMAs per product term, fp32 "
                         "accumulate): fp32-grade results (same 1e-5 parity bar), 3/16 of the fp32 matrix time")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# N > 1 without torchrun: this process only spawns and relays; it never initialises a GPU
# ---------------------------------------------------------------------------------------------------------------
def spawn_ranks(args) -> int:
    """Start one child per rank, relay rank 0's result line, and never hang: every child is polled; as soon as one exits
    non-zero (or the whole run exceeds GAMD_BENCH_TIMEOUT_S, default 1800) the others — which would otherwise sit in a
    collective until the process-group timeout — are terminated (exact PIDs of our own children; fresh processes, none
    of them is ever re-exec'd) and this process exits non-zero."""
    import tempfile
    import torch                                   # device_count() does not initialise the GPU on this stack
    share = os.environ.get("GAMD_BENCH_SHARE_GPU", "0") == "1"
    ndev = torch.cuda.device_count()
    if not share and ndev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {ndev} HIP device(s) are visible "
              "(set GAMD_BENCH_SHARE_GPU=1 GAMD_BENCH_BACKEND=gloo for a control-flow dry run on one GPU)", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import signal
    from gamd_amd.ensemble import stop_ranks, supervise_ranks
    # a SIGTERM to this launcher (a driver's timeout) must not orphan the ranks: turn it into an exception so that the
    # cleanup below runs — also when it arrives while the ranks are still being started
    signal.signal(signal.SIGTERM, lambda *_: sys.exit(143))
    procs, out0 = [], tempfile.TemporaryFile(mode="w+")      # a file, not a pipe: rank 0 can never block on a full pipe
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
        failed = supervise_ranks(procs, timeout_s=float(os.environ.get("GAMD_BENCH_TIMEOUT_S", "1800")))
    finally:
        stop_ranks(procs)                                    # no-op when supervise_ranks returned normally
    if failed is not None:
        r, c = failed
        print(f"bench.py: {'the run timed out' if r < 0 else f'rank {r} exited with status {c}'}; "
              f"the remaining ranks were terminated", file=sys.stderr)
        return c if c > 0 else 1
    # rank 0's stdout carries the result line; anything else a library printed there (gloo announces its peers on
    # stdout) goes to stderr so that this process still prints exactly ONE line
    out0.seek(0)
    lines = [l for l in out0.read().splitlines() if l.strip()]
    result = [l for l in lines if l.lstrip().startswith('{"metric"')]
    for l in lines:
        if l not in result:
            print(l, file=sys.stderr)
    if len(result) != 1:
        print(f"bench.py: rank 0 printed {len(result)} result lines, expected one", file=sys.stderr)
        return 1
    print(result[0])
    return 0


# ---------------------------------------------------------------------------------------------------------------
# workloads (SURVEY.md §8d synthetic inputs)
# ---------------------------------------------------------------------------------------------------------------
class Workload:
    pass


def build_workload(name, ctx, dev, skin, edge_dtype):
    import numpy as np
    import torch
    from gamd_amd import ensemble as ens
    from gamd_amd import workloads as wk
    from gamd_amd.engine import GamdForce
    from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS

    w = Workload()
    w.name, w.species, w.mass, w.dtype_name = name, None, 39.9, "f32"
    w.md_extra, w.flop_per_edge, w.kernel_name = {}, FLOP_PER_EDGE_CONV, "k_conv_edge"
    w.dt_ps, w.uses_skin, w.n_boxes = 0.0005, False, 1
    if name == "dft":
        # water/test_script/test_nosehoover_hb.py:64-113: 258 molecules, (20 A)^3 box and positions in bohr, cutoff 9.5
        from gamd_amd.compat import HARTREE_PER_BOHR_TO_KJ_PER_MOL_NM as CONV
        bohr = wk.BOHR_PER_NM / 10.0
        pos, box, w.species, bonds = wk.water_box(258, seed=ens.box_seed(4567, ctx), jitter=0.0, wrap=False)
        pos, box = pos * bohr, box * bohr
        cfg = ModelConfig(kind="dynbox", encoding_size=256, hidden_dim=128, edge_embedding_dim=256, conv_layer=5)
        w.sd = make_state_dict(cfg, 5, 3.1 * bohr, 1.2 * bohr)
        mean, var = SHIPPED_SCALERS["dft"]
        w.eng = GamdForce(w.sd, pos.shape[0], box, 9.5, nbr_flavour="torch", cfg=cfg, device=dev, neighbor_skin=skin * 9.5,
                          scaler=(mean * CONV, var * CONV ** 2), edge_dtype=edge_dtype)   # hartree/bohr -> kJ/mol/nm folded into the scaler
        w.cutoff = 9.5
        w.uses_skin = skin > 0
        w.mass = wk.MASS_O
        w.md_extra = dict(mass_h_amu=wk.MASS_H, length_per_nm=wk.BOHR_PER_NM, rigid_water=True,
                          r_oh=wk.TIP3P_R_OH * bohr, r_hh=wk.TIP3P_R_HH * bohr)
        w.flop_per_edge, w.kernel_name = 2 * 128 * 128 * (2 + 2 + 2), "k_conv_edge_wide<2,2>"
        if edge_dtype == "f16x3":
            w.dtype_name, w.kernel_name = "f16x3 (fp32 operands split into hi+lo fp16, fp32 accumulate)", "k_conv_edge_f16x3_wide<2,2>"
        elif edge_dtype == "bf16":
            w.dtype_name, w.kernel_name = "bf16", "k_conv_edge_bf16_wide<2,2>"
        w.label = ("DFT-water configuration: 258 rigid molecules = 774 atoms, positions/box in bohr (L = 37.8), cutoff 9.5, "
                   f"WaterMDDynamicBoxNet widths 256/256/128, 5 conv layers, {dict(f16x3='split-fp16 edge MLP', bf16='bf16 edge-MLP operands / fp32 accumulate').get(edge_dtype, 'fp32')}, "
                   "random-init weights (seed 5), SETTLE on device, 1 box per GPU")
    elif name in ("c2", "c2_batch8", "c1", "c1_batch"):
        n = N_ATOMS if name in ("c2", "c2_batch8") else 258
        w.cutoff = 3.0 * wk.LJ_SIGMA if n == N_ATOMS else 7.5       # C1: CUTOFF_RADIUS of LJ/train_network_lj.py:26-29
        w.n_boxes = C1_BATCH_BOXES if name == "c1_batch" else C2_BATCH_BOXES if name == "c2_batch8" else 1
        if name == "c2":
            pos, box = wk.lj_box(n, seed=ens.box_seed(1234, ctx))
        elif name == "c2_batch8":
            # box b = the box rank b of `bench.py --gpus 8` owns: positions lj_box(seed 1234 + b), velocities seed 99 + b,
            # Langevin noise seed 7 + b (gamd_md_run's seed means seed + b for box b)
            boxes = [wk.lj_box(n, seed=1234 + b) for b in range(w.n_boxes)]
            pos, box = np.concatenate([p for p, _ in boxes]), boxes[0][1]
        else:
            # C1 = the reference's code/LJ/init_pos.npy verbatim (committed as the `pos` array of the golden
            # fixture oracle/make_golden.py wrote from it), BOX_SIZE 27.27.  c1_batch: box 0 is the snapshot, the others are
            # the snapshot plus N(0, 0.05 sigma) jitter (different microstates of the same system)
            snap = np.load(os.path.join(ROOT, "tests", "golden", "lj258_seed0.npz"))["pos"].astype(np.float64)
            box = 27.27
            rng = np.random.default_rng(ens.box_seed(4321, ctx))
            pos = np.concatenate([np.mod(snap + (rng.normal(0.0, 0.05 * wk.LJ_SIGMA, snap.shape) if b else 0.0), box)
                                  for b in range(w.n_boxes)])
        w.sd = make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2)
        w.eng = GamdForce(w.sd, n, box, w.cutoff, scaler=SHIPPED_SCALERS["lj"], device=dev,
                          neighbor_skin=skin * w.cutoff, edge_dtype=edge_dtype, n_boxes=w.n_boxes)
        w.uses_skin = skin > 0
        if edge_dtype == "f16x3":
            w.dtype_name, w.kernel_name = "f16x3 (fp32 operands split into hi+lo fp16, fp32 accumulate)", "k_conv_edge_f16x3"
        elif edge_dtype == "bf16":
            w.dtype_name, w.kernel_name = "bf16", "k_conv_edge_bf16"
        elif name == "c1":
            w.kernel_name = "k_conv_edge_small"       # <= small_tile_limit 32-edge tiles: one tile per 4-wave workgroup
        w.dt_ps = 0.002
        w.label = ("C2: 10 000-atom LJ box, rho*=0.5, L=92.29 A, cutoff 3.0 sigma=10.2 A, fp32, 4 conv layers x 128, "
                   "random-init weights (seed 0), 1 box per GPU") if name == "c2" else \
                  (f"C2 x {w.n_boxes}: the {w.n_boxes} independent 10 000-atom LJ boxes of BASELINE config 4 (box seeds 1234 .. "
                   f"{1233 + w.n_boxes}: what the ranks of --gpus {w.n_boxes} own) evaluated and integrated as ONE batch on one GPU "
                   "(gamd_config.n_boxes), same model and cutoff as C2") if name == "c2_batch8" else \
                  ("C1: the reference's 258-atom LJ snapshot (code/LJ/init_pos.npy), L=27.27 A, cutoff 7.5 A, fp32, "
                   "random-init weights (seed 0)") if name == "c1" else \
                  (f"C1 x {w.n_boxes}: {w.n_boxes} independent 258-atom LJ boxes (the reference's snapshot + per-box jitter) "
                   "evaluated and integrated in one set of launches (gamd_config.n_boxes), L=27.27 A, cutoff 7.5 A, fp32, "
                   "random-init weights (seed 0)")
    else:
        nmol, dens, scal, seed0 = {"c3": (1390, 258.0, "tip3p", 2345), "c5": (2000, 251.0, "tip4p", 3456),
                                   "c5b": (2667, 251.0, "tip4p", 3456)}[name]
        pos, box, w.species, bonds = wk.water_box(nmol, mol_per_20A3=dens, seed=ens.box_seed(seed0, ctx), jitter=0.0, wrap=False)
        w.sd = make_state_dict(ModelConfig(kind="water", use_bond=True), 3, 2.9, 1.1)
        w.dtype_name = "bf16" if name in ("c5", "c5b") else edge_dtype
        if w.dtype_name == "f16x3":
            w.kernel_name = "k_conv_edge_f16x3"
        w.cutoff = 4.2
        # the water drivers search neighbours with the same jax-md NeighborSearcher (dr_threshold = cutoff / 6) as the LJ
        # one (water/train_network_tip3p.py:100-118, graph_utils.py:21-25): same Verlet-skin reuse
        w.eng = GamdForce(w.sd, pos.shape[0], box, 4.2, bond=bonds, scaler=SHIPPED_SCALERS[scal], device=dev,
                          edge_dtype=w.dtype_name, neighbor_skin=skin * w.cutoff)
        w.uses_skin = skin > 0
        w.mass = wk.MASS_O
        w.md_extra = dict(mass_h_amu=wk.MASS_H, rigid_water=True, r_oh=wk.TIP3P_R_OH, r_hh=wk.TIP3P_R_HH)
        w.label = (f"{name.upper()}: {nmol} rigid water molecules (SETTLE on device) = {pos.shape[0]} network atoms, cutoff 4.2 A, "
                   f"bond feature, {'bf16 edge-MLP operands / fp32 accumulate' if w.dtype_name == 'bf16' else 'fp32'}, "
                   "random-init weights (seed 3), 1 box per GPU")
    w.n_atoms, w.box, w.pos = pos.shape[0], box, pos            # n_atoms: all boxes of this rank together
    w.x = torch.from_numpy(pos).float().cuda(dev)
    if name == "c2_batch8":
        vel = np.concatenate([wk.maxwell_boltzmann(N_ATOMS, mass_amu=w.mass, seed=99 + b) for b in range(w.n_boxes)])
    else:
        vel = wk.maxwell_boltzmann(w.n_atoms, mass_amu=w.mass, seed=99 + ctx.rank)
    w.v = torch.from_numpy(vel).float().cuda(dev)
    if name == "dft":
        w.v *= float(wk.BOHR_PER_NM / 10.0)
    w.f = w.eng.forward(w.x, species=w.species, denormalize=True)
    w.md = dict(dt_ps=w.dt_ps, mass_amu=w.mass, temperature_k=100.0, gamma_per_ps=25.0,
                seed=ens.box_seed(7, ctx), species=w.species, **w.md_extra)
    return w


def timed_run(w, steps, warmup, ctx, dev, ddev):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + synchronize on both sides;
    the conv-edge kernel is timed live with HIP events on the launch stream.  Returns (seconds, max over ranks,
    conv ms, conv launches)."""
    import torch
    from gamd_amd import ensemble as ens
    w.eng.md_run(w.x, w.v, w.f, warmup, first_step=0, **w.md)
    w.warmup_status = int(w.eng.last_status)
    w.eng.timing_enable(True)                      # creates its event pools here, outside the timed region
    torch.cuda.synchronize(dev)
    ens.barrier(ctx)
    torch.cuda.synchronize(dev)
    rebuilds0 = w.eng.skin_stats()[0] if w.uses_skin else 0
    t0 = time.perf_counter()
    w.eng.md_run(w.x, w.v, w.f, steps, first_step=warmup, sync=True, **w.md)
    w.timed_status = int(w.eng.last_status)        # 1: a neighbour buffer overflowed inside the timed run (frozen, regrown, resumed)
    torch.cuda.synchronize(dev)
    ens.barrier(ctx)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    w.stages = w.eng.timing_read_stages()          # live HIP events of the timed region, per stage
    w.step_ms = w.eng.timing_read_steps()          # one HIP event per MD step of the timed region
    w.rebuilds_in_timed = (w.eng.skin_stats()[0] - rebuilds0) if w.uses_skin else steps
    conv_ms, conv_n = w.stages["conv_edge"]
    w.eng.timing_enable(False)
    return dt, ens.max_over_ranks(dt, ctx, device=ddev), conv_ms, conv_n


def step_report(w, steps):
    """What the timed region was made of, so that the record can explain its own number: the per-step distribution (device
    time between the HIP events in front of consecutive MD steps), how many candidate-list rebuilds and buffer regrows fell
    inside it, and what one candidate rebuild costs (event-timed replays of the neighbour stage after the timed region,
    with and without a forced rebuild)."""
    import numpy as np
    s = np.sort(np.asarray(w.step_ms, dtype=np.float64))
    rep = {"step_ms": ({"min": float(s[0]), "p50": float(np.percentile(s, 50)), "p99": float(np.percentile(s, 99)),
                        "max": float(s[-1]), "mean": float(s.mean()), "intervals": int(s.size),
                        "definition": "device ms between HIP events recorded in front of the first kernel of consecutive MD steps"}
                       if s.size else None),
           "rebuilds_in_timed": int(w.rebuilds_in_timed),
           "regrown_in_timed": bool(getattr(w, "timed_status", 0) == 1),
           "regrown_in_warmup": bool(getattr(w, "warmup_status", 0) == 1)}
    if s.size and s.size != steps:
        rep["step_ms"]["note"] = (f"{s.size} intervals for {steps} steps: a run that froze on a buffer overflow contributes its "
                                  "frozen steps and the resumed ones separately")
    if w.uses_skin:
        def nbr_ms(force):
            acc = []
            for _ in range(3):
                if force:
                    w.eng.build_neighbors(w.x, species=w.species)      # an exact build invalidates the candidate list ...
                ms = dict(w.eng.profile(w.x, species=w.species))       # ... so this evaluation rebuilds it
                acc.append(ms.get("neighbor_build", float("nan")))
            return float(np.median(acc))
        reuse, rebuild = nbr_ms(False), nbr_ms(True)
        rep["neighbour_stage_ms"] = {"reuse_step": reuse, "rebuild_step": rebuild}
        rep["rebuild_ms"] = rebuild - reuse
    return rep


def stage_replays(w, reps=8):
    """Event-timed replays of one force evaluation (gamd_profile): average ms per stage label."""
    acc, cnt = {}, {}
    for _ in range(reps):
        for label, ms in w.eng.profile(w.x, species=w.species):
            acc[label] = acc.get(label, 0.0) + ms
            cnt[label] = cnt.get(label, 0) + 1
    return {k: acc[k] / cnt[k] for k in acc}, {k: cnt[k] // reps for k in cnt}


def cpu_baseline(w, dev):
    """The oracle's forward (op-for-op port of nn_module.py: unfused nn.Linear on E gathered rows, index_add_
    aggregation) on the host cores.  torch's CPU kernels stop scaling long before this host's core count, so the
    thread count is chosen on a 2 000-atom sample (best of 8/16/32) and the reported figure is then measured on the
    timed run's own 10 000-atom inputs and weights: 1 warm-up + 2 evaluations.  Checker code, used
    as a *reported baseline only*; the GPU result on the same inputs is compared with it."""
    import numpy as np
    import torch
    from gamd_amd.engine import GamdForce
    from gamd_amd.workloads import lj_box
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gamd_oracle as orc
    ncpu = os.cpu_count() or 1
    # --- thread count: bounded sample -------------------------------------------------------------------------
    pos_s, box_s = lj_box(CPU_SAMPLE_ATOMS, seed=4321)
    p_s = torch.from_numpy(pos_s).float()
    edges_s = orc.neighbor_edges(p_s, box_s, w.cutoff, "jaxmd")
    best_s, best_thr, out_s = None, None, None
    for thr in sorted({min(ncpu, t) for t in (8, 16, 32)}):
        torch.set_num_threads(thr)
        orc.forward(w.sd, p_s, edges_s, box_s)                  # warm-up
        t0 = time.perf_counter()
        out_s = orc.forward(w.sd, p_s, edges_s, box_s)
        dt = time.perf_counter() - t0
        if best_s is None or dt < best_s:
            best_s, best_thr = dt, thr
    eng_s = GamdForce(w.sd, CPU_SAMPLE_ATOMS, box_s, w.cutoff, device=dev)
    gpu_s = eng_s.forward(p_s).cpu().numpy()
    same_edges_s = eng_s.counts()[0] == edges_s.shape[1]
    eng_s.close()
    err_s = float(np.abs(gpu_s - out_s.numpy()).max() / np.abs(out_s.numpy()).max())
    # --- the headline inputs: current positions of the timed box, its weights, the GPU's own edge list ---------------
    torch.set_num_threads(best_thr)
    x_host = w.x.detach().cpu()
    gpu = w.eng.forward(w.x).cpu().numpy()
    edges = torch.from_numpy(w.eng.debug_edges()).long()
    xw = torch.remainder(x_host, float(w.box))
    orc.forward(w.sd, xw, edges, w.box)                         # warm-up
    times = []
    for _ in range(2):
        t0 = time.perf_counter()
        out = orc.forward(w.sd, xw, edges, w.box)
        times.append(time.perf_counter() - t0)
    err = float(np.abs(gpu - out.numpy()).max() / np.abs(out.numpy()).max())
    # per-atom statistic next to the max-norm figure: |df_i| / |f_i| over the atoms whose force exceeds 1e-3 of the largest
    ref64 = out.numpy().astype(np.float64)
    nrm = np.linalg.norm(ref64, axis=1)
    keep = nrm > 1e-3 * nrm.max()
    rel_i = np.linalg.norm(gpu.astype(np.float64) - ref64, axis=1)[keep] / nrm[keep]
    sec = min(times)
    return {"value": w.n_atoms / sec, "unit": "atom-steps/s", "cores": best_thr, "kind": "port",
            "sample": f"force evaluation of the timed run's own {w.n_atoms}-atom LJ box (positions after the timed steps, "
                      f"same weights, {edges.shape[1]} edges from the GPU's neighbour list), 1 warm-up + 2 evaluations "
                      f"(best) with {best_thr} torch threads on a {ncpu}-thread host (os.cpu_count()); neighbour search "
                      "and integrator excluded",
            "sample_short": f"force eval of the timed run's own {w.n_atoms}-atom box ({edges.shape[1]} edges), 1 warm-up + 2 evals (best), "
                            f"{best_thr} torch threads of {ncpu}",
            "seconds_per_eval": sec, "seconds_all": times, "host_threads": ncpu, "gpu_vs_cpu_rel_err": err,
            "gpu_vs_cpu_per_atom": {"median": float(np.median(rel_i)), "p99": float(np.percentile(rel_i, 99)),
                                    "max": float(rel_i.max()), "atoms": int(keep.sum()),
                                    "definition": "|f_gpu_i - f_cpu_i| / |f_cpu_i| over atoms with |f_cpu_i| > 1e-3 max|f_cpu|"},
            "sample_2000": {"value": CPU_SAMPLE_ATOMS / best_s, "seconds_per_eval": best_s, "edges": int(edges_s.shape[1]),
                            "gpu_vs_cpu_rel_err": err_s, "same_edge_count": bool(same_edges_s),
                            "note": "2 000-atom box of the same density, cutoff and weights: thread-count scan "
                                    "(best of 8/16/32) and linearity cross-check"}}


def roofline_block(w, n_edges, conv_ms, conv_n):
    avg_ms = conv_ms / max(conv_n, 1)
    achieved = n_edges * w.flop_per_edge / (avg_ms * 1e-3) / 1e12
    r = {"kernel": w.kernel_name, "bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
         "avg_launch_ms": avg_ms, "launches": conv_n, "flop_per_launch": n_edges * w.flop_per_edge}
    if w.dtype_name.startswith("f16x3"):
        # every product term costs 3 fp16 MFMAs (Wh xh, Wh xl, Wl xh): the matrix pipe executes 3x the algorithmic FLOPs,
        # against the dense fp16 MFMA peak; the kernel is bound by LDS operand feed + VALU, not by that peak
        r.update({"peak": 2500.0, "frac": achieved / 2500.0, "executed_mfma_tflops": 3.0 * achieved,
                  "note": "achieved = algorithmic fp32-equivalent FLOP/s; peak = dense fp16 MFMA"})
    if w.dtype_name == "bf16":
        # SURVEY.md §8d "neighbour gather" figure (per edge: 4 B index + 512 B h[src] row + 512 B S[src] row) against HBM
        # peak.  These are ALGORITHMIC gather bytes: the rows are served from L2 (node tables are 3 MB), the HBM traffic
        # of the kernel is far lower; the kernel itself is VALU/MFMA-issue bound
        hn_row = 512.0 * (w.flop_per_edge // (2 * 128 * 128) - 2) / 2 if w.name == "dft" else 512.0     # 256-wide models: 1 KiB hn rows
        gbytes = n_edges * (4.0 + hn_row + 512.0)
        r = {"kernel": w.kernel_name if w.kernel_name.startswith("k_conv_edge_bf16") else "k_conv_edge_bf16", "bound": "hbm", "achieved": gbytes / (avg_ms * 1e-3) / 1e9,
             "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbytes / (avg_ms * 1e-3) / (PEAK_HBM_GBS * 1e9), "traffic": None,
             "avg_launch_ms": avg_ms, "launches": conv_n, "bytes_per_launch": gbytes,
             "note": "algorithmic neighbour-gather bytes (L2-served), not HBM traffic"}
    return r


def _sig(x, n=6):
    """Numbers of the result line: n significant digits (the full-precision values are in the detail file)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    return float(f"{float(x):.{n}g}")


def rocprof_record():
    """roofline.rocprof_avg_launch_ms: the rocprofv3 --kernel-trace --stats average of the dominant kernel, taken by
    tools/gpu_profile_round.sh with the same command and kept in profiles/pmc_conv_edge.json next to the PMC traffic; reported
    only for the kernel sources it was measured on (hash), otherwise null."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_conv_edge.json")))
    except Exception:
        return None
    return rec if rec.get("kernel_source_sha256_16") == kernel_source_hash() else None


def compact_line(d, args):
    """The ONE stdout line: the contract keys in the contract's order, numbers only (<= LINE_LIMIT characters).  Everything
    explanatory — notes, per-kernel list, per-rank summaries, per-step distributions of the secondary workloads — is in the
    detail file."""
    cfg, rl = d["config"], d["roofline"]
    tr = cfg["timed_region"]
    sm = tr.get("step_ms") or {}
    out = {k: (_sig(d[k]) if k in ("value", "ms_per_step") else d[k]) for k in
           ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")}
    out["config"] = {"workload": cfg["workload"], "n_atoms": cfg["n_atoms"], "edges_per_step": cfg["edges_per_step"],
                     "boxes": cfg["boxes"], "neighbour_list": cfg["neighbour_list"], "launch": cfg["launch"],
                     "timed_region": {"p50": _sig(sm.get("p50"), 4), "p99": _sig(sm.get("p99"), 4), "max": _sig(sm.get("max"), 4),
                                      "mean": _sig(sm.get("mean"), 4), "intervals": sm.get("intervals"),
                                      "rebuilds_in_timed": tr["rebuilds_in_timed"], "regrown_in_timed": tr["regrown_in_timed"],
                                      "regrown_in_warmup": tr["regrown_in_warmup"], "rebuild_ms": _sig(tr.get("rebuild_ms"), 3)}}
    r = {k: _sig(rl.get(k)) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches")}
    r["flop_per_launch" if "flop_per_launch" in rl else "bytes_per_launch"] = _sig(rl.get("flop_per_launch", rl.get("bytes_per_launch")))
    for k in ("whole_step_frac", "rocprof_avg_launch_ms", "rocprof_frac"):
        if k in rl:
            r[k] = _sig(rl[k])
    if "neighbour_gather" in rl:
        r["neighbour_gather"] = {k: _sig(rl["neighbour_gather"][k], 4) for k in ("bytes_per_step", "GB_per_s", "frac_of_hbm_peak")}
    if "kernels" in rl:                                     # [avg launch ms, frac of the fp32 matrix peak] of the other kernels
        r["other_kernels"] = {k["kernel"].split(" ")[0]: [_sig(k.get("avg_launch_ms", k.get("avg_stage_ms")), 4), _sig(k.get("frac"), 3)]
                              for k in rl["kernels"]}
    out["roofline"] = r
    if "cpu_baseline" in d:
        cb = d["cpu_baseline"]
        out["cpu_baseline"] = {"value": _sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                               "sample": cb.get("sample_short") or cb["sample"][:160], "host_threads": cb["host_threads"],
                               "seconds_per_eval": _sig(cb["seconds_per_eval"], 4), "gpu_vs_cpu_rel_err": _sig(cb["gpu_vs_cpu_rel_err"], 3),
                               "gpu_vs_cpu_per_atom_p99": _sig(cb["gpu_vs_cpu_per_atom"]["p99"], 3)}
    if "force_eval_only" in d:
        fe = d["force_eval_only"]
        out["force_eval_only"] = {"ms_per_eval": _sig(fe["ms_per_eval"], 4), "value": _sig(fe["value"]), "unit": fe["unit"]}
    en = d["ensemble"]
    out["ensemble"] = {"boxes": en["boxes"], "distinct_devices": en["distinct_devices"],
                       "collective_on_step_path": en["collective_on_step_path"],
                       "rank_seconds": [_sig(x["seconds"], 4) for x in en["per_rank"]]}
    if "secondary" in d:                                    # name -> [ms per step, atom-steps/s, conv-kernel roofline frac]
        out["secondary"] = {k: [_sig(v["ms_per_step"], 4), _sig(v["value"], 4), _sig(v["conv_kernel"]["frac"], 3)]
                            for k, v in d["secondary"].items() if k in SECONDARY_COMPACT}
        out["secondary_unit"] = "[ms_per_step, atom-steps/s, conv-kernel roofline frac], 20 steps each"
    if args.detail != "-":
        out["detail_file"] = os.path.basename(args.detail)
    return out


def main():
    args = parse_args()
    if args.no_secondary:
        args.secondary = "none"
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import numpy as np
    import torch
    from gamd_amd import ensemble as ens

    # RCCL ("nccl") over xGMI in production; GAMD_BENCH_BACKEND=gloo + GAMD_BENCH_SHARE_GPU=1 let the N>1 control
    # flow be dry-run with several ranks on a single-GPU box (the timing of such a run means nothing)
    backend = os.environ.get("GAMD_BENCH_BACKEND", "nccl")
    share = os.environ.get("GAMD_BENCH_SHARE_GPU", "0") == "1"
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: launch with matching values "
                         "(or without torchrun: bench.py --gpus N starts its own ranks)")
    local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    if not share and torch.cuda.device_count() <= local:
        raise SystemExit(f"bench.py: rank with LOCAL_RANK={local} has no device ({torch.cuda.device_count()} visible)")
    ctx = ens.init_ensemble(backend, device_index=0 if share else None,
                            timeout_s=float(os.environ.get("GAMD_BENCH_PG_TIMEOUT_S", "120")))
    dev = 0 if share else (ctx.local_rank if ctx.distributed else 0)
    torch.cuda.set_device(dev)
    ddev = f"cuda:{dev}" if (ctx.distributed and backend == "nccl") else "cpu"
    host_threads = ens.pin_host_threads(ctx) if ctx.distributed else torch.get_num_threads()
    # fault injection for tests/test_gpu_bench_contract.py: this rank dies after the rendezvous, before the first barrier
    if os.environ.get("GAMD_BENCH_FAIL_RANK", "") == str(ctx.rank) and ctx.distributed:
        print(f"bench.py: rank {ctx.rank} failing on purpose (GAMD_BENCH_FAIL_RANK)", file=sys.stderr)
        os._exit(3)

    # (the dft workload's reference, md_module.get_neighbor, searches from scratch every call; the Verlet-skin path yields
    # the same edge SET — tests/test_gpu_round4.py — so the MD loop uses it there too)
    skin = args.skin
    w = build_workload(args.workload, ctx, dev, skin, args.edge_dtype)
    dt, dt_max, conv_ms, conv_n = timed_run(w, args.steps, args.warmup, ctx, dev, ddev)
    n_edges = w.eng.counts()[0]
    summary = ens.gather_summary({"seconds": dt, "edges": float(n_edges), "fsum": float(w.f.abs().sum().item()),
                                  "box_seed": float(ens.box_seed(1234, ctx)), "host_threads": float(host_threads),
                                  **ens.device_identity(dev),
                                  "finite": float(torch.isfinite(w.x).all().item() and torch.isfinite(w.f).all().item())},
                                 ctx, device=ddev)
    if ctx.rank != 0:
        ens.shutdown(ctx)
        return
    if not all(s["finite"] == 1.0 for s in summary):
        raise SystemExit("non-finite state after the timed run")

    value = ens.aggregate_throughput(w.n_atoms * args.steps, dt_max, ctx)
    line = {
        "metric": "atom-steps/sec (force eval + integrate), 10k-atom LJ box" if args.workload == "c2"
                  else f"atom-steps/sec (force eval + integrate), {args.workload} box",
        "value": value, "unit": "atom-steps/s", "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": w.dtype_name, "data": "synthetic",
        "config": {"workload": w.label, "n_atoms": w.n_atoms, "edges_per_step": n_edges, "boxes": ctx.world,
                   "neighbour_list": (f"Verlet skin {skin:.3f} x cutoff, exact re-filter every step, "
                                      f"{w.eng.skin_stats()[0]} candidate rebuilds in warm-up + timed steps"
                                      if w.uses_skin else "exact cell-list rebuild every step"),
                   "step": "BAOAB half + neighbour build + GNN forces + BAOAB half, on device",
                   "buffers_regrown_in_timed_run": bool(getattr(w, "timed_status", 0) == 1),
                   "timed_region": step_report(w, args.steps),
                   "launch": "torch.distributed.run" if os.environ.get("TORCHELASTIC_RUN_ID") else
                             ("self-spawned ranks" if ctx.world > 1 else "single process")},
        "ensemble": {"boxes": ctx.world, "collective_on_step_path": False,
                     "distinct_devices": len({ens.pci_string(s) for s in summary}),
                     "per_rank": [{"rank": r, "box_seed": int(s["box_seed"]), "seconds": s["seconds"], "edges": int(s["edges"]),
                                   "force_abs_sum": s["fsum"], "device": int(s["device"]), "pci_bus_id": ens.pci_string(s),
                                   "group_world_size": int(s["group_world_size"]), "host_threads": int(s["host_threads"])}
                                  for r, s in enumerate(summary)]},
        "roofline": roofline_block(w, n_edges, conv_ms, conv_n),
    }
    rl = line["roofline"]
    # live HIP events of the timed region per stage: average ms per launch and launches (detail record; tools/ab_libs.sh)
    line["stages_ms"] = {k: [v[0] / max(v[1], 1), v[1]] for k, v in w.stages.items()}
    if args.workload == "c2" and w.dtype_name == "f32":
        # HBM bytes per launch from the PMC passes (profiles/), valid only for the kernel sources they were taken on
        pmc = os.path.join(ROOT, "profiles", "pmc_conv_edge.json")
        try:
            rec = json.load(open(pmc))
            if rec.get("kernel_source_sha256_16") == kernel_source_hash():
                rl["traffic"] = rec.get("hbm_bytes_per_launch")
                rl["traffic_source"] = "profiles/pmc_conv_edge.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, same kernel sources)"
            else:
                rl["traffic_note"] = ("profiles/pmc_conv_edge.json was taken on other kernel sources "
                                      f"({rec.get('kernel_source_sha256_16')} != {kernel_source_hash()}): not reported")
        except Exception as exc:                                        # missing / unreadable file: say so, report null
            rl["traffic_note"] = f"no PMC record: {exc}"
        # the rocprofv3 --kernel-trace --stats average of the same kernel sources (profiles/), beside the live HIP-event figure
        rp = rocprof_record()
        if rp is not None and rp.get("rocprof_avg_launch_us"):
            rl["rocprof_avg_launch_ms"] = rp["rocprof_avg_launch_us"] * 1e-3
            rl["rocprof_frac"] = rl["flop_per_launch"] / (rp["rocprof_avg_launch_us"] * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS
            rl["rocprof_source"] = rp.get("rocprof_source")
        else:
            rl["rocprof_avg_launch_ms"] = None
        rl["ceiling_note"] = ("gfx950 issues fp32 MFMA and VALU on the same lanes (simple VALU 4 cycles, v_exp_f32 / v_rcp_f32 8, next to "
                              "64 per v_mfma_f32_32x32x2_f32). Timing ablations of this kernel at this size (profiles/"
                              "r03_conv_edge_experiments.md): the GEMM chain alone (LDS-fed MFMAs, barriers, piece stores) 0.90 of the "
                              "matrix peak, of which 3 % is the launch tail (9.69 tiles per wave -> 10 rounds); the element-wise "
                              "post-ops (3 x 64 SiLU, S add, message / segment sum per 32-edge tile) cost 5-7 %, weight copies 1 %, "
                              "accumulator initialisation 0.8 %; gathers, barrier skew and LDS latency nothing")
        # SURVEY.md §8d "neighbour gather" figure: L2-served rows, so this is not HBM traffic; stated with its bound
        # the whole MD step against the fp32 matrix peak: SURVEY.md 8d algorithmic FLOPs of one force evaluation
        step_flop = FLOP_PER_EDGE_STEP * float(n_edges) + FLOP_PER_NODE_STEP * float(w.n_atoms)
        rl["whole_step_frac"] = step_flop / (line["ms_per_step"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS
        rl["whole_step_note"] = ("(603 136 E + 688 896 N) FLOP per step / ms_per_step / fp32 matrix peak: neighbour stage, "
                                 "encoder, 4 conv layers, node kernels, decoder and integrator together")
        n_layers = 4
        gather_bytes = n_layers * (n_edges * 1028.0 + w.n_atoms * 1024.0)
        gather_s = rl["avg_launch_ms"] * 1e-3 * n_layers
        rl["neighbour_gather"] = {"bytes_per_step": gather_bytes, "GB_per_s": gather_bytes / gather_s / 1e9,
                                  "frac_of_hbm_peak": gather_bytes / gather_s / (PEAK_HBM_GBS * 1e9),
                                  "bound": "mfma-bound in fp32: the gather rides inside k_conv_edge, whose time is set by "
                                           "the fp32 matrix pipe; the north star's >= 50 % of the memory roofline on the "
                                           "gather is reachable only by the bf16 kernel (see secondary c5)"}
        rl["neighbour_gather_hbm"] = gather_hbm_block()
    single = ctx.world == 1
    if single and args.secondary != "none":
        # the other MFMA kernels, from the same live HIP events of the timed region as the conv kernel: the edge encoder has
        # its own event pair; the node kernel between two conv layers is the interval between their event pairs (both
        # kernel boundaries included).  The neighbour stage comes from event-timed replays afterwards.
        F = 45 if w.species is not None else 44
        kern = []
        enc_ms, enc_n = w.stages["edge_encode"]
        node_ms, node_n = w.stages["node_mid"]
        if enc_n and w.dtype_name == "f32" and args.workload in ("c1", "c2", "c3"):
            fl, t = n_edges * 2.0 * (F * 128 + 2 * 128 * 128), enc_ms / enc_n
            kern.append({"kernel": "k_edge_encode", "bound": "mfma", "flop_per_launch": fl, "avg_launch_ms": t, "launches": enc_n,
                         "achieved": fl / (t * 1e-3) / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": fl / (t * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, "launches_per_step": 1})
        if node_n and args.workload in ("c1", "c2", "c3", "c5", "c5b"):
            fl, t = w.n_atoms * 10.0 * 128 * 128, node_ms / node_n
            kern.append({"kernel": "k_node (post + pre between two conv layers)", "bound": "latency (16-atom tiles, 5 chained GEMMs)",
                         "flop_per_launch": fl, "avg_launch_ms": t, "launches": node_n,
                         "achieved": fl / (t * 1e-3) / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": fl / (t * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         "launches_per_step": 5, "node_ms_per_step_estimate": 5 * t,
                         "note": "interval between the conv layers' event pairs: the kernel plus its two boundaries (~3 us)"})
        ms, per_eval = stage_replays(w, reps=4)
        if "neighbor_build" in ms:
            kern.append({"kernel": "neighbour stage (skin check + exact filter | cell-list build, CSR, chunk metadata)",
                         "bound": "latency / hbm (tiny)", "avg_stage_ms": ms["neighbor_build"],
                         "bytes_algorithmic": 12.0 * w.n_atoms + 8.0 * n_edges})
        rl["kernels"] = kern
        rl["kernels_note"] = "k_edge_encode / k_node: live HIP events over the timed region; neighbour stage: 4 event-timed replays"
    if single:
        # "report also force-eval-only atom-evals/s" — the call the rollout drivers make once per step
        # (predict_forces: neighbour stage + GNN, no integrator), synchronous like theirs, on the state the timed run left
        # behind; device-resident positions (the PCIe-inclusive form of the call is in DESIGN.md section 5)
        reps = 20
        w.eng.forward(w.x, species=w.species, inplace=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            w.eng.forward(w.x, species=w.species, inplace=True)
        torch.cuda.synchronize(dev)
        fe = (time.perf_counter() - t0) / reps
        line["force_eval_only"] = {"ms_per_eval": fe * 1e3, "value": w.n_atoms / fe, "unit": "atom-evals/s", "evals": reps,
                                   "what": "gamd_forces: neighbour stage (" + ("Verlet-skin reuse" if w.uses_skin else "exact rebuild")
                                           + ") + edge encoder + conv layers + decoder, one synchronous call per evaluation, "
                                           "positions resident on the device, no integrator"}
    if single and not args.no_cpu_baseline and args.workload == "c2":
        line["cpu_baseline"] = cpu_baseline(w, dev)
    if single and args.secondary != "none" and args.workload == "c2":
        sec = {}
        w.eng.close()
        # (entry, workload, edge dtype): the other single-GPU BASELINE configs, the batched C1, the opt-in split-fp16 and bf16
        # runs of C2 (c2_bf16: the north star's neighbour-gather figure on the 10k-atom LJ box itself, tolerance restated as
        # for config 5) and the DFT-water configuration, 20 timed steps each
        full = (("c1", "c1", "f32"), ("c1_batch", "c1_batch", "f32"), ("c1_batch_f16x3", "c1_batch", "f16x3"),
                ("c2_batch8", "c2_batch8", "f32"), ("c3", "c3", "f32"), ("c5", "c5", "f32"),
                ("c5b", "c5b", "f32"), ("c2_f16x3", "c2", "f16x3"), ("c2_bf16", "c2", "bf16"), ("dft", "dft", "f32"),
                ("dft_f16x3", "dft", "f16x3"), ("dft_bf16", "dft", "bf16"))
        for name, wname, dt_name in (full if args.secondary == "full" else [t for t in full if t[0] in SECONDARY_COMPACT]):
            s = build_workload(wname, ctx, dev, args.skin, dt_name)
            sdt, _, sconv_ms, sconv_n = timed_run(s, 20, 5, ctx, dev, ddev)
            first_attempt = None
            if len(s.step_ms) and float(np.max(s.step_ms)) > 5.0 * float(np.median(s.step_ms)) + 0.5:
                # one step of this short run took several times the median (and more than any rebuild costs): something
                # other than the workload landed in the region (round 4's driver record had one such entry).  Keep what was
                # measured, say so, and measure once more
                first_attempt = {"ms_per_step": sdt / 20 * 1e3, "timed_region": step_report(s, 20)}
                sdt, _, sconv_ms, sconv_n = timed_run(s, 20, 0, ctx, dev, ddev)
            se = s.eng.counts()[0]
            ok = bool(torch.isfinite(s.x).all().item() and torch.isfinite(s.f).all().item())
            rb = roofline_block(s, se, sconv_ms, sconv_n)
            sec[name] = {"workload": s.label, "n_atoms": s.n_atoms, "n_boxes": s.n_boxes, "edges_per_step": se, "dtype": s.dtype_name,
                         "steps": 20, "warmup": 5, "ms_per_step": sdt / 20 * 1e3, "value": s.n_atoms * 20 / sdt,
                         "unit": "atom-steps/s", "finite": ok, "timed_region": step_report(s, 20),
                         "conv_kernel": {k: rb[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms")}}
            if first_attempt is not None:
                sec[name]["stall_detected_in_first_attempt"] = first_attempt
            s.eng.close()
        line["secondary"] = sec
    out = compact_line(line, args)
    text = json.dumps(out, allow_nan=False, separators=(",", ":"))
    if len(text) > LINE_LIMIT:
        raise SystemExit(f"bench.py: the result line has {len(text)} characters (limit {LINE_LIMIT}): move fields to the detail file")
    if args.detail != "-":
        try:
            with open(args.detail, "w") as f:
                json.dump({"line": out, "detail": line}, f)
            print(f"bench.py: full record (per-step distributions, per-kernel list, per-rank summaries, notes, secondary "
                  f"workloads) written to {args.detail}", file=sys.stderr)
        except OSError as exc:
            print(f"bench.py: detail file not written: {exc}", file=sys.stderr)
    print(text if args.line == "compact" else json.dumps(line))
    ens.shutdown(ctx)


if __name__ == "__main__":
    main()
