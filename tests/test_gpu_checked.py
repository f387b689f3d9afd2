"""The checked build (`make checked` -> gamd_amd/libgamd_hip_chk.so, -DGAMD_CHECKED: SURVEY.md section 5 "add a debug build with
bounds checks") and a stress loop sized to the one real bug this code base has had.

* representative GPU tests — one or two per kernel family, the regrow protocol, skin rebuilds, batches, side streams — run in
  a child process with GAMD_LIB pointing at the checked library: every index a kernel reads from memory and then uses as an
  address passes a device-side range check there (GAMD_CHK_RANGE, gamd_amd/csrc/gamd_internal.h); a violation makes the
  synchronous entry points return -35, i.e. the child's tests fail;
* the reporting chain itself: with GAMD_CHK_INJECT set the checked library hands the conv kernels a node-table size of zero,
  and the call must come back as -35 naming the check — the release library ignores the variable;
* round 5's race (a NULL-stream memset wiping the first results of a kernel on the caller's non-blocking stream) showed once in
  ~300 two-stream runs and was found by luck: 300 iterations of fresh handles driven from two non-blocking streams, every one
  compared with the bits the handles give alone.
"""
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

import gamd_oracle as orc
from gamd_amd import workloads
from gamd_amd.engine import GamdForce
from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS
from helpers import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHK = os.path.join(ROOT, "gamd_amd", "libgamd_hip_chk.so")

# one or two per kernel family (fp32 throughput / small / generic-width / bf16 / split-fp16 conv and encoder kernels, node kernels,
# both neighbour flavours, exact build, skin reuse + candidate rebuild, regrow also inside an enqueued run, batches, integrators,
# fresh handles on side streams)
REPRESENTATIVE = [
    "tests/test_gpu_parity.py::test_golden_stages_and_forces",
    "tests/test_gpu_parity.py::test_neighbor_sets_match_oracle",
    "tests/test_gpu_parity.py::test_wide_and_unexpanded_configs_match_reference_golden",
    "tests/test_gpu_parity.py::test_isolated_atoms_and_regrow",
    "tests/test_gpu_parity.py::test_c5_bf16_edge_mlp_against_fp32_path_and_oracle",
    "tests/test_gpu_parity.py::test_tiny_and_crowded_systems",
    "tests/test_gpu_parity.py::test_verlet_skin_in_the_md_loop_and_candidate_regrow",
    "tests/test_gpu_parity.py::test_split_fp16_at_c2_size_against_the_fp32_path",
    "tests/test_gpu_round3.py::test_padding_slots_of_the_last_tile_over_many_remainders",
    "tests/test_gpu_batch.py::test_batch_overflow_is_regrown_also_in_the_middle_of_an_md_run",
    "tests/test_gpu_batch.py::test_batch_rigid_water_and_nose_hoover_chains_per_box",
    "tests/test_gpu_round4.py::test_reduced_precision_kernels_on_sparse_tiny_and_overflowing_inputs",
    "tests/test_gpu_round4.py::test_update_edge_emb_matches_the_reference_goldens",
    "tests/test_gpu_hidden256.py::test_reference_goldens_stage_by_stage",
    "tests/test_gpu_lifecycle.py::test_fresh_handles_on_side_streams_match_the_default_stream",
    "tests/test_gpu_lifecycle.py::test_candidate_rebuild_by_sliced_workgroups_sorts_like_the_exact_build",
]


def _child_env(**extra):
    e = {k: v for k, v in os.environ.items() if k not in ("GAMD_LIB", "GAMD_CHK_INJECT")}
    e.update(extra)
    return e


def test_checked_library_exists_and_says_so():
    assert os.path.exists(CHK), "gamd_amd/libgamd_hip_chk.so is missing: `make -C gamd_amd/csrc checked` (build() does it)"
    p = subprocess.run([sys.executable, "-c", "from gamd_amd import _lib; print(_lib.load().gamd_version().decode())"],
                       capture_output=True, text=True, cwd=ROOT, env=_child_env(GAMD_LIB=CHK), timeout=300)
    assert p.returncode == 0 and p.stdout.strip().endswith("checked"), (p.stdout, p.stderr[-500:])


def test_representative_gpu_tests_are_green_under_the_checked_library():
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", *REPRESENTATIVE],
                       capture_output=True, text=True, cwd=ROOT, env=_child_env(GAMD_LIB=CHK), timeout=1500)
    tail = p.stdout[-1500:]
    assert p.returncode == 0, tail + p.stderr[-1500:]
    assert " passed" in tail and "failed" not in tail and "error" not in tail.lower(), tail
    print(tail.strip().splitlines()[-1])


INJECT = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
from helpers import load_golden
from gamd_amd.engine import GamdForce
from gamd_amd._lib import GamdError
g, cfg, sd = load_golden("lj258_seed0")
box = float(g["box"])
eng = GamdForce(sd, 258, box, float(g["cutoff"]), scaler=(g["scaler_mean"], g["scaler_var"]))
x = torch.from_numpy(np.mod(g["pos"], box)).float()
try:
    out = eng.forward(x).cpu().numpy()
    print("NO_ERROR", float(np.abs(out - g["out_norm"]).max() / np.abs(g["out_norm"]).max()))
except GamdError as exc:
    print("GAMD_ERROR", exc)
"""


def test_an_injected_violation_comes_back_as_minus_35_and_only_from_the_checked_library():
    code = INJECT % (ROOT, os.path.join(ROOT, "tests"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                       env=_child_env(GAMD_LIB=CHK, GAMD_CHK_INJECT="1"), timeout=600)
    assert p.returncode == 0, p.stderr[-1500:]
    assert "GAMD_ERROR" in p.stdout and "status -35" in p.stdout and "range check 121" in p.stdout, p.stdout[-800:]
    # the same library without the injection, and the release library with it: no error, the golden's forces
    for env in (_child_env(GAMD_LIB=CHK), _child_env(GAMD_CHK_INJECT="1")):
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
        assert p.returncode == 0 and "NO_ERROR" in p.stdout, (p.stdout[-500:], p.stderr[-800:])
        assert float(p.stdout.split("NO_ERROR")[1].split()[0]) < 1e-5


def test_300_two_stream_iterations_of_fresh_handles_match_the_handles_alone():
    """Fresh handles every iteration (the race sat in what a handle's FIRST calls allocate and zero), an LJ box with COM removal
    and a bf16 rigid-water box, MD runs enqueued alternately on two non-blocking streams with nothing synchronised in between;
    every iteration must give the bits the two handles give alone on the default stream.  Budget ~60 s."""
    n_lj = 1500
    pos, box = workloads.lj_box(n_lj, seed=77)
    rc = 3.0 * workloads.LJ_SIGMA
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2)
    _, _, wsd = load_golden("tip3p774_seed3")
    n_mol = 216
    wpos, wbox, species, bonds = workloads.water_box(n_mol, seed=31, jitter=0.0, wrap=False)
    nw = 3 * n_mol
    mass = np.where(species == 1, workloads.MASS_O, workloads.MASS_H).astype(np.float64).reshape(-1, 1)
    pairs, _ = orc.water_constraints(nw, workloads.TIP3P_R_OH, workloads.TIP3P_R_HH)
    wv0 = np.random.default_rng(32).normal(0, 1.0, (nw, 3)) * 10.0 * np.sqrt(workloads.KB * 300.0 / mass)
    wv0 = orc.rattle_velocities(wpos, wv0, (1.0 / mass).reshape(-1), pairs)
    lmd = dict(dt_ps=0.002, mass_amu=39.9, temperature_k=100.0, gamma_per_ps=25.0, seed=9, remove_cm_motion=True)
    wmd = dict(dt_ps=0.0005, mass_amu=workloads.MASS_O, mass_h_amu=workloads.MASS_H, temperature_k=300.0, rigid_water=True,
               r_oh=workloads.TIP3P_R_OH, r_hh=workloads.TIP3P_R_HH, species=species, seed=5, remove_cm_motion=True)
    v_lj = workloads.maxwell_boltzmann(n_lj, seed=3)

    def run(sa, sb):
        a = GamdForce(sd, n_lj, box, rc, scaler=SHIPPED_SCALERS["lj"], neighbor_skin=rc / 6.0)
        b = GamdForce(wsd, nw, wbox, 4.2, bond=bonds, scaler=SHIPPED_SCALERS["tip3p"], edge_dtype="bf16", neighbor_skin=0.7)
        with torch.cuda.stream(sa):
            xa, va = torch.from_numpy(pos).float().cuda(), torch.from_numpy(v_lj).float().cuda()
            fa = a.forward(xa, denormalize=True).clone()
        with torch.cuda.stream(sb):
            xb, vb = torch.from_numpy(wpos).float().cuda(), torch.from_numpy(wv0).float().cuda()
            fb = b.forward(xb, species=species, denormalize=True).clone()
        for _ in range(2):
            with torch.cuda.stream(sa):
                a.md_run(xa, va, fa, 4, sync=False, **lmd)
            with torch.cuda.stream(sb):
                b.md_run(xb, vb, fb, 6, sync=False, **wmd)
        with torch.cuda.stream(sa):
            assert a.sync_status() == 0
        with torch.cuda.stream(sb):
            assert b.sync_status() == 0
        out = (xa.cpu(), fa.cpu(), xb.cpu(), fb.cpu())
        a.close(); b.close()
        return out

    cur = torch.cuda.current_stream()
    ref = run(cur, cur)
    assert all(torch.isfinite(t).all() for t in ref)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    t0 = time.monotonic()
    done = 0
    for it in range(300):
        got = run(s1, s2)
        for k, (g, r) in enumerate(zip(got, ref)):
            assert torch.equal(g, r), (it, k, float((g - r).abs().max()))
        done += 1
        if time.monotonic() - t0 > 150.0:                    # a slow box: never let the loop eat the suite's time
            break
    print(f"{done} two-stream iterations in {time.monotonic() - t0:.1f} s")
    assert done >= 100
