"""bench.py keeps the driver's contract: flags, exactly one JSON line on stdout, the required keys; N > 1 launches its
own ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LINE_LIMIT = 6000          # the driver keeps an 8 081-character tail of stdout: round 5's 21 KB line could not be parsed
CONTRACT_ORDER = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline"]


def _strict(text):
    """A strict parser: no NaN / Infinity literals, no duplicate keys; returns the dict with its key order."""
    def no_const(name):
        raise AssertionError(f"non-finite literal {name} in the result line")
    def pairs(items):
        keys = [k for k, _ in items]
        assert len(keys) == len(set(keys)), keys
        return dict(items)
    return json.loads(text, parse_constant=no_const, object_pairs_hook=pairs)


def _run(*flags, env=None, steps=6, warmup=2, with_detail=False, tmp=None):
    """Run bench.py; returns the ONE stdout line (checked: one line, <= LINE_LIMIT characters, strict JSON) and, with
    with_detail, the full record bench.py wrote beside it (--detail)."""
    import tempfile
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        e.pop(k, None)
    e.update(env or {})
    with tempfile.TemporaryDirectory() as td:
        det = os.path.join(td, "bench_detail.json")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warmup),
                            "--detail", det, *flags], capture_output=True, text=True, timeout=900, cwd=ROOT, env=e)
        assert p.returncode == 0, p.stderr[-2000:]
        assert p.stdout.endswith("\n") and p.stdout.count("\n") == 1, p.stdout[-3000:]      # exactly one line, nothing else
        text = p.stdout.strip()
        assert len(text) < LINE_LIMIT, len(text)
        d = _strict(text)
        if not with_detail:
            return d
        full = json.load(open(det))
        assert full["line"] == d
        return d, full["detail"]


def test_default_line_has_the_contract_keys():
    def timing_consistent(d):            # relations between clocks, which one stall from outside the process can break
        tr, fe = d["config"]["timed_region"], d["force_eval_only"]
        return (abs(tr["mean"] - d["ms_per_step"]) / d["ms_per_step"] < 0.05
                and 0.8 * d["ms_per_step"] < fe["ms_per_eval"] < 1.3 * d["ms_per_step"])
    d, full = _run("--no-cpu-baseline", "--no-secondary", with_detail=True)
    if not timing_consistent(d):
        print("first attempt:", d["ms_per_step"], d["config"]["timed_region"], d["force_eval_only"]["ms_per_eval"])
        d, full = _run("--no-cpu-baseline", "--no-secondary", with_detail=True)
    assert list(d)[:len(CONTRACT_ORDER)] == CONTRACT_ORDER, list(d)
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["unit"] == "atom-steps/s" and "10k-atom LJ box" in d["metric"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 10000 * 6 / (d["ms_per_step"] * 6e-3)) / d["value"] < 1e-4     # 6 significant digits in the line
    r = d["roofline"]
    assert list(r)[:10] == ["kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches", "flop_per_launch"]
    assert r["kernel"] == "k_conv_edge" and r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 and 0.3 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["flop_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) / r["achieved"] < 1e-4
    assert r["launches"] == 4 * 6                         # four conv layers per timed step, timed live with HIP events
    # the PMC traffic figure and the rocprofv3 launch average are reported only for the kernel sources they were measured on
    # (profiles/pmc_conv_edge.json, hash-stamped), otherwise null (+ a note in the detail record)
    fr = full["roofline"]
    assert (r["traffic"] is not None) != ("traffic_note" in fr)
    assert "rocprof_avg_launch_ms" in r and (r["rocprof_avg_launch_ms"] is None) == (r["traffic"] is None)
    if r["rocprof_avg_launch_ms"] is not None:
        assert abs(r["rocprof_avg_launch_ms"] / r["avg_launch_ms"] - 1.0) < 0.06 and abs(r["rocprof_frac"] / r["frac"] - 1.0) < 0.06
    assert 0.5 < r["whole_step_frac"] < 1.0
    g = r["neighbour_gather"]
    assert set(g) == {"bytes_per_step", "GB_per_s", "frac_of_hbm_peak"} and 0.0 < g["frac_of_hbm_peak"] < 1.0
    assert fr["neighbour_gather"]["bound"].startswith("mfma-bound")
    assert d["ensemble"]["boxes"] == 1 and len(d["ensemble"]["rank_seconds"]) == 1 and len(full["ensemble"]["per_rank"]) == 1
    # the record explains its own number: per-step distribution, rebuilds / regrows inside the timed region, cost of a rebuild
    tr, ftr = d["config"]["timed_region"], full["config"]["timed_region"]
    assert tr["intervals"] == 6 and ftr["step_ms"]["min"] <= tr["p50"] * (1 + 1e-3) and tr["p50"] <= tr["p99"] <= tr["max"]
    assert abs(tr["mean"] - d["ms_per_step"]) / d["ms_per_step"] < 0.05      # device events vs the host clock
    assert tr["regrown_in_timed"] is False and 0 <= tr["rebuilds_in_timed"] <= 6
    assert 0.0 < tr["rebuild_ms"] < 1.0 and ftr["neighbour_stage_ms"]["reuse_step"] < ftr["neighbour_stage_ms"]["rebuild_step"]
    # prose lives in the detail record only
    assert "ceiling_note" in fr and "ceiling_note" not in r and "kernels" not in r
    # the force-evaluation-only figure beside the MD step (one synchronous call = a step without the integrator
    # plus the host round trip of the call)
    fe = d["force_eval_only"]
    assert fe["unit"] == "atom-evals/s" and abs(fe["value"] - 10000 / (fe["ms_per_eval"] * 1e-3)) / fe["value"] < 1e-3
    assert 0.8 * d["ms_per_step"] < fe["ms_per_eval"] < 1.3 * d["ms_per_step"]


def test_headline_steps_are_evenly_paced():
    """C2, 100 timed steps: no step of the timed region may stand out (p99 / p50 < 1.3; a candidate rebuild costs ~0.1 ms of a
    3.2 ms step) and the device-event mean agrees with the host clock: a stall inside the region would show in `max`."""
    def paced(d):
        s = d["config"]["timed_region"]
        return (s["intervals"] == 100 and s["p99"] / s["p50"] < 1.3 and s["max"] / s["p50"] < 1.5
                and abs(s["mean"] - d["ms_per_step"]) / d["ms_per_step"] < 0.02)
    d = _run("--no-cpu-baseline", "--no-secondary", steps=100, warmup=20)
    if not paced(d):
        # a timing property of a shared machine: one stall from outside the process (seen once in ~20 runs of this test, and in
        # round 4's driver record) must not fail the suite; two in a row are the workload's
        print("first attempt not evenly paced:", d["config"]["timed_region"], d["ms_per_step"])
        d = _run("--no-cpu-baseline", "--no-secondary", steps=100, warmup=20)
    tr = d["config"]["timed_region"]
    assert paced(d), (tr, d["ms_per_step"])
    assert 0 <= tr["rebuilds_in_timed"] <= 10 and tr["regrown_in_timed"] is False and tr["regrown_in_warmup"] is False


def test_long_water_run_does_not_outgrow_its_neighbour_buffers():
    """1 500 steps of the rigid-water workload: the fixed-width candidate rows (capacity / N per atom) must hold the longest
    row as the box melts — an overflow is survivable (freeze, regrow, resume) but costs the frozen steps, and the bench
    line says so."""
    d = _run("--no-cpu-baseline", "--no-secondary", "--workload", "c3", steps=1500, warmup=10)
    assert d["config"]["timed_region"]["regrown_in_timed"] is False and d["config"]["timed_region"]["regrown_in_warmup"] is False
    assert "candidate rebuilds" in d["config"]["neighbour_list"]


def test_split_fp16_line_is_labelled_as_such():
    d = _run("--no-cpu-baseline", "--no-secondary", "--edge-dtype", "f16x3")
    assert d["dtype"].startswith("f16x3") and d["roofline"]["kernel"] == "k_conv_edge_f16x3"
    assert d["roofline"]["peak"] == 2500.0 and d["roofline"]["frac"] < 1.0


def test_gpus_2_spawns_two_ranks_by_itself():
    """`python bench.py --gpus 2` without torchrun: the parent starts one child per rank before touching a GPU and relays
    rank 0's line.  Dry run of the N > 1 control flow on this one-GPU box (both ranks share device 0, gloo for the result
    gather); on an 8-GPU node the same code puts rank r on device r with RCCL."""
    d, full = _run("--gpus", "2", "--no-cpu-baseline", "--no-secondary", steps=4, warmup=1, with_detail=True,
                   env={"GAMD_BENCH_SHARE_GPU": "1", "GAMD_BENCH_BACKEND": "gloo"})
    assert list(d)[:len(CONTRACT_ORDER)] == CONTRACT_ORDER, list(d)          # the same compact line as N = 1
    assert d["n_gpus"] == 2 and d["config"]["boxes"] == 2 and d["steps"] == 4 and d["scaling"] == "weak"
    assert d["config"]["launch"] == "self-spawned ranks"
    assert d["ensemble"]["boxes"] == 2 and len(d["ensemble"]["rank_seconds"]) == 2
    ranks = full["ensemble"]["per_rank"]
    assert [r["rank"] for r in ranks] == [0, 1] and [r["box_seed"] for r in ranks] == [1234, 1235]
    assert ranks[0]["force_abs_sum"] != ranks[1]["force_abs_sum"] and ranks[0]["edges"] != ranks[1]["edges"]
    assert d["ensemble"]["collective_on_step_path"] is False
    # whole-job throughput = all ranks' atom-steps / max-over-ranks time
    t_max = full["ms_per_step"] * 4e-3
    assert abs(full["value"] - 2 * 10000 * 4 / t_max) / full["value"] < 1e-6 and abs(d["value"] / full["value"] - 1) < 1e-5
    assert t_max >= max(r["seconds"] for r in ranks) * (1 - 1e-9)
    assert "cpu_baseline" not in d and "secondary" not in d


def test_the_drivers_own_launch_line_for_n_greater_1():
    """The driver starts N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W`: ranks come up with RANK / LOCAL_RANK / WORLD_SIZE in the
    environment and bench.py must NOT spawn again.  Same dry run as above (two ranks on this box's one GPU, gloo)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    e.update(GAMD_BENCH_SHARE_GPU="1", GAMD_BENCH_BACKEND="gloo")
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        det = os.path.join(td, "detail.json")
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                            "--no-cpu-baseline", "--no-secondary", "--detail", det], capture_output=True, text=True, timeout=600, cwd=ROOT, env=e)
        assert p.returncode == 0, p.stderr[-2000:]
        full = json.load(open(det))["detail"]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < LINE_LIMIT, p.stdout                  # rank 0 alone prints, the compact line
    d = _strict(lines[0])
    assert list(d)[:len(CONTRACT_ORDER)] == CONTRACT_ORDER, list(d)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["config"]["boxes"] == 2
    assert d["config"]["launch"] != "self-spawned ranks" and d["ensemble"]["boxes"] == 2
    assert [r["rank"] for r in full["ensemble"]["per_rank"]] == [0, 1] and all(r["group_world_size"] == 2 for r in full["ensemble"]["per_rank"])
    assert abs(d["value"] - 2 * 10000 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-4


def test_gpus_2_line_names_each_ranks_device():
    """per_rank carries the device ordinal, the PCI bus id and the size of the process group each rank saw: on an 8-GPU node
    'N ranks on N distinct GPUs' is readable from the line (here both ranks share device 0 on purpose)."""
    d, full = _run("--gpus", "2", "--no-cpu-baseline", "--no-secondary", steps=2, warmup=1, with_detail=True,
                   env={"GAMD_BENCH_SHARE_GPU": "1", "GAMD_BENCH_BACKEND": "gloo"})
    ranks = full["ensemble"]["per_rank"]
    assert all(r["group_world_size"] == 2 and r["device"] == 0 and r["host_threads"] >= 1 for r in ranks)
    assert all(len(r["pci_bus_id"].split(":")) == 3 for r in ranks) and ranks[0]["pci_bus_id"] == ranks[1]["pci_bus_id"]
    assert d["ensemble"]["distinct_devices"] == 1


def test_a_failing_rank_ends_the_run_quickly_and_leaves_no_orphan():
    """Rank 1 dies after the rendezvous (GAMD_BENCH_FAIL_RANK): the launcher must stop rank 0 — which would otherwise wait
    in the first barrier until the process-group timeout — exit non-zero well within a minute and leave no process behind."""
    import time
    import uuid
    import psutil
    mark = uuid.uuid4().hex
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(GAMD_BENCH_SHARE_GPU="1", GAMD_BENCH_BACKEND="gloo", GAMD_BENCH_FAIL_RANK="1", GAMD_TEST_MARK=mark)
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2000", "--warmup", "1",
                        "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=e)
    dt = time.monotonic() - t0
    assert p.returncode != 0 and p.stdout.strip() == "", (p.returncode, p.stdout[-500:])
    assert "rank 1 exited with status 3" in p.stderr or "rank 0 exited" in p.stderr, p.stderr[-1000:]
    assert dt < 60, dt
    left = []
    for q in psutil.process_iter(["pid"]):
        try:
            if q.environ().get("GAMD_TEST_MARK") == mark:
                left.append(q.pid)
        except (psutil.Error, OSError):
            pass
    assert left == [], left


RCCL_WS1 = r"""
import os, sys
sys.path.insert(0, %r)
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="%d")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
from gamd_amd import ensemble as ens
ctx = ens.init_ensemble("nccl", force_group=True, timeout_s=120)       # device_id=cuda:0 -> eager RCCL communicator
assert ctx.distributed and ctx.backend == "nccl" and dist.get_backend() == "nccl"
ens.barrier(ctx)                                                        # dist.barrier(device_ids=[0])
assert ens.max_over_ranks(2.5, ctx, device="cuda:0") == 2.5            # all_reduce(MAX) on a CUDA tensor
ident = ens.device_identity(0)
summ = ens.gather_summary({"seconds": 1.25, "edges": 634930.0, **ident}, ctx, device="cuda:0")   # all_gather_into_tensor
assert len(summ) == 1 and summ[0]["seconds"] == 1.25 and summ[0]["edges"] == 634930.0
assert summ[0]["group_world_size"] == 1.0 and summ[0]["pci_bus"] >= 0
print("PCI", ens.pci_string(summ[0]))
ens.shutdown(ctx)
print("RCCL_WS1_OK")
"""


def test_rccl_branch_at_world_size_1():
    """The branch an 8-GPU run takes (backend "nccl" = RCCL, device_id, barrier(device_ids), all_reduce(MAX) and
    all_gather_into_tensor on CUDA tensors), executed on this one-GPU box with a process group of size 1."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-c", RCCL_WS1 % (ROOT, port)], capture_output=True, text=True, timeout=600, env=e)
    assert p.returncode == 0 and "RCCL_WS1_OK" in p.stdout, (p.stdout[-1000:], p.stderr[-3000:])


def test_gpus_2_without_a_second_device_fails_loudly():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "GAMD_BENCH_SHARE_GPU")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=e)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices present")
    assert p.returncode != 0 and p.stdout.strip() == "" and "only 1 HIP device" in p.stderr


def test_the_drivers_default_command_prints_one_short_line_with_roofline_and_cpu_baseline():
    """`python3 bench.py --gpus 1 --steps 20 --warmup 5` — the driver's command: ONE line of < 6 000 characters (round 5's line
    was 21 KB and the driver, which keeps an 8 081-character tail, parsed nothing), strict JSON, the contract keys in the
    contract's order, `roofline` and `cpu_baseline` present, compact secondary triples; everything else in the detail file."""
    d, full = _run("--gpus", "1", steps=20, warmup=5, with_detail=True)
    assert list(d)[:len(CONTRACT_ORDER)] == CONTRACT_ORDER, list(d)
    assert list(d)[len(CONTRACT_ORDER):len(CONTRACT_ORDER) + 2] == ["cpu_baseline", "force_eval_only"]
    assert list(d["config"])[:5] == ["workload", "n_atoms", "edges_per_step", "boxes", "neighbour_list"]
    assert d["config"]["n_atoms"] == 10000 and d["config"]["boxes"] == 1 and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    cb, fcb = d["cpu_baseline"], full["cpu_baseline"]
    assert list(cb)[:5] == ["value", "unit", "cores", "kind", "sample"]
    assert cb["kind"] == "port" and "10000-atom" in cb["sample"] and cb["cores"] >= 1 and cb["host_threads"] >= cb["cores"]
    assert cb["gpu_vs_cpu_rel_err"] < 1e-5 and fcb["sample_2000"]["same_edge_count"] is True
    assert cb["gpu_vs_cpu_per_atom_p99"] < 1e-5 and fcb["gpu_vs_cpu_per_atom"]["atoms"] > 9000
    assert abs(cb["value"] - 10000 / cb["seconds_per_eval"]) / cb["value"] < 1e-3
    assert d["value"] / cb["value"] > 10.0                                # north star: >= 10x the CPU path
    r = d["roofline"]
    assert 0.5 < r["whole_step_frac"] < 1.0 and 0.3 < r["frac"] < 1.0 and r["launches"] == 4 * 20
    ok = r["other_kernels"]                                               # name -> [avg launch ms, frac of the matrix peak]
    assert list(ok)[0] == "k_edge_encode" and any(n.startswith("k_node") for n in ok) and any("neighbour" in n for n in ok)
    assert 0.3 < ok["k_edge_encode"][1] < 1.0
    names = [k["kernel"] for k in full["roofline"]["kernels"]]
    assert names[0] == "k_edge_encode" and any(n.startswith("k_node") for n in names) and any("neighbour" in n for n in names)
    enc = full["roofline"]["kernels"][0]
    assert enc["bound"] == "mfma" and 0.3 < enc["frac"] < 1.0 and abs(enc["frac"] - enc["achieved"] / 157.3) < 1e-9
    s = d["secondary"]
    assert set(s) == {"c1", "c3", "c5", "c5b", "dft"} and set(full["secondary"]) == set(s)
    for k, (ms, val, frac) in s.items():
        fs = full["secondary"][k]
        assert ms > 0 and abs(val - fs["n_atoms"] * 20 / (ms * 20e-3)) / val < 2e-3 and 0.0 < frac < 1.0, (k, ms, val, frac)
        assert fs["finite"] is True and fs["timed_region"]["regrown_in_timed"] is False
    assert full["secondary"]["c1"]["conv_kernel"]["kernel"] == "k_conv_edge_small"     # the kernel the trace shows for 258 atoms
    assert full["secondary"]["c5"]["dtype"] == "bf16" and full["secondary"]["c5"]["conv_kernel"]["bound"] == "hbm"
    assert d["detail_file"] == "bench_detail.json"


def test_secondary_full_goes_to_the_detail_file_only():
    """--secondary full: all twelve secondary workloads with their per-step distributions — in the detail record; the line
    still carries only the compact triples and stays under the limit."""
    d, full = _run("--no-cpu-baseline", "--secondary", "full", steps=10, warmup=3, with_detail=True)
    assert set(d["secondary"]) == {"c1", "c3", "c5", "c5b", "dft"}
    s = full["secondary"]
    assert set(s) == {"c1", "c1_batch", "c1_batch_f16x3", "c2_batch8", "c3", "c5", "c5b", "c2_f16x3", "c2_bf16", "dft", "dft_f16x3",
                      "dft_bf16"}
    # config 4's eight rank boxes as one batch on this GPU: per-atom-step cost no worse than the single box
    assert s["c2_batch8"]["n_boxes"] == 8 and s["c2_batch8"]["n_atoms"] == 80000 and s["c2_batch8"]["value"] > 0.97 * full["value"]
    for k, v in s.items():
        tr = v["timed_region"]
        assert tr["step_ms"]["intervals"] >= 20 and tr["regrown_in_timed"] is False, (k, tr)
        assert tr["step_ms"]["max"] < 3.0 * tr["step_ms"]["p50"] + 0.3, (k, tr["step_ms"])       # no stall inside a timed region
    assert s["c1_batch_f16x3"]["n_boxes"] == 38 and s["c1_batch_f16x3"]["value"] > 1.5 * s["c1_batch"]["value"]
    assert s["dft_bf16"]["dtype"] == "bf16" and s["dft_bf16"]["conv_kernel"]["kernel"] == "k_conv_edge_bf16_wide<2,2>"
    assert s["dft_f16x3"]["dtype"].startswith("f16x3") and s["dft_f16x3"]["conv_kernel"]["kernel"] == "k_conv_edge_f16x3_wide<2,2>"
    assert s["dft_f16x3"]["value"] > 1.2 * s["dft"]["value"] and s["c2_f16x3"]["value"] > 1.5 * full["value"]
    assert s["c2_bf16"]["dtype"] == "bf16" and s["c2_bf16"]["conv_kernel"]["bound"] == "hbm" and s["c2_bf16"]["n_atoms"] == 10000
    assert s["c1_batch"]["n_boxes"] == 38 and s["c1_batch"]["n_atoms"] == 38 * 258 and s["c1_batch"]["value"] > 1.5 * s["c1"]["value"]
    assert s["c2_f16x3"]["dtype"].startswith("f16x3") and s["c2_f16x3"]["n_atoms"] == 10000 and s["dft"]["n_atoms"] == 774
    assert s["c1"]["edges_per_step"] > 6000                      # the reference snapshot (6 114 edges at the start)
    assert s["c1"]["n_atoms"] == 258 and s["c3"]["n_atoms"] == 4170 and s["c5"]["n_atoms"] == 6000 and s["c5b"]["n_atoms"] == 8001
    assert s["c5"]["dtype"] == "bf16" and s["c5"]["conv_kernel"]["bound"] == "hbm" and s["c3"]["dtype"] == "f32"
    for v in s.values():
        assert v["finite"] is True and v["value"] > 0 and abs(v["value"] - v["n_atoms"] * 20 / (v["ms_per_step"] * 20e-3)) / v["value"] < 1e-6


def test_graft_entry_smoke_runs():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()
