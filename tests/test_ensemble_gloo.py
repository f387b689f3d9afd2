"""N>1 path on CPU: two processes, gloo backend — box sharding (seed+rank), the max-over-ranks
timing and the final result gather of gamd_amd/ensemble.py."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np
    from gamd_amd import ensemble as ens
    from gamd_amd.workloads import lj_box
    ctx = ens.init_ensemble("gloo")
    seed = ens.box_seed(1234, ctx)
    pos, box = lj_box(512, seed=seed)
    ens.barrier(ctx)
    tmax = ens.max_over_ranks(1.0 + rank, ctx)
    summ = ens.gather_summary({"steps": 10.0, "seconds": 1.0 + rank, "chk": float(np.abs(pos).sum())}, ctx)
    agg = ens.aggregate_throughput(512 * 10, tmax, ctx)
    q.put((rank, seed, tmax, summ, agg, float(np.abs(pos).sum())))
    ens.shutdown(ctx)


def test_two_rank_ensemble_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, t0, g0, a0, c0), (r1, s1, t1, g1, a1, c1) = res
    assert (s0, s1) == (1234, 1235)               # independent boxes, one per rank
    assert c0 != c1
    assert t0 == t1 == 2.0                        # MAX over ranks
    assert g0 == g1 and [d["seconds"] for d in g0] == [1.0, 2.0]
    assert abs(g0[0]["chk"] - c0) < 1e-6 and abs(g0[1]["chk"] - c1) < 1e-6
    assert a0 == a1 == 512 * 10 * 2 / 2.0          # whole-job units / max time


def test_single_process_context():
    from gamd_amd import ensemble as ens
    ctx = ens.EnsembleContext()
    assert not ctx.distributed and ens.max_over_ranks(3.0, ctx) == 3.0
    assert ens.gather_summary({"a": 1.0}, ctx) == [{"a": 1.0}]


def _spawn(code):
    import subprocess
    return subprocess.Popen([sys.executable, "-c", code])


def test_supervise_ranks_stops_the_survivors_of_a_failed_rank():
    """bench.py --gpus N (self-spawned ranks): a rank that dies must not leave its peers waiting in a collective.  The
    supervisor sees the non-zero exit, terminates the others (exact PIDs) and reports (rank, status) quickly."""
    import time
    from gamd_amd.ensemble import supervise_ranks
    procs = [_spawn("import time; time.sleep(300)"), _spawn("import sys, time; time.sleep(0.5); sys.exit(3)"),
             _spawn("import signal, time; signal.signal(signal.SIGTERM, signal.SIG_IGN); time.sleep(300)")]
    t0 = time.monotonic()
    failed = supervise_ranks(procs, timeout_s=120, grace_s=1.0)
    dt = time.monotonic() - t0
    assert failed == (1, 3)
    assert dt < 15, dt
    assert all(p.poll() is not None for p in procs)          # nobody left behind, not even the SIGTERM-deaf rank
    assert procs[0].returncode < 0 and procs[2].returncode < 0


def test_supervise_ranks_all_ok_and_timeout():
    from gamd_amd.ensemble import supervise_ranks
    assert supervise_ranks([_spawn("pass"), _spawn("pass")], timeout_s=60) is None
    procs = [_spawn("import time; time.sleep(300)")]
    assert supervise_ranks(procs, timeout_s=1.0, grace_s=1.0) == (-1, 124)
    assert procs[0].poll() is not None


def _dying_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    from gamd_amd import ensemble as ens
    ctx = ens.init_ensemble("gloo", timeout_s=10.0)
    if rank == 1:
        os._exit(3)                                             # dies after the rendezvous, before the barrier
    t0 = time.monotonic()
    try:
        ens.barrier(ctx)
        q.put(("no error", time.monotonic() - t0))
    except Exception as exc:                                    # the bounded process-group timeout / a closed peer
        q.put((type(exc).__name__, time.monotonic() - t0))


def test_dead_peer_surfaces_as_an_error_not_a_hang():
    """init_ensemble's timeout bounds every collective: the survivor of a dead peer gets an exception within it."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_dying_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    what, dt = q.get(timeout=90)
    for p in procs:
        p.join(30)
    assert what != "no error" and dt < 60, (what, dt)


def test_world_size_1_group_runs_the_collectives():
    """force_group: the process-group code path (barrier, all_reduce MAX, gather) at world size 1 — with backend "nccl" on
    a GPU box this is the RCCL branch (tests/test_gpu_bench_contract.py); here gloo."""
    code = (
        "import os, sys; sys.path.insert(0, %r)\n"
        "os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='%d')\n"
        "from gamd_amd import ensemble as ens\n"
        "ctx = ens.init_ensemble('gloo', force_group=True)\n"
        "assert ctx.distributed and ctx.world == 1\n"
        "ens.barrier(ctx)\n"
        "assert ens.max_over_ranks(2.5, ctx) == 2.5\n"
        "assert ens.gather_summary({'a': 1.0, 'b': 2.0}, ctx) == [{'a': 1.0, 'b': 2.0}]\n"
        "ens.shutdown(ctx)\n")
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    p = subprocess.run([sys.executable, "-c", code % (ROOT, port)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
