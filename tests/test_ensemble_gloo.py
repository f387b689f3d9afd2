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
