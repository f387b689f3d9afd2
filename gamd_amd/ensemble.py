"""Multi-GPU ensemble: independent MD boxes, one per GPU / process.

The path shards by *independent units*: every rank owns its own box (positions,
velocities, neighbour state, RNG stream seed+rank) and a replica of the 2.6 MB
weights; there is NO collective on the step path.  The only communication is
the trivial result gather at the end (per-rank step counts, timings, force
checksums) — `torch.distributed` all_gather / all_reduce, which is RCCL over
xGMI with backend "nccl" on ROCm and gloo in the CPU tests.
"""
from __future__ import annotations

import datetime
import os
import subprocess
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


@dataclass
class EnsembleContext:
    rank: int = 0
    world: int = 1
    local_rank: int = 0
    backend: str = ""
    grouped: bool = False      # a process group exists (world > 1, or world == 1 with force_group)

    @property
    def distributed(self) -> bool:
        return self.world > 1 or self.grouped


def init_ensemble(backend: str = "nccl", device_index=None, timeout_s: float = 120.0,
                  force_group: bool = False) -> EnsembleContext:
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract) and join
    the process group when WORLD_SIZE > 1 (``force_group``: also at world size 1, so that the RCCL code path can be
    exercised on a single-GPU box).  ``timeout_s`` bounds the rendezvous and every collective: a rank that died
    surfaces as an error on the others instead of a hang."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or force_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # the host driver only supports dmabuf IPC; without this RCCL fails with hipIpcGetMemHandle errors
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {}
        if backend == "nccl":
            d = local if device_index is None else device_index
            torch.cuda.set_device(d)
            kw["device_id"] = torch.device("cuda", d)
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=float(timeout_s)), **kw)
    grouped = world > 1 or force_group
    return EnsembleContext(rank, world, local, backend if grouped else "", grouped)


def box_seed(base_seed: int, ctx: EnsembleContext) -> int:
    """Independent unit owned by this rank: box number = rank (seed 1234+rank in BASELINE.md)."""
    return base_seed + ctx.rank


def barrier(ctx: EnsembleContext, device=None) -> None:
    if ctx.distributed:
        if ctx.backend == "nccl":
            d = None if device is None else torch.device(device).index      # "cuda" carries no index: the current device
            dist.barrier(device_ids=[torch.cuda.current_device() if d is None else d])
        else:
            dist.barrier()


def device_identity(device: int) -> Dict[str, float]:
    """Which physical GPU a rank runs on, as floats for gather_summary: ordinal, PCI domain / bus / device (from the HIP
    device properties) and the size of the process group this rank sees.  'N ranks on N distinct GPUs' is then readable
    from the bench line."""
    p = torch.cuda.get_device_properties(device)
    return {"device": float(device), "pci_domain": float(getattr(p, "pci_domain_id", -1)),
            "pci_bus": float(getattr(p, "pci_bus_id", -1)), "pci_device": float(getattr(p, "pci_device_id", -1)),
            "group_world_size": float(dist.get_world_size() if dist.is_initialized() else 1)}


def pci_string(s: Dict[str, float]) -> str:
    return "%04x:%02x:%02x" % (int(s["pci_domain"]) & 0xffff, int(s["pci_bus"]) & 0xff, int(s["pci_device"]) & 0xff)


def pin_host_threads(ctx: EnsembleContext, local_world: int = 0) -> int:
    """One share of the host cores per rank (torch's intra-op pool): N ranks each defaulting to all cores oversubscribe
    the host side of a step (launch enqueue, pinned copies)."""
    lw = local_world or int(os.environ.get("LOCAL_WORLD_SIZE", str(max(ctx.world, 1))))
    n = max(1, (os.cpu_count() or 1) // max(lw, 1))
    torch.set_num_threads(n)
    return n


def max_over_ranks(value: float, ctx: EnsembleContext, device="cpu") -> float:
    if not ctx.distributed:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_summary(local: Dict[str, float], ctx: EnsembleContext, device="cpu") -> List[Dict[str, float]]:
    """The 'trivial result gather': every rank contributes a small dict of floats (steps done,
    seconds, force checksum, ...); every rank receives the list ordered by rank."""
    keys = sorted(local)
    if not ctx.distributed:
        return [dict(local)]
    t = torch.tensor([float(local[k]) for k in keys], dtype=torch.float64, device=device)
    out = torch.empty((ctx.world, len(keys)), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, t) if device != "cpu" else dist.all_gather(list(out.unbind(0)), t)
    out = out.cpu()
    return [{k: float(out[r, i]) for i, k in enumerate(keys)} for r in range(ctx.world)]


def aggregate_throughput(units_per_rank: float, seconds_max: float, ctx: EnsembleContext) -> float:
    """whole-job throughput: all ranks' units / max-over-ranks time (weak scaling)."""
    return units_per_rank * ctx.world / seconds_max


def shutdown(ctx: EnsembleContext) -> None:
    if ctx.distributed and dist.is_initialized():
        dist.destroy_process_group()


def stop_ranks(procs: Sequence["subprocess.Popen"], grace_s: float = 5.0) -> None:
    """SIGTERM, then SIGKILL after `grace_s`, to whichever of the given children (exact PIDs) are still running."""
    alive = [p for p in procs if p.poll() is None]
    for p in alive:
        p.terminate()
    t_end = time.monotonic() + grace_s
    for p in alive:
        try:
            p.wait(timeout=max(0.0, t_end - time.monotonic()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()


def supervise_ranks(procs: Sequence["subprocess.Popen"], timeout_s: float = 1800.0, poll_s: float = 0.2,
                    grace_s: float = 5.0) -> Optional[Tuple[int, int]]:
    """Watch the rank processes a launcher started (bench.py --gpus N without torchrun).  Returns None when every rank
    exited 0; otherwise (rank, status) of the first rank seen to fail — or (-1, 124) when `timeout_s` ran out — after
    the remaining ranks have been stopped: SIGTERM, then SIGKILL after `grace_s`, exact PIDs of the given children only.
    A rank that dies would otherwise leave its peers inside a barrier / collective until the process-group timeout."""
    deadline = time.monotonic() + float(timeout_s)
    failed = None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                failed = (-1, 124)
                break
            time.sleep(poll_s)
    finally:
        stop_ranks(procs, grace_s)
    return failed
