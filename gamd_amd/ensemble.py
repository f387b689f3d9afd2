"""Multi-GPU ensemble: independent MD boxes, one per GPU / process.

The path shards by *independent units*: every rank owns its own box (positions,
velocities, neighbour state, RNG stream seed+rank) and a replica of the 2.6 MB
weights; there is NO collective on the step path.  The only communication is
the trivial result gather at the end (per-rank step counts, timings, force
checksums) — `torch.distributed` all_gather / all_reduce, which is RCCL over
xGMI with backend "nccl" on ROCm and gloo in the CPU tests.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List

import torch
import torch.distributed as dist


@dataclass
class EnsembleContext:
    rank: int = 0
    world: int = 1
    local_rank: int = 0
    backend: str = ""

    @property
    def distributed(self) -> bool:
        return self.world > 1


def init_ensemble(backend: str = "nccl", device_index=None) -> EnsembleContext:
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract) and join
    the process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # the host driver only supports dmabuf IPC; without this RCCL fails with hipIpcGetMemHandle errors
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {}
        if backend == "nccl":
            d = local if device_index is None else device_index
            torch.cuda.set_device(d)
            kw["device_id"] = torch.device("cuda", d)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return EnsembleContext(rank, world, local, backend if world > 1 else "")


def box_seed(base_seed: int, ctx: EnsembleContext) -> int:
    """Independent unit owned by this rank: box number = rank (seed 1234+rank in BASELINE.md)."""
    return base_seed + ctx.rank


def barrier(ctx: EnsembleContext, device=None) -> None:
    if ctx.distributed:
        if ctx.backend == "nccl":
            dist.barrier(device_ids=[ctx.local_rank])
        else:
            dist.barrier()


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
