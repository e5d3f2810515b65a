"""Batch-parallel sharding of independent PDE trajectories.

Trajectories in a batch never interact in any forward (block-diagonal graphs, per-graph norms), so
the rollout shards over ranks with NO data-path collective; the only communication is the reduction
of scalar metrics / timings at the end.  One process per GPU, torch.distributed ('nccl' = RCCL on
ROCm; 'gloo' in the CPU tests).  Training adds exactly one collective per step: the all-reduce of
the parameter gradients (``allreduce_gradients``, what Lightning's strategy='ddp' does for the
reference, configs/trainer/default.yaml) - one flat 2.5 MB buffer, issued once after the rollout's
backward, because the gradients accumulate over the rollout's forward calls and are final only then.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (1-process default)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced slice of ``n_items`` trajectories owned by ``rank`` (what Lightning's
    DistributedSampler does for strategy='ddp', minus the shuffling)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """Slice every tensor of a batch dict along dim 0."""
    n = next(iter(batch.values())).shape[0]
    r = shard_range(n, rank, world)
    return {k: v[r.start:r.stop] for k, v in batch.items()}


def max_over_ranks(value: float, device="cpu") -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device="cpu") -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_predictions(pred: torch.Tensor) -> torch.Tensor:
    """Concatenate per-rank predictions along dim 0 on every rank (evaluation only)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return pred
    parts = [torch.empty_like(pred) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, pred.contiguous())
    return torch.cat(parts, 0)


def allreduce_gradients(parameters, average: bool = True) -> None:
    """Sum (or average) ``.grad`` of every parameter over the ranks with ONE all-reduce of a flat buffer
    (RCCL over xGMI when the gradients live on GPUs).  DDP semantics: every rank ends with identical
    gradients; with equal shard sizes and a mean loss the average equals the full-batch gradient."""
    params = [p for p in parameters if p.grad is not None]
    if not params or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    from .optim import flat_gradient
    flat = flat_gradient(params)                    # the HIP backward leaves all gradients in one buffer: reduce in place
    if flat is not None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= dist.get_world_size()
        return
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    o = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n
