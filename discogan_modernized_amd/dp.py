"""Data-parallel exchange step (reference: DistributedDataParallel's gradient all-reduce,
distributed_image_translation.py:401-404,513-518).

Pure data parallelism: every rank holds full replicas (identical from the shared seed), draws its own
batch, keeps BatchNorm / feature-matching statistics local, and exchanges ONE message per iteration:
the flat gradient buffer of the side being stepped (D_A+D_B or G_A+G_B), summed over ranks.  The
division by the world size is folded into the Adam kernel (``grad_scale``).  No BatchNorm-buffer
broadcast (that is what breaks the reference's DDP run) and no all-reduce of the
side whose gradients are discarded (F5).  Backend: ``nccl`` == RCCL over xGMI on MI355X; the same
code runs on ``gloo`` for the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def all_reduce_flat(flat: torch.Tensor, group=None, async_op: bool = False):
    """Sum ``flat`` over the ranks of ``group`` in place.  Returns (scale, work): multiply the summed
    gradient by ``scale`` (= 1/W) to obtain DDP's mean; ``work`` is the async handle or None."""
    w = world_size(group)
    if w == 1:
        return 1.0, None
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return 1.0 / w, work


def rank_data_seed(rank: int, base: int = 1000) -> int:
    """Per-rank synthetic-batch seed (seed 1000 + rank)."""
    return base + rank


def broadcast_flat(flat: torch.Tensor, src: int = 0, group=None):
    """Optional initial replica sync (DDP constructor broadcast, C3); replicas are already identical
    when every rank seeds 1234, so this is only used after loading checkpoints on rank 0."""
    if world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
