"""Multi-GPU use of the selected-branch path: shard, don't communicate.

Every (b, g) pair -- and every query token -- of the path is independent: Eq.10 reduces only over the
h heads inside one KV group, the top-n is per (b,t,g) row, K_sel/V_sel are per (b,g).  So the path is
partitioned over the batch x group axis with NO data-path collective; the only
collective of the reference is DDP's gradient all-reduce in its trainer (scripts/train_showcase.py:619-663),
which is torch.distributed machinery (backend "nccl" = RCCL on ROCm) and not part of this package.

This module holds the small host helpers bench.py and a data-parallel caller use:
  * shard_batch(B, world, rank)       -- the reference's split, B_local = B // W (+1 for the first B % W ranks)
                                         (scripts/train_showcase.py:718-723)
  * shard_bg(B, G, world, rank)       -- finer split of the flattened (b,g) axis for inference replicas
  * max_over_ranks(x) / barrier()     -- timing helpers (gloo on CPU, nccl=RCCL on GPUs)
"""
from __future__ import annotations

from typing import List, Tuple

import torch


def shard_batch(B: int, world: int, rank: int) -> Tuple[int, int]:
    """[start, end) of the sequences rank `rank` owns; sizes differ by at most one, every sequence owned once."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(B, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_bg(B: int, G: int, world: int, rank: int) -> List[Tuple[int, int]]:
    """(b, g) pairs owned by `rank` when the flattened batch x group axis is dealt contiguously."""
    s, e = shard_batch(B * G, world, rank)
    return [(i // G, i % G) for i in range(s, e)]


def dist_ready() -> bool:
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def barrier(device=None) -> None:
    if dist_ready():
        torch.distributed.barrier()
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(x: float, device="cpu") -> float:
    """max of a python float over all ranks (identity when not distributed)."""
    if not dist_ready():
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float, device="cpu") -> float:
    if not dist_ready():
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
    return float(t.item())
