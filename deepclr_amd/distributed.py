"""Multi-GPU sharding of scan pairs.

The reference has no distributed code at all. Pairs are independent -- no batch
norm, every max is per pair -- so a batch is split contiguously over ranks (one process per GPU), each rank
keeps the reference's local layout ``[T..., S...]``, and the only exchange is one all-gather of the
``(B_local, label_dim)`` outputs: 256 bytes per rank at B_local = 8. Backend ``nccl`` is RCCL on ROCm; the
same code runs on ``gloo`` for the CPU tests.
"""
from typing import Callable, Tuple

import torch
import torch.distributed as dist


def pair_range(n_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, stop) of the pairs owned by `rank` (earlier ranks take the remainder)."""
    if not 0 <= rank < world:
        raise ValueError("rank outside world")
    base, extra = divmod(n_pairs, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def local_batch(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Slice a global batch (2B, N, C) = [T0..TB-1, S0..SB-1] into this rank's (2B_local, N, C)."""
    if x.shape[0] % 2 != 0:
        raise RuntimeError("batch must hold templates followed by the same number of sources")
    pairs = x.shape[0] // 2
    lo, hi = pair_range(pairs, rank, world)
    return torch.cat((x[lo:hi], x[pairs + lo:pairs + hi]), dim=0)


def gather_outputs(y_local: torch.Tensor, n_pairs: int) -> torch.Tensor:
    """All-gather per-rank outputs (B_local, D) into (n_pairs, D), in global pair order, on every rank. With a process
    group of ONE rank the collective still runs (a copy through RCCL: what `bench.py --force-dist` and the one-GPU RCCL test
    exercise); without a process group the outputs are returned as they are."""
    if not dist.is_initialized():
        return y_local
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [pair_range(n_pairs, r, world) for r in range(world)]
    sizes = [hi - lo for lo, hi in counts]
    if len(set(sizes)) == 1:
        out = y_local.new_empty(n_pairs, y_local.shape[1])
        dist.all_gather_into_tensor(out, y_local.contiguous())
        return out
    width = max(sizes)                                   # ragged split: pad to the widest shard
    padded = y_local.new_zeros(width, y_local.shape[1])
    padded[:sizes[rank]] = y_local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def sharded_forward(forward: Callable[[torch.Tensor], torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """Run `forward` ((2b, N, C) -> (b, D)) on this rank's shard of the global batch `x`, return all outputs."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return forward(x)
    rank, world = dist.get_rank(), dist.get_world_size()
    return gather_outputs(forward(local_batch(x, rank, world)), x.shape[0] // 2)
