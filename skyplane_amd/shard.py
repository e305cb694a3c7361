"""Multi-GPU sharding of the chunk queue: chunks are independent, so N GPUs = N ranks, chunk i
goes to rank i % N, and there is NO collective on the data path.  torch.distributed is used only around the
timed region (barrier + max over ranks), exactly as bench.py needs."""
from __future__ import annotations

import time
from typing import Callable, List, Tuple


def shard_indices(n_chunks: int, rank: int, world: int) -> List[int]:
    """Round-robin: chunk_index % world == rank."""
    assert 0 <= rank < world
    return list(range(rank, n_chunks, world))


def timed_region(fn: Callable[[], None], steps: int, dist=None, sync: Callable[[], None] = lambda: None) -> Tuple[float, float]:
    """Run fn() `steps` times between barriers; returns (local_seconds, max_over_ranks_seconds)."""
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    local = time.perf_counter() - t0
    worst = local
    if dist is not None:
        import torch

        t = torch.tensor([local], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        worst = float(t.item())
    return local, worst
