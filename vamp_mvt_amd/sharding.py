"""Multi-GPU sharding of configuration / edge batches (one process per GPU, torch.distributed over RCCL).

Every unit (configuration or edge) is independent given the read-only environment, so a batch is cut into
contiguous shards aligned to 64 units (= one wavefront = one 64-bit validity word; edges keep whole 8-lane rakes)
and the only exchange step is an all-gather of the packed validity words (N/8 bytes for the whole job)."""
from __future__ import annotations


def shard_range(n: int, rank: int, world: int):
    """[lo, hi) of rank's shard; every boundary except the last is a multiple of 64."""
    words = (n + 63) // 64
    per = (words + world - 1) // world
    lo = min(rank * per * 64, n)
    hi = min((rank + 1) * per * 64, n)
    return lo, hi


def gather_bits(local_words, n: int, world: int):
    """local_words: int64 tensor holding this rank's packed validity words (shard_range order).
    Returns the int64 words of the whole batch on every rank (all_gather over the default process group)."""
    import torch
    import torch.distributed as dist

    words = (n + 63) // 64
    per = (words + world - 1) // world
    buf = torch.zeros(per, dtype=torch.int64, device=local_words.device)
    buf[: local_words.numel()] = local_words
    if world == 1:
        return buf[:words]
    out = torch.empty(per * world, dtype=torch.int64, device=local_words.device)
    dist.all_gather_into_tensor(out, buf)
    return out[:words]
