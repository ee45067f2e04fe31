"""Multi-GPU sharding of configuration / edge batches (one process per GPU, torch.distributed over RCCL).

Every unit (configuration or edge) is independent given the read-only environment, so a batch is cut into
contiguous shards aligned to 64 units (= one wavefront = one 64-bit validity word; an edge is a whole sequence of
8-lane rakes, so no rake ever straddles a shard) and the only exchange step is an all-gather of the packed validity
words (N/8 bytes for the whole job).  The same arithmetic is exported for C callers as vmv_shard_range
(include/vamp_mvt_amd.h)."""
from __future__ import annotations

import os
import subprocess
import sys


def shard_range(n: int, rank: int, world: int):
    """[lo, hi) of rank's shard; every boundary except the last is a multiple of 64."""
    words = (n + 63) // 64
    per = (words + world - 1) // world
    lo = min(rank * per * 64, n)
    hi = min((rank + 1) * per * 64, n)
    return lo, hi


def gather_bits(local_words, n: int, world: int, device=None):
    """local_words: int64 tensor holding this rank's packed validity words (shard_range order; empty for a rank whose
    shard is empty).  Returns the int64 words of the whole batch on every rank (all_gather over the default process
    group).  `device`: where the collective's buffers live — it must be the SAME kind of device on every rank (the GPU
    of the rank under RCCL), also on a rank with nothing to contribute; default: the device of `local_words`."""
    import torch
    import torch.distributed as dist

    device = local_words.device if device is None else torch.device(device)
    words = (n + 63) // 64
    per = (words + world - 1) // world
    buf = torch.zeros(per, dtype=torch.int64, device=device)
    buf[: local_words.numel()] = local_words.to(device)
    if world == 1:
        return buf[:words]
    out = torch.empty(per * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, buf)
    return out[:words]


def validate_sharded(n: int, local_fn, rank: int, world: int, device=None):
    """The whole multi-GPU control flow of the path: this rank validates units [lo, hi) of an n-unit batch with
    `local_fn(lo, hi) -> int64 tensor of ceil((hi - lo) / 64) packed validity words` and every rank gets the words of
    the whole batch back (one all-gather, the path's only exchange step).  A batch smaller than 64 x world units leaves
    the last ranks without a shard (100 edges on 8 GPUs: ranks 2..7): they launch nothing and still enter the
    collective, with buffers on `device` (their GPU) like everyone else."""
    import torch

    lo, hi = shard_range(n, rank, world)
    if hi > lo:
        local = local_fn(lo, hi)
        if device is not None and local.device != torch.device(device):
            raise ValueError(f"local_fn returned words on {local.device}, the collective runs on {torch.device(device)}")
    else:
        local = torch.zeros(0, dtype=torch.int64, device=device)
    assert local.numel() == (hi - lo + 63) // 64
    return gather_bits(local, n, world, device=device)


def validate_batch_sharded(robot, configurations, environment, goals=None, rank=None, world=None):
    """<robot>.validate_batch / validate_motion_batch over all GPUs of the job (BASELINE configs 4 and 5).

    Call on every rank (one process per GPU, default process group = RCCL) with the SAME batch: `configurations` (and
    `goals` for edges) are [n][dim] float32 numpy arrays or CUDA tensors holding the whole batch; each rank uploads /
    reads only its 64-aligned shard, runs the kernels on its own GPU against its own copy of the environment and the
    packed validity words are all-gathered.  Returns the int64 validity words of the whole batch as a CUDA tensor on
    every rank (`vamp_mvt_amd.unpack_bits(words.cpu().numpy(), n)` -> bool[n])."""
    import torch
    import torch.distributed as dist

    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
    n = int(configurations.shape[0])
    dev = torch.device("cuda", torch.cuda.current_device())

    def shard(x, lo, hi):
        t = x[lo:hi] if isinstance(x, torch.Tensor) else torch.from_numpy(x[lo:hi])
        return t.to(dev, torch.float32).contiguous()

    def local_fn(lo, hi):
        bits = torch.zeros((hi - lo + 63) // 64, dtype=torch.int64, device=dev)
        robot.validate_bits_device(shard(configurations, lo, hi), environment, bits,
                                   goals=None if goals is None else shard(goals, lo, hi))
        return bits

    return validate_sharded(n, local_fn, rank, world, device=dev)


def respawn_one_rank_per_gpu(n_gpus: int, script: str, argv):
    """`python <script> --gpus N` started without a launcher: start N fresh ranks (python -m torch.distributed.run, one
    process per GPU, rendezvous on 127.0.0.1) as CHILD processes and return their exit code.  Must be called before
    anything in this process touches the GPU (no HIP call, no torch.cuda call): a process that has initialised the
    GPU must never be replaced or forked into ranks."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script, *argv]
    return subprocess.call(cmd, env=env)
