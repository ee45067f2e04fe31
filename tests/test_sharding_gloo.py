"""Multi-rank driver logic on CPU: contiguous shards + all-gather of packed validity words (gloo, world 2).

The data path of a shard is the GPU kernel (not available here), so each rank packs a deterministic stand-in
bit pattern; what is under test is the sharding arithmetic and the exchange step bench.py / the multi-GPU
driver use: 64-aligned contiguous shards, all_gather_into_tensor of int64 words, reassembly."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vamp_mvt_amd.sharding import gather_bits, shard_range  # noqa: E402


def _truth(n):
    i = np.arange(n, dtype=np.int64)
    return ((i * 2654435761) >> 7) % 3 != 0


def _worker(rank, world, port, n, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n, rank, world)
    local = _truth(n)[lo:hi]
    words = np.packbits(np.pad(local, (0, (-len(local)) % 64)), bitorder="little").view(np.int64)
    full = gather_bits(torch.from_numpy(words.copy()), n, world)
    got = np.unpackbits(full.numpy().view(np.uint8), bitorder="little")[:n].astype(bool)
    out[rank] = bool(np.array_equal(got, _truth(n)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1 << 12, 100_003, 130])
def test_two_rank_allgather_of_bitmasks(n):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, n, out), nprocs=2, join=True)
    assert out[0] and out[1]


def test_shard_ranges_are_wave_aligned():
    for n in (1, 63, 64, 65, 1 << 20, 999_999):
        for world in (1, 2, 4, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for (a, b), (c, d) in zip(edges, edges[1:]):
                assert b == c and (b % 64 == 0 or b == n)
