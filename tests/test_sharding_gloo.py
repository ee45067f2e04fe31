"""Multi-rank driver logic on CPU: contiguous shards + all-gather of packed validity words (gloo, world 2).

The data path of a shard is the GPU kernel (not available here).  What is under test is everything around it that
bench.py, tools/bench_configs.py and vamp_mvt_amd.sharding.validate_batch_sharded use at N > 1: the 64-aligned
contiguous shards, the all_gather_into_tensor of int64 words and the reassembly — first with a stand-in bit pattern,
then with REAL edge answers: each rank validates its shard of one batch of edges with the oracle (standing in for the
kernel launch on its GPU), and the gathered words must equal the unsharded answer on every rank."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vamp_mvt_amd.sharding import gather_bits, shard_range, validate_sharded  # noqa: E402


def _truth(n):
    i = np.arange(n, dtype=np.int64)
    return ((i * 2654435761) >> 7) % 3 != 0


def _pack(valid):
    return torch.from_numpy(np.packbits(np.pad(valid, (0, (-len(valid)) % 64)), bitorder="little").view(np.int64).copy())


def _unpack(words, n):
    return np.unpackbits(words.numpy().view(np.uint8), bitorder="little")[:n].astype(bool)


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker(rank, world, port, n, out):
    _init(rank, world, port)
    lo, hi = shard_range(n, rank, world)
    full = gather_bits(_pack(_truth(n)[lo:hi]), n, world)
    out[rank] = bool(np.array_equal(_unpack(full, n), _truth(n)))
    dist.destroy_process_group()


def _edge_problem(n):
    """one batch of Panda edges in the 64-primitive shell (mixed valid / invalid), the same on every rank"""
    from envs import build_oracle_env, spec_for
    from oracle_lib import Oracle
    from workmix import mixed_edges

    o = Oracle()
    env = build_oracle_env(o, spec_for("shell64", "panda"))
    rid, a, b, want = mixed_edges(o, "panda", env, n, seed=4242, zero_every=9)
    return o, rid, env, a, b, want


def _edge_worker(rank, world, port, n, out):
    _init(rank, world, port)
    o, rid, env, a, b, want = _edge_problem(n)
    launches = []

    def local_fn(lo, hi):  # stands in for robot.validate_bits_device on this rank's GPU
        launches.append((lo, hi))
        return _pack(o.validate_motion_batch(rid, env, a[lo:hi], b[lo:hi]))

    words = validate_sharded(n, local_fn, rank, world)
    lo, hi = shard_range(n, rank, world)
    out[rank] = bool(np.array_equal(_unpack(words, n), want)) and launches == ([(lo, hi)] if hi > lo else []) and \
        0.05 * n < want.sum() < 0.95 * n
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawn(fn, n, world=2):
    out = mp.Manager().dict()
    mp.spawn(fn, args=(world, _free_port(), n, out), nprocs=world, join=True)
    return out


@pytest.mark.parametrize("n", [1 << 12, 100_003, 130])
def test_two_rank_allgather_of_bitmasks(n):
    out = _spawn(_worker, n)
    assert out[0] and out[1]


@pytest.mark.parametrize("n", [1000, 64, 40])
def test_two_rank_sharded_edge_validation(n):
    """BASELINE configs 4 / 5 control flow at world size 2: whole rakes per shard, real edge answers, ragged tails
    (n = 40: the second rank's shard is empty and still takes part in the exchange)"""
    out = _spawn(_edge_worker, n)
    assert out[0] and out[1]


def test_empty_shard_enters_the_collective_on_the_ranks_device(monkeypatch):
    """a rank whose shard is empty (n = 40 or 100 units on 8 GPUs) used to build CPU buffers and hand them to
    the RCCL all-gather.  Every tensor given to the collective must live on the device the caller names — here the
    `meta` device, which no default could produce — on ranks with and without a shard."""
    seen = []

    def fake_all_gather(out, buf):
        seen.append((out.device.type, buf.device.type, out.numel(), buf.numel()))

    monkeypatch.setattr(dist, "all_gather_into_tensor", fake_all_gather)
    for n, world in ((40, 2), (100, 8), (1000, 8)):
        for rank in range(world):
            lo, hi = shard_range(n, rank, world)
            words = validate_sharded(n, lambda a, b: torch.zeros((b - a + 63) // 64, dtype=torch.int64, device="meta"),
                                     rank, world, device="meta")
            per = ((n + 63) // 64 + world - 1) // world
            assert seen[-1] == ("meta", "meta", per * world, per), (n, world, rank, lo, hi)
            assert words.device.type == "meta" and words.numel() == (n + 63) // 64
    with pytest.raises(ValueError):  # a local_fn that answers on another device than the collective's is refused
        validate_sharded(1000, lambda a, b: torch.zeros((b - a + 63) // 64, dtype=torch.int64), 0, 2, device="meta")


def test_shard_ranges_are_wave_aligned_and_match_the_c_abi(vamp):
    for n in (1, 63, 64, 65, 1 << 20, 999_999):
        for world in (1, 2, 4, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for (a, b), (c, d) in zip(edges, edges[1:]):
                assert b == c and (b % 64 == 0 or b == n)
            for r, (a, b) in enumerate(edges):  # include/vamp_mvt_amd.h: vmv_shard_range / vmv_shard_words
                lo, hi = ctypes.c_size_t(0), ctypes.c_size_t(0)
                assert vamp.lib.vmv_shard_range(n, r, world, ctypes.byref(lo), ctypes.byref(hi)) == 0
                assert (lo.value, hi.value) == (a, b)
            assert vamp.lib.vmv_shard_words(n, world) == ((n + 63) // 64 + world - 1) // world
    assert vamp.lib.vmv_shard_range(10, 2, 2, ctypes.byref(ctypes.c_size_t()), ctypes.byref(ctypes.c_size_t())) != 0


def test_bench_scripts_start_their_own_ranks():
    """`python bench.py --gpus N` without a launcher must become the launcher before anything touches the GPU: the
    package (which loads the HIP library) is not imported on that path, and the ranks are child processes"""
    import re
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    for script in ("bench.py", os.path.join("tools", "bench_configs.py")):
        text = open(os.path.join(root, script)).read()
        head = text[:text.index("respawn_one_rank_per_gpu(")]
        assert not re.search(r"^\s*import (torch|vamp_mvt_amd)|^\s*from (torch|vamp_mvt_amd)", head.split("def main")[1], re.M), script
        assert "os.exec" not in text
