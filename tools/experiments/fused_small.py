#!/usr/bin/env python3
"""vmv_validate_batch on device buffers for small and shard-sized batches (Panda, 64-primitive shell): pipelined step time and
one call + synchronize.  Run once plain and once with VMV_FUSED_KERNEL=1 (read once per process) to compare the two-kernel
path with the fused one-FK kernel where a batch is far too small to fill the chip."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vamp_mvt_amd as vamp  # noqa: E402
from vamp_mvt_amd.workloads import environment_from_spec, shell_spec  # noqa: E402

vamp.set_device(0)
env = environment_from_spec(shell_spec(0))
mod = getattr(vamp, sys.argv[1] if len(sys.argv) > 1 else "panda")
lo = torch.from_numpy(mod.lower_bounds()).cuda()
hi = torch.from_numpy(mod.upper_bounds()).cuda()
g = torch.Generator(device="cuda").manual_seed(1)
tag = "fused" if os.environ.get("VMV_FUSED_KERNEL") else "two kernels"
for n in (64, 1024, 16384, 65536, 131072, 262144, 524288, 1 << 20):
    q = (lo + (hi - lo) * torch.rand((n, mod.dimension()), generator=g, device="cuda")).contiguous()
    bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    for _ in range(10):
        mod.validate_bits_device(q, env, bits)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        mod.validate_bits_device(q, env, bits)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(50):
        mod.validate_bits_device(q, env, bits)
        torch.cuda.synchronize()
    ds = (time.perf_counter() - t0) / 50
    print(f"{mod._name:6s} {tag:12s} n={n:8d}: {dt * 1e6:8.1f} us/step pipelined   {ds * 1e6:8.1f} us call+sync   valid {int(vamp.unpack_bits(bits.cpu().numpy().view('uint64'), n).sum())}", flush=True)
