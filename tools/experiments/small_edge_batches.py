#!/usr/bin/env python3
"""Edge-validation time for planner-sized batches (UR5, 64 primitives): the launcher shrinks the per-workgroup edge
chunk for small batches so that the grid still fills the chip."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vamp_mvt_amd as vamp  # noqa: E402
from vamp_mvt_amd.workloads import environment_from_spec, shell_spec  # noqa: E402

vamp.set_device(0)
env = environment_from_spec(shell_spec(0))
mod = vamp.ur5
lo = torch.from_numpy(mod.lower_bounds()).cuda()
hi = torch.from_numpy(mod.upper_bounds()).cuda()
g = torch.Generator(device="cuda").manual_seed(3)
for n in (256, 2048, 16384, 131072, 1 << 20):
    a = (lo + (hi - lo) * torch.rand((n, 6), generator=g, device="cuda")).contiguous()
    d = torch.randn((n, 6), generator=g, device="cuda")
    b = (a + d / d.norm(dim=1, keepdim=True) * (0.2 + 1.3 * torch.rand((n, 1), generator=g, device="cuda"))).contiguous()
    bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    for _ in range(5):
        mod.validate_bits_device(a, env, bits, goals=b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 50 if n < 500000 else 10
    for _ in range(iters):
        mod.validate_bits_device(a, env, bits, goals=b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{n:8d} edges: {dt * 1e3:8.4f} ms  {n / dt:.3e} edges/s", flush=True)
