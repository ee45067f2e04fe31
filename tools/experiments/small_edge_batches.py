#!/usr/bin/env python3
"""Edge-validation time for planner-sized batches and for the shards of the 1M-edge job (UR5, 64 primitives; BASELINE
config 4 cut 8 / 4 / 2 ways), on both workload shapes: `prm` = valid Halton samples joined to valid neighbours at
U[0.2, 1.5] rad, `uniform` = the uniform, mostly invalid starts of rounds 1-2.
Steps are submitted back to back on one stream (kernels of one stream run in order), so ms per step is the per-call
device time; `sync` is one call followed by a synchronize (what a planner loop sees, launch + wait included)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vamp_mvt_amd as vamp  # noqa: E402
from vamp_mvt_amd.workloads import environment_from_spec, prm_shaped_edges, shell_spec  # noqa: E402

vamp.set_device(0)
env = environment_from_spec(shell_spec(0))
mod = getattr(vamp, os.environ.get("SMALL_ROBOT", "ur5"))  # (SMALL_ROBOT=panda: the other robot with a fused task kernel)
lo = torch.from_numpy(mod.lower_bounds()).cuda()
hi = torch.from_numpy(mod.upper_bounds()).cuda()
g = torch.Generator(device="cuda").manual_seed(3)
N = int(os.environ.get("SMALL_N", 1 << 20))
pa, pb = prm_shaped_edges(mod, env, N, 0.2, 1.5, seed=3)
ua = (lo + (hi - lo) * torch.rand((N, len(lo)), generator=g, device="cuda")).contiguous()
d = torch.randn((N, len(lo)), generator=g, device="cuda")
ub = (ua + d / d.norm(dim=1, keepdim=True) * (0.2 + 1.3 * torch.rand((N, 1), generator=g, device="cuda"))).contiguous()
sizes = [int(x) for x in sys.argv[1:]] or [256, 2048, 16384, 131072, 262144, 524288, N]
modes = os.environ.get("EDGE_MODES", "default").split(",")  # VMV_EDGE_TASKS values to compare (default = by batch size)
for shape, A, B, mode in ((s, x, y, m) for (s, x, y) in (("prm", pa, pb), ("uniform", ua, ub)) for m in modes):
    if mode == "default":
        os.environ.pop("VMV_EDGE_TASKS", None)
    else:
        os.environ["VMV_EDGE_TASKS"] = mode
    for n in sizes:
        a, b = A[:n].contiguous(), B[:n].contiguous()
        bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
        for _ in range(5):
            mod.validate_bits_device(a, env, bits, goals=b)
        torch.cuda.synchronize()
        iters = 50 if n < 500000 else 10
        t0 = time.perf_counter()
        for _ in range(iters):
            mod.validate_bits_device(a, env, bits, goals=b)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        t0 = time.perf_counter()
        for _ in range(10):
            mod.validate_bits_device(a, env, bits, goals=b)
            torch.cuda.synchronize()
        ds = (time.perf_counter() - t0) / 10
        valid = float(vamp.unpack_bits(bits.cpu().numpy().view("uint64"), n).mean())
        print(f"{shape:8s} mode {mode:8s} {n:8d} edges: {dt * 1e3:8.4f} ms/step  {n / dt:.3e} edges/s   sync {ds * 1e3:8.4f} ms   valid {valid:.3f}",
              flush=True)
