#!/usr/bin/env python3
"""Diagnostic timings of the other BASELINE.json configurations on one MI355X (not the headline metric; that is
bench.py).  One JSON line per configuration:

  config2  Panda, 1M configs vs 64 primitives                  (same as bench.py, for reference)
  config3  Fetch 8-DoF, 1M configs vs a 10k-point CAPT cloud
  config4  UR5, 1M edge validations vs 64 primitives           (single GPU share of the 8-GPU job)
  config5  Baxter 14-DoF, 262,144 edges, 32 primitives + a 10k-point CAPT cloud
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import vamp_mvt_amd as vamp  # noqa: E402
from vamp_mvt_amd.workloads import POINT_RADIUS, RADII, environment_from_spec, shell_cloud, shell_spec  # noqa: E402


def timed(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def uniform(mod, n, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    lo = torch.from_numpy(mod.lower_bounds()).cuda()
    hi = torch.from_numpy(mod.upper_bounds()).cuda()
    return (lo + (hi - lo) * torch.rand((n, mod.dimension()), generator=g, device="cuda")).contiguous()


def edges(mod, n, seed, dmin=0.2, dmax=1.5):
    a = uniform(mod, n, seed)
    g = torch.Generator(device="cuda").manual_seed(seed + 1)
    d = torch.randn((n, mod.dimension()), generator=g, device="cuda")
    d = d / d.norm(dim=1, keepdim=True)
    length = dmin + (dmax - dmin) * torch.rand((n, 1), generator=g, device="cuda")
    return a, (a + d * length).contiguous()


def main():
    which = sys.argv[1:] or ["config2", "config3", "config4", "config5"]
    vamp.set_device(0)
    torch.cuda.set_device(0)
    for cfg in which:
        if cfg == "config2":
            mod, n = vamp.panda, 1 << 20
            env = environment_from_spec(shell_spec(0))
            q = uniform(mod, n, 1)
            bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
            ms = timed(lambda: mod.validate_bits_device(q, env, bits))
            unit, frac = "checks/s", float(mod.validate_batch(q[:65536], env).float().mean())
        elif cfg == "config3":
            mod, n = vamp.fetch, 1 << 20
            env = environment_from_spec([("capt", (shell_cloud(10000, 3), *RADII["fetch"], POINT_RADIUS))])
            q = uniform(mod, n, 2)
            bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
            ms = timed(lambda: mod.validate_bits_device(q, env, bits), iters=3, warm=1)
            unit, frac = "checks/s", float(mod.validate_batch(q[:65536], env).float().mean())
        elif cfg == "config4":
            mod, n = vamp.ur5, 1 << 20
            env = environment_from_spec(shell_spec(0))
            a, b = edges(mod, n, 3)
            bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
            ms = timed(lambda: mod.validate_bits_device(a, env, bits, goals=b), iters=3, warm=1)
            unit, frac = "edges/s", float(mod.validate_motion_batch(a[:65536], b[:65536], env).float().mean())
        elif cfg == "config5":
            mod, n = vamp.baxter, 1 << 18
            spec = shell_spec(2, 16, 16, 0.9, 1.6) + [("capt", (shell_cloud(10000, 4, 1.0, 1.8), *RADII["baxter"],
                                                                POINT_RADIUS))]
            env = environment_from_spec(spec)
            a, b = edges(mod, n, 5, 0.1, 0.6)
            bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
            ms = timed(lambda: mod.validate_bits_device(a, env, bits, goals=b), iters=2, warm=1)
            unit, frac = "edges/s", float(mod.validate_motion_batch(a[:16384], b[:16384], env).float().mean())
        else:
            raise SystemExit(cfg)
        print(json.dumps({"config": cfg, "robot": mod._name, "n": n, "ms": ms, "value": n / (ms * 1e-3), "unit": unit,
                          "valid_fraction": frac}), flush=True)


if __name__ == "__main__":
    t0 = time.time()
    main()
    print(f"# total {time.time() - t0:.1f}s", flush=True)
