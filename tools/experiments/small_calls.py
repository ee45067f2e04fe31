#!/usr/bin/env python3
"""Latency of the host-buffer entry points for planner-sized calls (Panda, sphere cage)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vamp_mvt_amd as vamp  # noqa: E402
from vamp_mvt_amd.workloads import SPHERE_CAGE  # noqa: E402

vamp.set_device(0)
env = vamp.Environment()
for c in SPHERE_CAGE:
    env.add_sphere(vamp.Sphere(c, 0.2))
p = vamp.panda
rng = np.random.default_rng(0)
lo, hi = p.lower_bounds(), p.upper_bounds()
for n in (1, 8, 64, 1024):
    a = (lo + (hi - lo) * rng.random((n, 7), dtype=np.float32)).astype(np.float32)
    b = (a + rng.normal(0, 0.2, a.shape)).astype(np.float32)
    for name, fn in (("validate_batch", lambda: p.validate_batch(a, env)), ("validate_motion_batch", lambda: p.validate_motion_batch(a, b, env))):
        for _ in range(20):
            fn()
        t0 = time.perf_counter()
        for _ in range(300):
            fn()
        print(f"{name:22s} n={n:5d}: {(time.perf_counter() - t0) / 300 * 1e6:8.1f} us per call", flush=True)
