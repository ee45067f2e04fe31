#!/usr/bin/env python3
"""How long does vmv_env_finalize take with a CAPT cloud (upload or in-place use of the arrays + the query copy:
distance-sorted points, leaf records, blocked planes)?   python tools/experiments/capt_finalize_time.py"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import vamp_mvt_amd as vamp  # noqa: E402
from vamp_mvt_amd.workloads import POINT_RADIUS, RADII, environment_from_spec, shell_cloud  # noqa: E402

vamp.set_device(0)
for robot, n in (("panda", 10000), ("fetch", 10000), ("baxter", 10000), ("fetch", 65536)):
    k = 1.6 if robot == "baxter" else 1.0
    pts = shell_cloud(n, 3, 0.5 * k, 1.2 * k, 0.0, 1.5)
    for no_prefix in ("1", "0"):
        os.environ["VMV_CAPT_NO_PREFIX"] = no_prefix
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            env = environment_from_spec([("capt", (pts, *RADII[robot], POINT_RADIUS))])
            best = min(best, time.perf_counter() - t0)
        print(f"{robot} {n} points: build + finalize {best * 1e3:.1f} ms ({'plain copy' if no_prefix == '1' else 'distance-sorted copy'})", flush=True)
