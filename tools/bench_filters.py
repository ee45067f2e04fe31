#!/usr/bin/env python3
"""Developer bench on a GPU box: point-cloud filters and CAPT build, HIP path vs the CPU oracle (single core).
Prints one JSON line per case (not the headline metric)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import vamp_mvt_amd as vamp  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from test_filters import HI, LO, ORIGIN, RANGE, scene_cloud  # noqa: E402


def best_of(fn, reps=5):
    out = []
    for _ in range(reps):
        t = time.perf_counter()
        r = fn()
        out.append(time.perf_counter() - t)
    return min(out), r


def main():
    o = Oracle()
    for n in (50_000, 300_000, 1_000_000):
        pc = scene_cloud(n, 31)
        for ftype, args in (("scdf", dict(min_dist=0.02, voxel=0.03)), ("centervox", dict(min_dist=0.0, voxel=0.0303))):
            vamp.filter_pointcloud(pc, args["min_dist"], RANGE, args["voxel"], ORIGIN, LO, HI, True, ftype)  # warm up
            wall, dev = [], []
            for _ in range(5):
                pts, ns, dns = vamp.filter_pointcloud(pc, args["min_dist"], RANGE, args["voxel"], ORIGIN, LO, HI, True,
                                                      ftype, return_device_time=True)
                wall.append(ns)
                dev.append(dns)
            if ftype == "scdf":
                cpu, ref = best_of(lambda: o.filter_scdf(pc, args["min_dist"], RANGE, ORIGIN, LO, HI, True), 3)
            else:
                cpu, ref = best_of(lambda: o.filter_centervox(pc, args["voxel"], RANGE, ORIGIN, LO, HI), 3)
            print(json.dumps({"case": f"filter_{ftype}", "points_in": n, "points_out": int(len(pts)),
                              "gpu_ms_wall_incl_transfers": min(wall) / 1e6, "gpu_ms_device": min(dev) / 1e6,
                              "cpu_oracle_ms_1core": cpu * 1e3, "identical": bool(np.array_equal(pts, ref))}), flush=True)


def capt_cases():
    from vamp_mvt_amd.workloads import RADII, shell_cloud
    for robot, n in (("panda", 10_000), ("fetch", 10_000), ("baxter", 10_000), ("panda", 65_536), ("fetch", 65_536)):
        r_min, r_max = RADII[robot]
        pts = shell_cloud(n, 51)
        e = vamp.Environment()
        e.add_capt_pointcloud(pts, r_min, r_max, vamp.POINT_RADIUS, build="gpu")  # warm up
        gpu = [vamp.Environment().add_capt_pointcloud(pts, r_min, r_max, vamp.POINT_RADIUS, build="gpu", return_device_time=True)
               for _ in range(3)]
        host = [vamp.Environment().add_capt_pointcloud(pts, r_min, r_max, vamp.POINT_RADIUS) for _ in range(2)]
        t = vamp.Environment()
        t.add_capt_pointcloud(pts, r_min, r_max, vamp.POINT_RADIUS, build="gpu")
        print(json.dumps({"case": "capt_build", "robot_radii": robot, "points": n,
                          "affordance_vectors": int(t.host_tables()["capt"][0]["aff"].shape[1]),
                          "gpu_ms_device": min(g[1] for g in gpu) / 1e6,
                          "gpu_ms_wall_incl_upload": min(g[0] for g in gpu) / 1e6,
                          "host_builder_ms_1core": min(host) / 1e6}), flush=True)


if __name__ == "__main__":
    capt_cases()
    main()
