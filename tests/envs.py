"""Synthetic environments built identically for the product (vamp_mvt_amd.Environment) and the oracle.

Test infrastructure.  Every primitive is specified by its canonical parameters so both sides receive the same
fp32 numbers (the euler->axes constructors are exercised separately, to tolerance)."""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle_lib import SPHERE_CAGE  # noqa: E402

from vamp_mvt_amd.workloads import (POINT_RADIUS, RADII, WORKSPACE, capsule, environment_from_spec,  # noqa: F401
                                    rot_cuboid, shell_cloud, shell_spec, yaw_cuboid)


# The larger robots need room: Fetch's base and Baxter's torso fill the core of the Panda-sized scenes, where no
# configuration at all is valid (a degenerate test).  Obstacle positions are pushed outwards in x, y for them.
XY_SCALE = {"panda": 1.0, "ur5": 1.0, "fetch": 1.7, "baxter": 2.2}
SHELL = {"panda": (0.45, 0.95), "ur5": (0.45, 0.95), "fetch": (0.6, 1.2), "baxter": (0.9, 1.6)}


def _spread(p, robot, n_points=1):
    """scales the x, y of the first n_points points of a parameter vector (a no-op for panda / ur5)"""
    k = np.float32(XY_SCALE[robot])
    p = np.array(p, np.float32)
    for i in range(n_points):
        p[3 * i] *= k
        p[3 * i + 1] *= k
    return p


def spec_for(kind, robot="panda", seed=0):
    if kind == "empty":
        return []
    if kind == "cage":
        return [("sphere", _spread([*c, 0.2], robot)) for c in SPHERE_CAGE]
    if kind == "shell64":
        return shell_spec(seed, 32, 32, *SHELL[robot])
    if kind == "mixed":
        rng = np.random.default_rng(seed + 11)
        spec = shell_spec(seed + 3, 6, 6, *SHELL[robot])
        for _ in range(6):
            c = _spread(rng.uniform([-0.9, -0.9, 0.0], [0.9, 0.9, 1.3]), robot)
            spec.append(("cuboid", rot_cuboid(c, rng.uniform(-1, 1, 3), rng.uniform(0.03, 0.12, 3))))
        for _ in range(6):
            p1 = _spread(rng.uniform([-0.9, -0.9, 0.0], [0.9, 0.9, 1.3]), robot)
            p2 = (p1 + rng.uniform(-0.3, 0.3, 3)).astype(np.float32)
            spec.append(("capsule", capsule(p1, p2, rng.uniform(0.02, 0.08))))
        for _ in range(6):
            p1 = _spread(rng.uniform([-0.9, -0.9, 0.0], [0.9, 0.9, 1.0]), robot)
            p2 = p1.copy()
            p2[2] += np.float32(rng.uniform(0.1, 0.5))
            spec.append(("capsule", capsule(p1, p2, rng.uniform(0.02, 0.08))))
        return spec
    if kind == "many":  # more than 64 primitives per list: exercises the multi-chunk live-prefix count
        return shell_spec(seed + 21, 150, 100, 0.35, 1.1)
    if kind == "capt":
        r_min, r_max = RADII[robot]
        spec = shell_spec(seed + 5, 4, 4, *SHELL[robot])
        k = XY_SCALE[robot] if robot == "baxter" else 1.0
        spec.append(("capt", (shell_cloud(2000, seed + 7, 0.6 * k, 1.2 * k), r_min, r_max, POINT_RADIUS)))
        return spec
    if kind == "clouds":  # two point clouds and nothing else: the kernels' cloud-only variant (vmv_device.h kEnvClouds)
        r_min, r_max = RADII[robot]
        k = XY_SCALE[robot] if robot == "baxter" else 1.0
        return [("capt", (shell_cloud(1500, seed + 9, 0.6 * k, 1.2 * k), r_min, r_max, POINT_RADIUS)),
                ("capt", (shell_cloud(700, seed + 10, 0.9 * k, 1.5 * k), r_min, r_max, 2 * POINT_RADIUS))]
    if kind == "config3":  # BASELINE config 3: Fetch vs a 10,000-point CAPT cloud (tools/bench_configs.py, same generator)
        return [("capt", (shell_cloud(10000, 3), *RADII[robot], POINT_RADIUS))]
    if kind == "config5":  # BASELINE config 5: Baxter, 32 primitives + a 10,000-point CAPT cloud (tools/bench_configs.py)
        return shell_spec(2, 16, 16, 0.9, 1.6) + [("capt", (shell_cloud(10000, 4, 1.0, 1.8), *RADII[robot], POINT_RADIUS))]
    if kind == "mvt":  # the fork's Multi-level Voxel Table + a few primitives
        r_min, r_max = RADII[robot]
        lo, hi = WORKSPACE[robot]
        # the reference sizes the table's pools for at most 10 % occupied voxels / 50 % occupied columns, so the
        # coarse grids of the large-radius robots only take small clouds (larger ones abort there)
        if robot in ("panda", "ur5"):
            pts = shell_cloud(1500, seed + 9, 0.5, 1.0, 0.0, 1.2)
        elif robot == "fetch":
            pts = shell_cloud(200, seed + 9, 0.6, 0.8, 0.4, 0.9)
        else:
            pts = shell_cloud(60, seed + 9, 0.9, 1.0, 0.3, 0.6)
            pts = pts[pts[:, 0] > 0.5]
        spec = shell_spec(seed + 5, 3, 3)
        spec.append(("mvt", (pts, r_min, r_max, lo, hi, POINT_RADIUS)))
        return spec
    if kind in ("attach", "attach_free"):  # a held object: spheres attached at a frame relative to the end effector
        rng = np.random.default_rng(seed + 17)
        a = 0.4
        tf = np.array([[np.cos(a), -np.sin(a), 0, 0.02], [np.sin(a), np.cos(a), 0, -0.01], [0, 0, 1, 0.06], [0, 0, 0, 1]],
                      np.float32)
        # a 25 cm rod along the attachment's z axis + a cross piece: 11 spheres (more than one slab chunk)
        sp = [[0, 0, z, 0.03] for z in np.linspace(0.0, 0.25, 7)] + [[x, 0, 0.25, 0.025] for x in (-0.1, -0.05, 0.05, 0.1)]
        spec = shell_spec(seed + 5, 5, 5) if kind == "attach" else []
        spec.append(("attach", (tf, np.array(sp, np.float32))))
        return spec
    if kind == "heightfield":  # terrain under/around the robot + a few primitives (sphere_heightfield.hh)
        rng = np.random.default_rng(seed + 13)
        xd, yd = 48, 48
        gx, gy = np.meshgrid(np.arange(xd), np.arange(yd))
        rr = np.hypot(gx - xd / 2, gy - yd / 2) / (xd / 2)
        img = 0.85 * rr ** 2 + 0.12 * np.sin(gx / 3.0) * np.cos(gy / 4.0) + 0.05 * rng.random((yd, xd))
        img = np.clip(img, 0.0, 1.0).astype(np.float32)
        # a bowl around the robot: the image spans 2.4 m x 2.4 m (scale = metres per pixel), floor under the robot's base, rim 1.4 m higher;
        # small enough that lanes also run off the image on every side (the clamped border cells)
        spec = shell_spec(seed + 5, 3, 3)
        spec.append(("heightfield", (np.array([0.1, -0.05, {"panda": -0.3, "ur5": 0.1, "fetch": -0.42, "baxter": -1.38}[robot]], np.float32),
                                     np.array([2.4 / xd, 2.4 / yd, 1.0 / 1.6], np.float32), xd, yd, img.reshape(-1))))
        return spec
    raise KeyError(kind)


def counted_spec(robot, counts, seed=0):
    """small primitives scattered through the robot's workspace, exactly counts = (spheres, capsules, z-capsules, cuboids,
    z-cuboids) of them: list lengths chosen by the caller (the candidate-word layouts of vmv_api.hip: finalize)"""
    rng = np.random.default_rng(seed)
    rmin, rmax = SHELL[robot]
    ns, nc, nzc, nb, nzb = counts

    def pos(zmax=1.4):
        ang, rad = rng.uniform(0, 2 * np.pi), rng.uniform(0.8 * rmin, 1.1 * rmax)
        return np.array([rad * np.cos(ang), rad * np.sin(ang), rng.uniform(-0.1, zmax)], np.float32)

    spec = [("sphere", np.array([*pos(), rng.uniform(0.01, 0.06)], np.float32)) for _ in range(ns)]
    for _ in range(nc):
        p1 = pos()
        spec.append(("capsule", capsule(p1, (p1 + rng.uniform(-0.2, 0.2, 3)).astype(np.float32), rng.uniform(0.01, 0.04))))
    for _ in range(nzc):
        p1 = pos(1.0)
        p2 = p1.copy()
        p2[2] += np.float32(rng.uniform(0.05, 0.3))
        spec.append(("capsule", capsule(p1, p2, rng.uniform(0.01, 0.04))))
    for _ in range(nb):
        spec.append(("cuboid", rot_cuboid(pos(), rng.uniform(-1, 1, 3), rng.uniform(0.01, 0.06, 3))))
    for _ in range(nzb):
        spec.append(("cuboid", yaw_cuboid(pos(), rng.uniform(0, 2 * np.pi), rng.uniform(0.01, 0.06, 3))))
    return spec


def build_oracle_env(o, spec):
    e = o.env()
    for kind, p in spec:
        if kind == "sphere":
            e.add_sphere(*p)
        elif kind == "cuboid":
            e.add_cuboid(p)
        elif kind == "capsule":
            e.add_capsule(p)
        elif kind == "attach":
            e.attach(*p)
        elif kind == "heightfield":
            e.add_heightfield(*p)
        elif kind == "mvt":
            rc = e.add_mvt(*p)
            if rc != 0:
                raise ValueError(f"oracle MVT build failed with reason {rc}")
        else:
            e.add_capt(*p)
    return e


def build_product_env(spec):
    return environment_from_spec(spec)


def make_env(kind, o, robot="panda", seed=0):
    spec = spec_for(kind, robot, seed)
    return build_product_env(spec), build_oracle_env(o, spec)
