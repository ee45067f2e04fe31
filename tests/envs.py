"""Synthetic environments built identically for the product (vamp_mvt_amd.Environment) and the oracle.

Test infrastructure.  Every primitive is specified by its canonical parameters so both sides receive the same
fp32 numbers (the euler->axes constructors are exercised separately, to tolerance)."""
from __future__ import annotations

import numpy as np

from oracle_lib import SPHERE_CAGE

RADII = {"panda": (0.012, 0.08), "ur5": (0.015, 0.08), "fetch": (0.012, 0.24), "baxter": (0.012, 0.5)}
POINT_RADIUS = 0.0025


def yaw_cuboid(center, yaw, half):
    c, s = np.float32(np.cos(yaw)), np.float32(np.sin(yaw))
    return np.array([*center, c, s, 0, -s, c, 0, 0, 0, 1, *half], np.float32)


def rot_cuboid(center, rpy, half):
    r, p, y = (float(v) for v in rpy)
    rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]])
    m = (rz @ ry @ rx).astype(np.float32)
    return np.array([*center, *m[:, 0], *m[:, 1], *m[:, 2], *half], np.float32)


def capsule(p1, p2, r):
    p1, p2 = np.asarray(p1, np.float32), np.asarray(p2, np.float32)
    v = (p2 - p1).astype(np.float32)
    dot = np.float32(np.float32(v[0] * v[0]) + np.float32(v[1] * v[1])) + np.float32(v[2] * v[2])
    return np.array([*p1, *v, r, np.float32(1.0 / float(dot))], np.float32)


def shell_spec(seed=0, n_spheres=32, n_cuboids=32, rmin=0.45, rmax=0.95):
    """BASELINE config 2 generator: cylindrical shell of spheres + z-aligned cuboids."""
    rng = np.random.default_rng(seed)
    spec = []
    for i in range(n_spheres + n_cuboids):
        ang = rng.uniform(0, 2 * np.pi)
        rad = rng.uniform(rmin, rmax)
        z = rng.uniform(0.0, 1.2)
        c = np.array([rad * np.cos(ang), rad * np.sin(ang), z], np.float32)
        if i < n_spheres:
            spec.append(("sphere", np.array([*c, rng.uniform(0.03, 0.08)], np.float32)))
        else:
            spec.append(("cuboid", yaw_cuboid(c, rng.uniform(0, 2 * np.pi), rng.uniform(0.03, 0.08, 3))))
    return spec


def shell_cloud(n, seed=0, rmin=0.6, rmax=1.2, zmin=0.2, zmax=1.5):
    rng = np.random.default_rng(seed)
    ang = rng.uniform(0, 2 * np.pi, n)
    rad = rng.uniform(rmin, rmax, n)
    z = rng.uniform(zmin, zmax, n)
    return np.stack([rad * np.cos(ang), rad * np.sin(ang), z], 1).astype(np.float32)


def spec_for(kind, robot="panda", seed=0):
    if kind == "empty":
        return []
    if kind == "cage":
        return [("sphere", np.array([*c, 0.2], np.float32)) for c in SPHERE_CAGE]
    if kind == "shell64":
        return shell_spec(seed)
    if kind == "mixed":
        rng = np.random.default_rng(seed + 11)
        spec = shell_spec(seed + 3, 6, 6)
        for _ in range(6):
            c = rng.uniform([-0.9, -0.9, 0.0], [0.9, 0.9, 1.3]).astype(np.float32)
            spec.append(("cuboid", rot_cuboid(c, rng.uniform(-1, 1, 3), rng.uniform(0.03, 0.12, 3))))
        for _ in range(6):
            p1 = rng.uniform([-0.9, -0.9, 0.0], [0.9, 0.9, 1.3]).astype(np.float32)
            p2 = (p1 + rng.uniform(-0.3, 0.3, 3)).astype(np.float32)
            spec.append(("capsule", capsule(p1, p2, rng.uniform(0.02, 0.08))))
        for _ in range(6):
            p1 = rng.uniform([-0.9, -0.9, 0.0], [0.9, 0.9, 1.0]).astype(np.float32)
            p2 = p1.copy()
            p2[2] += np.float32(rng.uniform(0.1, 0.5))
            spec.append(("capsule", capsule(p1, p2, rng.uniform(0.02, 0.08))))
        return spec
    if kind == "capt":
        r_min, r_max = RADII[robot]
        spec = shell_spec(seed + 5, 4, 4)
        spec.append(("capt", (shell_cloud(2000, seed + 7), r_min, r_max, POINT_RADIUS)))
        return spec
    raise KeyError(kind)


def build_oracle_env(o, spec):
    e = o.env()
    for kind, p in spec:
        if kind == "sphere":
            e.add_sphere(*p)
        elif kind == "cuboid":
            e.add_cuboid(p)
        elif kind == "capsule":
            e.add_capsule(p)
        else:
            e.add_capt(*p)
    return e


def build_product_env(spec):
    import vamp_mvt_amd as vamp

    e = vamp.Environment()
    for kind, p in spec:
        if kind == "sphere":
            e.add_sphere(vamp.Sphere(p[:3], p[3]))
        elif kind == "cuboid":
            e.add_cuboid(vamp.Cuboid.from_canonical(p))
        elif kind == "capsule":
            e.add_capsule(vamp.Cylinder.from_canonical(p))
        else:
            e.add_capt_pointcloud(*p)
    return e


def make_env(kind, o, robot="panda", seed=0):
    spec = spec_for(kind, robot, seed)
    return build_product_env(spec), build_oracle_env(o, spec)
