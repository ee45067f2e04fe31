"""Synthetic workload definitions (BASELINE.json configs): canonical primitive parameters and
point clouds with fixed seeds, shared by bench.py and the tests.  Pure numpy; no collision code here."""
from __future__ import annotations

import numpy as np

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


SPHERE_CAGE = [  # problem data of the reference's scripts/sphere_cage_example.py:16-31 (radius 0.2)
    [0.55, 0, 0.25], [0.35, 0.35, 0.25], [0, 0.55, 0.25], [-0.55, 0, 0.25], [-0.35, -0.35, 0.25], [0, -0.55, 0.25],
    [0.35, -0.35, 0.25], [0.35, 0.35, 0.8], [0, 0.55, 0.8], [-0.35, 0.35, 0.8], [-0.55, 0, 0.8], [-0.35, -0.35, 0.8],
    [0, -0.55, 0.8], [0.35, -0.35, 0.8]]


WORKSPACE = {  # workspace boxes as src/vamp/pointcloud.py builds them: first-joint location +- max reach
    "panda": ([-1.19, -1.19, -0.857], [1.19, 1.19, 1.523]), "ur5": ([-1.2, -1.2, -0.29], [1.2, 1.2, 2.11]),
    "fetch": ([-1.4, -1.4, -1.0], [1.5, 1.4, 1.8]), "baxter": ([-1.5, -1.5, -1.2], [1.5, 1.5, 1.8])}


def environment_from_spec(spec):
    """spec: list of (kind, params) -> vamp_mvt_amd.Environment"""
    import vamp_mvt_amd as vamp

    e = vamp.Environment()
    for kind, p in spec:
        if kind == "sphere":
            e.add_sphere(vamp.Sphere(p[:3], p[3]))
        elif kind == "cuboid":
            e.add_cuboid(vamp.Cuboid.from_canonical(p))
        elif kind == "capsule":
            e.add_capsule(vamp.Cylinder.from_canonical(p))
        elif kind == "mvt":
            e.add_mvt_pointcloud(*p)
        elif kind == "attach":
            tf, spheres = p
            a = vamp.Attachment(tf)
            a.add_spheres([vamp.Sphere(sp[:3], sp[3]) for sp in spheres])
            e.attach(a)
        elif kind == "heightfield":
            center, scale, xd, yd, data = p
            e.add_heightfield(vamp.make_heightfield(center, scale, (xd, yd), data))
        else:
            e.add_capt_pointcloud(*p)
    return e
