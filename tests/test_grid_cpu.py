"""The broad-phase grid of the gate pass (vmv_grid_build.h) is an index, not part of the reference's algorithm:
it may only ever ADD candidates.  Property checked here on the CPU: whenever the oracle's exact predicate says a
query sphere (radius <= R) collides with primitive p, p's bit is set in the cell the device would look up
(cell arithmetic restated in fp32 exactly as vmv_device.h env_hit_grid does)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle_lib import Oracle

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


@pytest.fixture(scope="module")
def probe():
    out = os.path.join(ROOT, "build", "libgrid_probe.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = os.path.join(ROOT, "tests", "native", "grid_probe.cc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out, src])
    L = ctypes.CDLL(out)
    L.grid_probe_build.restype = ctypes.c_int
    L.grid_probe_build.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    return L


def _random_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _primitives(rng, n):
    """type (grid list id), 16 canonical floats, and a function adding the primitive to an oracle env"""
    types, params, adders = [], [], []
    for i in range(n):
        kind = i % 5
        c = rng.uniform(-1.2, 1.2, 3).astype(np.float32)
        p = np.zeros(16, np.float32)
        if kind == 0:
            p[:4] = [*c, rng.uniform(0.02, 0.3)]
            adders.append(lambda e, p=p: e.add_sphere(*[float(v) for v in p[:4]]))
        elif kind in (1, 2):
            v = rng.normal(size=3) * 0.4
            if kind == 2:
                v[:2] = 0.0
            v = v.astype(np.float32)
            r = np.float32(rng.uniform(0.02, 0.2))
            rdv = np.float32(1.0) / np.float32(np.dot(v, v))
            p[:8] = [*c, *v, r, rdv]
            adders.append(lambda e, p=p: e.add_capsule(p[:8]))
        else:
            R = _random_rotation(rng) if kind == 3 else np.eye(3)
            if kind == 4:  # z-aligned: rotation about z only, axis_3_z == 1 exactly
                a = rng.uniform(0, 2 * np.pi)
                R = np.array([[np.cos(a), np.sin(a), 0], [-np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
            p[:3] = c
            p[3:12] = R.astype(np.float32).ravel()
            p[12:15] = rng.uniform(0.02, 0.4, 3)
            adders.append(lambda e, p=p: e.add_cuboid(p[:15]))
        types.append(kind)
        params.append(p)
    return np.array(types, np.int32), np.stack(params), adders


@pytest.mark.parametrize("seed,R", [(1, 0.12), (2, 0.25), (3, 0.05)])
def test_grid_never_drops_a_colliding_primitive(probe, seed, R):
    rng = np.random.default_rng(seed)
    o = Oracle()
    n = 40
    types, params, adders = _primitives(rng, n)
    # the oracle classifies capsules/cuboids as z-aligned itself; the grid is given the same classes
    dims = np.zeros(3, np.uint32)
    origin = np.zeros(3, np.float32)
    inv_cell = np.zeros(1, np.float32)
    words = np.zeros(1, np.uint32)
    cells = np.zeros(40000 * 2, np.uint32)
    rc = probe.grid_probe_build(types.ctypes.data, params.ctypes.data, n, float(R), dims.ctypes.data,
                                origin.ctypes.data, inv_cell.ctypes.data, words.ctypes.data, cells.ctypes.data,
                                cells.size)
    assert rc == 1
    W = int(words[0])
    cells = cells[: int(dims.prod()) * W].reshape(int(dims[0]), int(dims[1]), int(dims[2]), W)
    envs = []
    for add in adders:
        e = o.env()
        add(e)
        envs.append(e)
    # z classification must agree with what the grid was told
    for t, e in zip(types, envs):
        assert e.counts()[:5] == [int(t == k) for k in (0, 1, 2, 3, 4)]

    hits = listed = 0
    nq = 6000
    # queries: uniform in a box larger than the grid, plus points pushed right next to primitive surfaces
    ctr = rng.uniform(-1.9, 1.9, (nq, 3)).astype(np.float32)
    rad = (R * rng.uniform(0.05, 1.0, nq)).astype(np.float32)
    rad[::7] = np.float32(R)
    for c, r in zip(ctr, rad):
        f = (c - origin) * inv_cell[0]  # fp32, as on the device
        inside = bool((f >= 0).all() and (f < dims.astype(np.float32)).all())
        idx = f.astype(np.uint32) if inside else None
        for p in range(n):
            if o.L.vo_sphere_environment_in_collision(envs[p].h, c.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                                      ctypes.c_float(float(r))):
                hits += 1
                assert inside, (c, r, p)
                assert (cells[idx[0], idx[1], idx[2], p // 32] >> (p % 32)) & 1, (c, r, p, types[p])
        if inside:
            listed += int(sum(bin(int(w)).count("1") for w in cells[idx[0], idx[1], idx[2]]))
    assert hits > 200  # the property was actually exercised
    assert listed < 0.6 * nq * n  # and the grid does prune
