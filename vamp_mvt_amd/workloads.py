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


def mixed_spec(seed=0, rmin=0.45, rmax=0.95):
    """64 primitives of every kind the sorted lists hold — 16 spheres + 16 z-aligned cuboids (shell_spec), 16 rotated
    cuboids, 8 capsules, 8 z-aligned capsules — what MotionBenchMaker-style scenes look like to the kernels (the
    five-list variant), next to BASELINE config 2's spheres and z-aligned cuboids.  Diagnostics workload, not a BASELINE one."""
    rng = np.random.default_rng(seed + 101)
    spec = shell_spec(seed, 16, 16, rmin, rmax)

    def pos(zmax=1.3):
        ang, rad = rng.uniform(0, 2 * np.pi), rng.uniform(rmin, rmax)
        return np.array([rad * np.cos(ang), rad * np.sin(ang), rng.uniform(0.0, zmax)], np.float32)

    for _ in range(16):
        spec.append(("cuboid", rot_cuboid(pos(), rng.uniform(-1, 1, 3), rng.uniform(0.03, 0.12, 3))))
    for _ in range(8):
        p1 = pos()
        spec.append(("capsule", capsule(p1, (p1 + rng.uniform(-0.3, 0.3, 3)).astype(np.float32), rng.uniform(0.02, 0.08))))
    for _ in range(8):
        p1 = pos(1.0)
        p2 = p1.copy()
        p2[2] += np.float32(rng.uniform(0.1, 0.5))
        spec.append(("capsule", capsule(p1, p2, rng.uniform(0.02, 0.08))))
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


# ------------------------------------------------------------------------------------------------------------------
# Roadmap-shaped edge batches (BASELINE configs 4 and 5).  The reference's planners only ever ask
# about edges between VALID samples: PRM connects a new valid sample to its nearest valid neighbours
# (planning/prm.hh:109-145), FCIT* validates edges of the graph over the valid samples (planning/fcit.hh:137-260).
# These generators build such batches on the device with the product's own kernels (Halton samples from
# vmv_halton_configs, validity from vmv_validate_batch); they are workload SETUP and run outside every timed region.
# torch is plumbing here (device arrays, RNG, top-k).
# ------------------------------------------------------------------------------------------------------------------
def valid_halton_samples(mod, env, want, max_draw=1_000_000, chunk=1 << 18):
    """-> torch float32 CUDA [<= want][dim]: the first `want` VALID samples of the reference's Halton sequence
    (vamp.<robot>.halton(), generated on the device), in sequence order."""
    import torch

    kept, have, skip = [], 0, 0
    while have < want and skip < max_draw:
        n = min(chunk, max_draw - skip)
        q = mod.halton_device(n, skip)
        ok = mod.validate_batch(q, env)
        q = q[ok]
        kept.append(q)
        have += q.shape[0]
        skip += n
    out = torch.cat(kept)[:want] if kept else None
    if out is None or out.shape[0] == 0:
        raise RuntimeError("no valid Halton sample in this environment")
    return out.contiguous()


def prm_shaped_edges(mod, env, n, dmin, dmax, seed, max_rounds=64):
    """PRM-roadmap shaped batch (config 4): every edge joins a valid Halton sample to a VALID neighbour configuration at
    distance U[dmin, dmax] rad in a uniform random direction (rejection sampling on the neighbour's validity; SURVEY.md
    §8d-4: "valid Halton samples paired with a neighbour at distance U[0.2, 1.5]").  -> (start, goal) CUDA [n][dim]."""
    import torch

    samples = valid_halton_samples(mod, env, n)
    dev = samples.device
    start = samples[torch.arange(n, device=dev) % samples.shape[0]].contiguous()  # sample i serves edges i, i + V, ...
    goal = torch.empty_like(start)
    todo = torch.arange(n, device=dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    for _ in range(max_rounds):
        if todo.numel() == 0:
            break
        d = torch.randn((todo.numel(), start.shape[1]), generator=g, device=dev)
        d = d / d.norm(dim=1, keepdim=True)
        length = dmin + (dmax - dmin) * torch.rand((todo.numel(), 1), generator=g, device=dev)
        cand = (start[todo] + d * length).contiguous()
        ok = mod.validate_batch(cand, env)
        goal[todo[ok]] = cand[ok]
        todo = todo[~ok]
    if todo.numel():  # a start boxed in on all sides: give its edge to a start that found neighbours
        done = torch.ones(n, dtype=torch.bool, device=dev)
        done[todo] = False
        src = torch.nonzero(done).flatten()
        if src.numel() == 0:
            raise RuntimeError("no valid neighbour found for any sample")
        pick = src[torch.arange(todo.numel(), device=dev) % src.numel()]
        start[todo], goal[todo] = start[pick], goal[pick]
    return start.contiguous(), goal.contiguous()


def knn_shaped_edges(mod, env, n, k, seed=0, block=4096):
    """FCIT*-shaped batch (config 5): the n / k first valid Halton samples, each joined to its k nearest valid samples
    (L2 in joint space, as the reference's distance).  -> (start, goal) CUDA [n][dim]; n must be a multiple of k."""
    import torch

    v = n // k
    samples = valid_halton_samples(mod, env, v)
    v = samples.shape[0]
    nbr = []
    for lo in range(0, v, block):
        dist = torch.cdist(samples[lo:lo + block], samples)
        dist[torch.arange(dist.shape[0]), torch.arange(lo, lo + dist.shape[0])] = float("inf")  # not itself
        nbr.append(dist.topk(min(k, v - 1), largest=False).indices)
    nbr = torch.cat(nbr)
    start = samples[:, None, :].expand(-1, nbr.shape[1], -1).reshape(-1, samples.shape[1])
    goal = samples[nbr.reshape(-1)]
    reps = (n + start.shape[0] - 1) // start.shape[0]  # fewer valid samples than asked for: repeat the batch
    return start.repeat(reps, 1)[:n].contiguous(), goal.repeat(reps, 1)[:n].contiguous()
