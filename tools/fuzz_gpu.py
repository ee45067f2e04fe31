#!/usr/bin/env python3
"""Randomised parity soak on the GPU box: random environments (every primitive kind, lists beyond 64 entries, CAPT clouds
of odd sizes, heightfields, attachments, ill-formed primitives) x robots x configurations / edges / free spheres, HIP path
vs the CPU oracle (test infrastructure), bit for bit.  Prints one line per case; exits non-zero at the first mismatch with
the seed that reproduces it.

    python tools/fuzz_gpu.py [--minutes 8] [--seed 0]"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vamp_mvt_amd as vamp  # noqa: E402
from envs import build_oracle_env, build_product_env  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from vamp_mvt_amd.workloads import POINT_RADIUS, RADII, capsule, rot_cuboid, shell_cloud, yaw_cuboid  # noqa: E402

ROBOTS = ["panda", "ur5", "fetch", "baxter"]
REACH = {"panda": 1.0, "ur5": 1.1, "fetch": 1.3, "baxter": 1.5}


def random_spec(rng, robot):
    R = REACH[robot]
    spec = []

    def pos(rmin=0.25):
        while True:
            p = rng.uniform([-R, -R, -0.2], [R, R, 1.6])
            if np.hypot(p[0], p[1]) > rmin * (2.0 if robot in ("fetch", "baxter") else 1.0):
                return p.astype(np.float32)

    many = rng.random() < 0.2
    for _ in range(int(rng.integers(0, 90 if many else 30))):
        spec.append(("sphere", np.array([*pos(), rng.uniform(0.01, 0.15)], np.float32)))
    for _ in range(int(rng.integers(0, 70 if many else 16))):
        spec.append(("cuboid", yaw_cuboid(pos(), rng.uniform(0, 2 * np.pi), rng.uniform(0.02, 0.15, 3))))
    for _ in range(int(rng.integers(0, 12))):
        spec.append(("cuboid", rot_cuboid(pos(), rng.uniform(-1.5, 1.5, 3), rng.uniform(0.02, 0.15, 3))))
    for _ in range(int(rng.integers(0, 12))):
        p1 = pos()
        spec.append(("capsule", capsule(p1, p1 + rng.uniform(-0.4, 0.4, 3).astype(np.float32), rng.uniform(0.01, 0.08))))
    for _ in range(int(rng.integers(0, 12))):
        p1 = pos()
        p2 = p1.copy()
        p2[2] += np.float32(rng.uniform(0.05, 0.6))
        spec.append(("capsule", capsule(p1, p2, rng.uniform(0.01, 0.08))))
    if rng.random() < 0.15:  # ill-formed primitives: the pruning layers must switch themselves off
        for k, (kind, p) in enumerate(spec):
            if kind == "cuboid" and k % 3 == 0:
                p[3:6] *= np.float32(rng.uniform(0.5, 2.0))
    if rng.random() < 0.45:
        n = int(rng.choice([2, 3, 17, 300, 1000, 4096, 10000]))
        k = 1.6 if robot == "baxter" else 1.0
        pts = shell_cloud(n, int(rng.integers(1 << 30)), 0.5 * k, 1.2 * k, 0.0, 1.5)
        if rng.random() < 0.3 and n >= 17:
            k = min(len(pts[::5]), len(pts[1::5]))
            pts[::5][:k] = pts[1::5][:k]  # duplicated points: equal coordinates on every axis
        r_min, r_max = RADII[robot]
        spec.append(("capt", (pts, r_min, r_max, POINT_RADIUS)))
    if rng.random() < 0.2:
        xd, yd = int(rng.integers(4, 40)), int(rng.integers(4, 40))
        img = rng.random((yd, xd)).astype(np.float32)
        spec.append(("heightfield", (np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-1.5, -0.2)], np.float32),
                                     np.array([2.6 / xd, 2.6 / yd, 1.0 / rng.uniform(0.3, 1.2)], np.float32), xd, yd,
                                     img.reshape(-1))))
    if rng.random() < 0.2:
        a = rng.uniform(-1, 1)
        tf = np.array([[np.cos(a), -np.sin(a), 0, 0.02], [np.sin(a), np.cos(a), 0, -0.01], [0, 0, 1, 0.05], [0, 0, 0, 1]], np.float32)
        sp = [[rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), rng.uniform(0, 0.3), rng.uniform(0.01, 0.05)]
              for _ in range(int(rng.integers(1, 14)))]
        spec.append(("attach", (tf, np.array(sp, np.float32))))
    if rng.random() < 0.12:  # point clouds and nothing else (one or two): the kernels' cloud-only variant
        spec = [e for e in spec if e[0] in ("capt", "attach")]
        k = 1.6 if robot == "baxter" else 1.0
        r_min, r_max = RADII[robot]
        while sum(e[0] == "capt" for e in spec) < (2 if rng.random() < 0.4 else 1):
            n = int(rng.choice([2, 17, 300, 1000, 4096, 10000]))
            spec.insert(0, ("capt", (shell_cloud(n, int(rng.integers(1 << 30)), 0.5 * k, 1.3 * k, 0.0, 1.5), r_min, r_max,
                                     POINT_RADIUS * float(rng.choice([1.0, 2.0])))))
    return spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=8.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    vamp.set_device(0)
    o = Oracle()
    t_end = time.time() + args.minutes * 60
    case = 0
    totals = dict(configs=0, edges=0, spheres=0)
    while time.time() < t_end:
        seed = args.seed * 1000003 + case
        rng = np.random.default_rng(seed)
        robot = ROBOTS[case % 4]
        spec = random_spec(rng, robot)
        try:
            env, oenv = build_product_env(spec), build_oracle_env(o, spec)
        except Exception as e:  # an environment either side rejects (capacity): both must reject
            print(f"case {case} seed {seed} {robot}: environment rejected ({type(e).__name__}: {e})", flush=True)
            case += 1
            continue
        # the self-collision kernel's words-per-wave grouping is sized from the batch; small batches would always get 1
        group = rng.choice(["", "1", "2", "3", "5", "8"])
        if group:
            os.environ["VMV_SELF_GROUP"] = str(group)
        else:
            os.environ.pop("VMV_SELF_GROUP", None)
        # the edge schedule (vmv_robot_tu.inc: launch_validate_motion): the default most of the time, the others too
        sched = rng.choice(["", "", "", "0", "1", "2", "3"])
        if sched:
            os.environ["VMV_EDGE_TASKS"] = str(sched)
        else:
            os.environ.pop("VMV_EDGE_TASKS", None)
        rid = o.robot(robot)
        lo, span = o.bounds(rid)
        mod = getattr(vamp, robot)
        n = int(rng.choice([1, 63, 200, 4097, 12000]))
        q = (lo + span * rng.random((n, len(lo)), dtype=np.float32)).astype(np.float32)
        q[:: 17] = (q[:: 17] * np.float32(1.7)).astype(np.float32)  # outside the joint bounds too
        try:
            got = mod.validate_batch(q, env)
        except vamp.VmvError as e:
            print(f"case {case} seed {seed} {robot}: product status {e.status} ({len(spec)} objects)", flush=True)
            case += 1
            continue
        want = o.validate_batch(rid, oenv, q, threads=8)
        ok = np.array_equal(got, want)
        m = int(rng.choice([1, 9, 300, 2500]))
        a = q[rng.integers(n, size=m)]
        b = (a + rng.normal(0, rng.choice([0.02, 0.2, 0.8]), a.shape)).astype(np.float32)
        b[::5] = a[::5]
        got_e = mod.validate_motion_batch(a, b, env)
        want_e = o.validate_motion_batch(rid, oenv, a, b, threads=8)
        ok_e = np.array_equal(got_e, want_e)
        ok_s = True
        if not any(k == "attach" for k, _ in spec):
            import ctypes
            s = np.concatenate([rng.uniform([-1.6, -1.6, -0.4], [1.6, 1.6, 1.8], (3000, 3)), rng.uniform(0.003, 0.5, (3000, 1))], 1).astype(np.float32)
            got_s = env.spheres_in_collision(s)
            f = ctypes.POINTER(ctypes.c_float)
            want_s = np.array([bool(o.L.vo_sphere_environment_in_collision(oenv.h, s[i, :3].ctypes.data_as(f), ctypes.c_float(float(s[i, 3]))))
                               for i in range(len(s))])
            ok_s = np.array_equal(got_s, want_s)
            totals["spheres"] += len(s)
        totals["configs"] += n
        totals["edges"] += m
        kinds = sorted({k for k, _ in spec})
        print(f"case {case} seed {seed} {robot} sched {sched or '-'}: {len(spec)} objects {kinds} | configs {n} valid {int(want.sum())} "
              f"{'ok' if ok else 'MISMATCH'} | edges {m} valid {int(want_e.sum())} {'ok' if ok_e else 'MISMATCH'} | "
              f"spheres {'ok' if ok_s else 'MISMATCH'}", flush=True)
        if not (ok and ok_e and ok_s):
            print(f"FAILED: reproduce with --seed {args.seed} (case {case}, generator seed {seed})", flush=True)
            sys.exit(1)
        case += 1
    print(f"soak ok: {case} cases, {totals}", flush=True)


if __name__ == "__main__":
    main()
