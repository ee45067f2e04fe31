#!/usr/bin/env python3
"""Offline study (oracle FK = test infrastructure; geometry in numpy): candidate evaluations of the fine phase per wave of 64
Panda configurations in the 64-primitive shell scene — max-over-lanes loop steps as built vs (item, candidate) pairs dealt
evenly.   python tools/experiments/fine_eval_study.py [waves]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle  # noqa: E402
from vamp_mvt_amd.workloads import shell_spec  # noqa: E402

waves = int(sys.argv[1]) if len(sys.argv) > 1 else 30
CHUNK = 5
m = json.load(open(os.path.join(ROOT, "vamp_mvt_amd", "robots", "panda.json")))
o = Oracle()
rid = o.robot("panda")
lob, span = o.bounds(rid)
spec = shell_spec(0)
sph = np.array([p for k, p in spec if k == "sphere"], np.float64)
cub = np.array([p for k, p in spec if k == "cuboid"], np.float64)  # c(3) ax1(3) ax2(3) ax3(3) half(3)


def dist_spheres(c):  # distance from point c to every sphere primitive's surface
    return np.linalg.norm(sph[:, :3] - c, axis=1) - sph[:, 3]


def dist_cuboids(c):
    d = c - cub[:, :3]
    loc = np.stack([(d * cub[:, 3 + 3 * i:6 + 3 * i]).sum(1) for i in range(3)], 1)
    e = np.maximum(np.abs(loc) - cub[:, 12:15], 0.0)
    return np.linalg.norm(e, axis=1)


rng = np.random.default_rng(0)
tot = dict(rounds=0, steps_now_s=0, steps_now_z=0, pairs_s=0, pairs_z=0, items=0)
for w in range(waves):
    q = (lob + span * rng.random((64, len(lob)), dtype=np.float32)).astype(np.float32)
    S = np.stack([o.fk_all(rid, c) for c in q]).astype(np.float64)
    bad = np.zeros(64, bool)
    for g in m["env_groups"]:
        b = g["bound"]
        cs, cz, gate = [], [], []
        for i in range(64):
            ds, dz = dist_spheres(S[i, b, :3]) - S[i, b, 3], dist_cuboids(S[i, b, :3]) - S[i, b, 3]
            gate.append((not bad[i]) and (ds.min() < 0 or dz.min() < 0))
            cs.append(int((ds < 1e-4).sum()))
            cz.append(int((dz < 1e-4).sum()))
        lanes = [i for i in range(64) if gate[i]]
        k = len(lanes)
        if k == 0:
            continue
        fine = g["fine"]
        for c0 in range(0, len(fine), CHUNK):
            n = len(fine[c0:c0 + CHUNK])
            items = [lanes[t % k] for t in range(k * n)]  # sphere-major dealing: item t -> owner lanes[t % k]
            for r0 in range(0, len(items), 64):
                rd = items[r0:r0 + 64]
                tot["rounds"] += 1
                tot["steps_now_s"] += max(cs[i] for i in rd)
                tot["steps_now_z"] += max(cz[i] for i in rd)
                tot["pairs_s"] += sum(cs[i] for i in rd)
                tot["pairs_z"] += sum(cz[i] for i in rd)
                tot["items"] += len(rd)
        for i in lanes:
            for s in fine:
                if (dist_spheres(S[i, s, :3]) - S[i, s, 3]).min() < 0 or (dist_cuboids(S[i, s, :3]) - S[i, s, 3]).min() < 0:
                    bad[i] = True
                    break
print({k: round(v / waves, 2) for k, v in tot.items()}, "per wave")
print("even dealing would need", round(tot["pairs_s"] / 64 / waves, 2), "sphere steps and", round(tot["pairs_z"] / 64 / waves, 2), "cuboid steps per wave")
