#!/usr/bin/env python3
"""Offline study: which share of (q_i, q_j) space is CERTAINLY free for the self-collision groups whose relative pose depends
on two joints only (Panda: link5 vs link7 / hand / fingers depend on joints 6, 7)?   oracle = test infrastructure."""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle  # noqa: E402

robot = sys.argv[1] if len(sys.argv) > 1 else "panda"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
m = json.load(open(os.path.join(ROOT, "vamp_mvt_amd", "robots", f"{robot}.json")))
o = Oracle()
rid = o.robot(robot)
lo, span = o.bounds(rid)
radii = np.array(m["radii"])
groups = [g for g in m["self_groups"] if g["a"] == "panda_link5"]
pairs = np.array([p for g in groups for p in g["pairs"]])
print(len(pairs), "pairs in", [g["b"] for g in groups])
ja, jb = 5, 6  # joints 6, 7 (0-based 5, 6)
qa = lo[ja] + span[ja] * (np.arange(N) + 0.5) / N
qb = lo[jb] + span[jb] * (np.arange(N) + 0.5) / N
clear = np.zeros((N, N))
reach = 0.0
for i, a in enumerate(qa):
    for j, b in enumerate(qb):
        q = np.zeros(len(lo), np.float32)
        q[ja], q[jb] = a, b
        S = o.fk_all(rid, q).astype(np.float64)
        d = np.linalg.norm(S[pairs[:, 0], :3] - S[pairs[:, 1], :3], axis=1) - radii[pairs[:, 0]] - radii[pairs[:, 1]]
        clear[i, j] = d.min()
        if i == 0 and j == 0:
            # lever arms: distance of the B spheres from the A spheres bounds the speed of |pB - pA| per radian
            reach = np.linalg.norm(S[pairs[:, 0], :3] - S[pairs[:, 1], :3], axis=1).max()
L = 0.45  # generous lever arm [m/rad] (the whole wrist + hand is < 0.3 m from joint 6)
slack = L * 0.5 * (span[ja] / N + span[jb] / N) + 1e-4
print("grid", N, "max |pB - pA| at one pose", reach, "slack", slack)
print("min clearance over the grid", clear.min(), " share certainly free:", (clear > slack).mean(), " share colliding:", (clear < 0).mean())
for thr in (0.0, 0.002, 0.005, 0.01, 0.02):
    print("  clearance >", thr, ":", (clear > thr).mean())
