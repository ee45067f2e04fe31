#!/usr/bin/env python3
"""Offline study: how many CAPT gate queries (one per link per wave of 64 configurations) could a coarse "distance to the
nearest cloud point" grid reject for the WHOLE wave before the descent?   python tools/experiments/capt_skip_study.py [robot] [waves] [cell]"""
import json
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle  # noqa: E402
from vamp_mvt_amd.workloads import POINT_RADIUS, shell_cloud  # noqa: E402

robot = sys.argv[1] if len(sys.argv) > 1 else "baxter"
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cell = float(sys.argv[3]) if len(sys.argv) > 3 else 0.04
m = json.load(open(os.path.join(ROOT, "vamp_mvt_amd", "robots", f"{robot}.json")))
o = Oracle()
rid = o.robot(robot)
lob, span = o.bounds(rid)
pts = shell_cloud(10000, 3) if robot != "baxter" else shell_cloud(10000, 4, 1.0, 1.8)
tree = cKDTree(pts)
top_lo, top_hi = pts.min(0), pts.max(0)
rng = np.random.default_rng(0)
half_diag = cell * np.sqrt(3) / 2
calls = skip_wave = lanes_in = lanes_rej = 0
for w in range(waves):
    if w % 2 == 0:  # uniform configurations
        q = (lob + span * rng.random((64, len(lob)), dtype=np.float32)).astype(np.float32)
    else:           # 8 edges x 8 consecutive interpolated configurations (what a wave of the edge kernels holds)
        a = (lob + span * rng.random((8, len(lob)), dtype=np.float32)).astype(np.float32)
        d = rng.normal(0, 1, a.shape); d /= np.linalg.norm(d, axis=1, keepdims=True)
        q = np.concatenate([a[i] + d[i] * np.linspace(0, 0.25, 8)[:, None] for i in range(8)]).astype(np.float32)
    S = np.stack([o.fk_all(rid, c) for c in q]).astype(np.float64)
    for g in m["env_groups"]:
        c, r = S[:, g["bound"], :3], S[:, g["bound"], 3]
        inb = np.all((c + r[:, None] >= top_lo) & (c - r[:, None] <= top_hi), axis=1)  # the query's own top-box test
        if not inb.any():
            continue
        calls += 1
        dist, _ = tree.query(c)
        # grid lower bound: distance at the cell centre - half diagonal <= dist - (0 .. 2 half diagonals): take the worst
        near = dist - 2 * half_diag <= r + POINT_RADIUS + 1e-4
        lanes_in += int(inb.sum())
        lanes_rej += int((inb & ~near).sum())
        if not (inb & near).any():
            skip_wave += 1
print(f"{robot} cell {cell}: {calls / waves:.1f} gate calls per wave reach the descent today; a distance grid would stop {skip_wave / calls:.1%} of them "
      f"for the whole wave ({lanes_rej / max(lanes_in, 1):.1%} of the lanes)")
