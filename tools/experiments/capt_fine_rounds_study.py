#!/usr/bin/env python3
"""Offline study (oracle = test infrastructure): CAPT queries of the fine phase per wave of 64 configurations — rounds of the
per-link / per-chunk re-dealing vs one queue across links.   python tools/experiments/capt_fine_rounds_study.py [robot] [waves]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle  # noqa: E402
from vamp_mvt_amd.workloads import POINT_RADIUS, RADII, shell_cloud  # noqa: E402

robot = sys.argv[1] if len(sys.argv) > 1 else "fetch"
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 10
CHUNK = 8
m = json.load(open(os.path.join(ROOT, "vamp_mvt_amd", "robots", f"{robot}.json")))
o = Oracle()
env = o.env()
pts = shell_cloud(10000, 3) if robot != "baxter" else shell_cloud(10000, 4, 1.0, 1.8)
env.add_capt(pts, *RADII[robot], POINT_RADIUS)
rid = o.robot(robot)
lob, span = o.bounds(rid)
rng = np.random.default_rng(0)
tot = dict(gates=0, rounds_now=0, items=0, rounds_queue=0, links_with_hits=0, calls_now=0)
for w in range(waves):
    q = (lob + span * rng.random((64, len(lob)), dtype=np.float32)).astype(np.float32)
    S = np.stack([o.fk_all(rid, c) for c in q])  # [64][n_total][4]
    bad = np.zeros(64, bool)
    items_wave = 0
    for g in m["env_groups"]:
        act = ~bad
        hit = np.array([act[i] and env.capt_collides(S[i, g["bound"], :3], S[i, g["bound"], 3]) for i in range(64)])
        tot["gates"] += 1
        k = int(hit.sum())
        if k == 0:
            continue
        tot["links_with_hits"] += 1
        fine = g["fine"]
        link_bad = np.zeros(64, bool)
        for c0 in range(0, len(fine), CHUNK):
            n = len(fine[c0:c0 + CHUNK])
            tot["rounds_now"] += -(-k * n // 64)
            tot["calls_now"] += 1
            items_wave += k * n
            for i in np.nonzero(hit)[0]:
                for s in fine[c0:c0 + CHUNK]:
                    if env.capt_collides(S[i, s, :3], S[i, s, 3]):
                        link_bad[i] = True
        bad |= link_bad
    tot["items"] += items_wave
    tot["rounds_queue"] += -(-items_wave // 64)
print(robot, {k: v / waves for k, v in tot.items()}, "per wave")
