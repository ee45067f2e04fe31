#!/usr/bin/env python3
"""Offline study (oracle = test infrastructure): fine-phase rounds of the environment kernel per wave of 64 configurations —
per link and slab chunk (as built) vs one queue across links.   python tools/experiments/fine_rounds_study.py [robot] [waves] [chunk]"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from envs import build_oracle_env  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from vamp_mvt_amd.workloads import shell_spec  # noqa: E402

robot = sys.argv[1] if len(sys.argv) > 1 else "panda"
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 20
CHUNK = int(sys.argv[3]) if len(sys.argv) > 3 else 5
m = json.load(open(os.path.join(ROOT, "vamp_mvt_amd", "robots", f"{robot}.json")))
o = Oracle()
env = build_oracle_env(o, shell_spec(0))
rid = o.robot(robot)
lob, span = o.bounds(rid)
rng = np.random.default_rng(0)
f = ctypes.POINTER(ctypes.c_float)


def hit(s):
    c = np.ascontiguousarray(s[:3], np.float32)
    return bool(o.L.vo_sphere_environment_in_collision(env.h, c.ctypes.data_as(f), ctypes.c_float(float(s[3]))))


tot = dict(links_with_gate=0, calls=0, rounds_now=0, items=0, rounds_queue=0, fine_fk_links=0)
for w in range(waves):
    q = (lob + span * rng.random((64, len(lob)), dtype=np.float32)).astype(np.float32)
    S = np.stack([o.fk_all(rid, c) for c in q])
    bad = np.zeros(64, bool)
    items_wave = 0
    for g in m["env_groups"]:
        gate = np.array([(not bad[i]) and hit(S[i, g["bound"]]) for i in range(64)])
        k = int(gate.sum())
        if k == 0:
            continue
        tot["links_with_gate"] += 1
        fine = g["fine"]
        for c0 in range(0, len(fine), CHUNK):
            n = len(fine[c0:c0 + CHUNK])
            tot["calls"] += 1
            tot["rounds_now"] += -(-k * n // 64)
        items_wave += k * len(fine)
        for i in np.nonzero(gate)[0]:
            if any(hit(S[i, s]) for s in fine):
                bad[i] = True
    tot["items"] += items_wave
    tot["rounds_queue"] += -(-items_wave // 64)
print(robot, "chunk", CHUNK, {k: round(v / waves, 2) for k, v in tot.items()}, "per wave; links", len(m["env_groups"]))
