"""The two-joint clearance tables of the self-collision kernels (tools/gen_hip.py: self_tables; data in
vamp_mvt_amd/csrc/gen/<robot>_dev.inc) against the oracle's own fp32 FK: wherever a table says "group certainly free",
no fine pair of that group may collide — for configurations drawn everywhere in joint space, on cell borders, and with the
other joints anywhere.  (The device skips a group whose bit is 0; a wrong 0 would be a wrong answer.)"""
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
GEN = os.path.join(ROOT, "vamp_mvt_amd", "csrc", "gen")


def _tables(robot):
    path = os.path.join(GEN, f"{robot}_dev.inc")
    if not os.path.exists(path):
        pytest.skip("generated sources not built")
    text = open(path).read()
    out = []
    for m in re.finditer(r"// table (\d+): joints (\d+), (\d+); (.*)\n\s*__device__ const unsigned char kSelfTable\d+\[(\d+) \* \d+\] = \{\n(.*?)\n    \};",
                         text, re.S):
        ti, i, j, names, n = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4), int(m.group(5))
        data = np.array([int(v) for v in m.group(6).replace("\n", "").split(",") if v.strip()], np.uint8).reshape(n, n)
        fn = text[text.index(f"unsigned self_table{ti}(const float"):]
        lo_i, inv_i = re.search(r"fi = \(q\[\d+\] - (\S+)f\) \* (\S+)f;", fn).groups()
        lo_j, inv_j = re.search(r"fj = \(q\[\d+\] - (\S+)f\) \* (\S+)f;", fn).groups()
        groups = re.findall(r"bit (\d+): (\S+) vs\. (\S+) \(", names)
        out.append(dict(joints=(i, j), n=n, table=data, lo=(np.float32(float.fromhex(lo_i)), np.float32(float.fromhex(lo_j))),
                        inv=(np.float32(float.fromhex(inv_i)), np.float32(float.fromhex(inv_j))),
                        groups=[(int(b), a, bb) for b, a, bb in groups]))
    return out


@pytest.mark.parametrize("robot", ["panda", "ur5", "fetch", "baxter"])
def test_cells_marked_free_hold_no_colliding_pair(oracle, robot):
    tables = _tables(robot)
    if robot in ("panda", "ur5"):
        assert tables, "the two-joint groups of this robot should have a table"
    model = json.load(open(os.path.join(ROOT, "vamp_mvt_amd", "robots", f"{robot}.json")))
    radii = np.array(model["radii"], np.float32)
    rid = oracle.robot(robot)
    lo, span = oracle.bounds(rid)
    rng = np.random.default_rng(2024)
    n = 40000 if robot in ("panda", "ur5") else 12000
    for t in tables:
        i, j = t["joints"]
        q = (lo + span * rng.random((n, len(lo)), dtype=np.float32)).astype(np.float32)
        # a third of the samples on (and a hair off) cell borders of both joints
        k = n // 3
        for axis, (lo_a, inv_a) in zip((i, j), zip(t["lo"], t["inv"])):
            edge = rng.integers(0, t["n"] + 1, size=k).astype(np.float64) / float(inv_a) + float(lo_a)
            q[:k, axis] = (edge + rng.choice([-1e-6, 0.0, 1e-6], size=k)).astype(np.float32)
        q[k:k + 50] *= np.float32(1.5)  # outside the joint bounds: every bit must read 1
        fi = (q[:, i] - t["lo"][0]) * t["inv"][0]  # the device's fp32 arithmetic
        fj = (q[:, j] - t["lo"][1]) * t["inv"][1]
        inside = (fi >= 0) & (fj >= 0) & (fi < t["n"]) & (fj < t["n"])
        cell = np.where(inside, t["table"][np.clip(fi.astype(np.int64), 0, t["n"] - 1), np.clip(fj.astype(np.int64), 0, t["n"] - 1)], 0xff)
        S = np.stack([oracle.fk_all(rid, c) for c in q])  # fp32 sphere centres, the oracle's FK
        for bit, a, b in t["groups"]:
            g = next(g for g in model["self_groups"] if g["a"] == a and g["b"] == b)
            pr = np.array(g["pairs"])
            d = S[:, pr[:, 0], :3] - S[:, pr[:, 1], :3]
            sq = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]  # fp32, sql2_3's order
            rs = radii[pr[:, 0]] + radii[pr[:, 1]]
            collides = (sq - rs * rs < 0).any(axis=1)
            free = ((cell >> bit) & 1) == 0
            assert not (collides & free).any(), (robot, a, b, q[np.nonzero(collides & free)[0][:3]])
            if free.any():  # how close the skipped configurations come: must stay clear of the fp32 noise floor (~1e-6 m)
                clearance = (np.sqrt(sq[free].astype(np.float64)) - rs).min()
                assert clearance > 5e-5, (robot, a, b, clearance)
        assert (cell[~inside] == 0xff).all()
