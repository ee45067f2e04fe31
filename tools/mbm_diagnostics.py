#!/usr/bin/env python3
"""Why is a MotionBenchMaker problem invalid here?  (VERDICT r2 item 5 / ADVICE r2.)

The reference publishes valid-problem counts for the seven standard scenario families (resources/README.md:146,81,210):
Panda 699 / 700, UR5 608 / 700, Fetch 679 / 700.  This build reproduces Panda's and gets UR5 689, Fetch 671.  This tool
records, with the CPU oracle (test infrastructure), what could tell a reader with the real reference where the gap is:

  * per robot and family: problems whose start is valid / whose goal is valid / both (the reference's rule,
    resources/problem_tar_to_pkl_json.py:79-84);
  * per invalid problem: which endpoint fails, WHICH predicate fires — the self-collision pairs (link names) or the
    environment objects (kind and index in the scene) — and the deepest penetration in metres, so that a 1-mm graze
    (data / rounding) can be told from a deep overlap (model).

Penetration depths are computed in float64 from the oracle's FK spheres and the canonical primitive parameters; validity
itself is the oracle's answer (the reference's fp32 predicates).  Output: tests/golden/mbm_diagnostics.json, which
tests/test_mbm.py re-derives and compares.

    python tools/mbm_diagnostics.py [--print]"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

STANDARD = ["bookshelf_small", "bookshelf_tall", "bookshelf_thin", "box", "cage", "table_pick", "table_under_pick"]


def env_depths(spheres, spec, clearance=None):
    """-> list of (depth_m, sphere index, kind, object index within its kind) for every overlapping (sphere, primitive);
    clearance (a one-element list) receives the smallest sphere-to-primitive gap (negative = overlap)"""
    out = []
    c = spheres[:, :3].astype(np.float64)
    r = spheres[:, 3].astype(np.float64)
    count = {}
    for kind, p in spec:
        j = count.get(kind, 0)
        count[kind] = j + 1
        p = np.asarray(p, np.float64)
        if kind == "sphere":
            dist = np.linalg.norm(c - p[:3], axis=1) - p[3]
        elif kind == "capsule":  # x1 y1 z1 | xv yv zv | r | 1 / |v|^2
            v = p[3:6]
            t = np.clip(((c - p[:3]) @ v) * p[7], 0.0, 1.0)
            dist = np.linalg.norm(c - (p[:3] + t[:, None] * v), axis=1) - p[6]
        else:  # cuboid: centre | three axes | half extents
            d = c - p[:3]
            local = np.stack([d @ p[3:6], d @ p[6:9], d @ p[9:12]], 1)
            out_of = np.maximum(np.abs(local) - p[12:15], 0.0)
            inside = np.max(np.abs(local) - p[12:15], axis=1)
            dist = np.where((out_of > 0).any(1), np.linalg.norm(out_of, axis=1), inside)
        depth = r - dist
        if clearance is not None and len(depth):
            clearance[0] = min(clearance[0], float(-depth.max()))
        for s in np.flatnonzero(depth > 0):
            out.append((float(depth[s]), int(s), kind, j))
    return out


def self_depths(spheres, groups, clearance=None):
    out = []
    for a, b, _, _, pairs in groups:
        for i, j in pairs:
            d = float(spheres[i, 3]) + float(spheres[j, 3]) - float(np.linalg.norm(spheres[i, :3].astype(np.float64) -
                                                                                   spheres[j, :3].astype(np.float64)))
            if clearance is not None:
                clearance[0] = min(clearance[0], -d)
            if d > 0:
                out.append((d, a, b, i, j))
    return out


def diagnose(vamp, oracle, golden_dir, robot, families=STANDARD):
    from envs import build_oracle_env
    from test_mbm import problem_primitives

    g = np.load(os.path.join(golden_dir, f"mbm_{robot}.npz"))
    groups = json.load(open(os.path.join(golden_dir, f"groups_{robot}.json")))["self_groups"]
    rid = oracle.robot(robot)
    table, invalid = {}, []
    margins = []  # of the problems valid here: the smallest (environment gap, allowed-self-pair gap) over start and goal
    for i in np.flatnonzero(np.isin(g["names"], families)):
        fam = str(g["names"][i])
        row = table.setdefault(fam, {"problems": 0, "start_valid": 0, "goal_valid": 0, "both": 0})
        spec = problem_primitives(vamp, g, i)
        env = build_oracle_env(oracle, spec)
        ok = {}
        for which in ("start", "goal"):
            q = g[which][i].astype(np.float32)
            ok[which] = bool(oracle.validate(rid, env, q))
            sp = oracle.fk(rid, q)
            if ok[which]:
                gap_e, gap_s = [float("inf")], [float("inf")]
                env_depths(sp, spec, gap_e)
                self_depths(sp, groups, gap_s)
                ok[which + "_gap"] = (gap_e[0], gap_s[0])
                continue
            selfs = sorted(self_depths(sp, groups), reverse=True)
            envs = sorted(env_depths(sp, spec), reverse=True)
            entry = {"family": fam, "index": int(g["index"][i]), "endpoint": which,
                     "self_pairs": sorted({f"{a} x {b}" for _, a, b, _, _ in selfs}),
                     "self_depth_m": round(selfs[0][0], 5) if selfs else 0.0,
                     "env_objects": sorted({f"{kind}[{j}]" for _, _, kind, j in envs}),
                     "env_depth_m": round(envs[0][0], 5) if envs else 0.0}
            entry["cause"] = "+".join(k for k, v in (("self", selfs), ("environment", envs)) if v) or "graze below float64 resolution"
            invalid.append(entry)
        row["problems"] += 1
        row["start_valid"] += ok["start"]
        row["goal_valid"] += ok["goal"]
        row["both"] += ok["start"] and ok["goal"]
        if ok["start"] and ok["goal"]:
            margins.append(np.minimum(ok["start_gap"], ok["goal_gap"]))
    margins = np.asarray(margins).reshape(-1, 2)
    near = {f"{what}_gap_below_{mm}mm": int((margins[:, k] < mm * 1e-3).sum())
            for k, what in enumerate(("environment", "self")) for mm in (1, 2, 5, 10, 20)}
    return {"families": table, "near_misses": near, "valid": sum(r["both"] for r in table.values()),
            "total": sum(r["problems"] for r in table.values()), "invalid": invalid}


def main():
    import vamp_mvt_amd as vamp
    from oracle_lib import Oracle

    golden = os.path.join(ROOT, "tests", "golden")
    oracle = Oracle()
    out = {"published": {"panda": 699, "ur5": 608, "fetch": 679},
           "note": "oracle answers on the seven standard families of tests/golden/mbm_<robot>.npz; depths in metres (float64 "
                   "geometry on the oracle's fp32 FK spheres); see tools/mbm_diagnostics.py"}
    for robot in ("panda", "ur5", "fetch"):
        out[robot] = diagnose(vamp, oracle, golden, robot)
        d = out[robot]
        print(f"{robot}: {d['valid']} / {d['total']} valid (published {out['published'][robot]})")
        nm = list(d["near_misses"].values())
        print("   problems valid here whose smallest gap is below 1 / 2 / 5 / 10 / 20 mm: environment", nm[:5], "self pairs", nm[5:])
        for fam, r in d["families"].items():
            print(f"   {fam:18s} start {r['start_valid']:3d}  goal {r['goal_valid']:3d}  both {r['both']:3d}")
        if "--print" in sys.argv:
            for e in d["invalid"]:
                print(f"   {e['family']:18s} #{e['index']:<3d} {e['endpoint']:5s} {e['cause']:16s} self {e['self_depth_m']:.4f} m "
                      f"{e['self_pairs']}  env {e['env_depth_m']:.4f} m {e['env_objects']}")
    with open(os.path.join(golden, "mbm_diagnostics.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
