#!/usr/bin/env python3
"""In-container check: tools/robot_trace.py output vs the reference's generated FK (read as data).

Validation tooling only (needs /root/reference and oracle/_ref).  For each robot
it evaluates (a) the build's robot model JSON and (b) the reference's fkcc
statements (tools/ref_fk_eval.py) on the same random configurations and reports
bit-level agreement of every sphere coordinate and radius, plus equality of the
environment / self-collision group structure.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_fk_eval as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROBOT_DIR = os.path.join(HERE, "..", "vamp_mvt_amd", "robots")


def eval_model(model, q, rv):
    """numpy fp32 evaluation of a robot model's op tape. q: [N, dim] -> [n_total_spheres, 4, N]."""
    q = np.ascontiguousarray(q, np.float32)
    n = q.shape[0]
    vals = []
    for op, a, b in model["ops"]:
        if op == "in":
            v = np.ascontiguousarray(q[:, a])
        elif op == "sin":
            v = rv.sin(vals[a])
        elif op == "cos":
            v = rv.cos(vals[a])
        elif op == "neg":
            v = -vals[a]
        elif op == "const":
            v = np.full(n, np.float32(a), np.float32)
        elif op == "mul":
            v = vals[a] * vals[b]
        elif op == "add":
            v = vals[a] + vals[b]
        elif op == "sub":
            v = vals[a] - vals[b]
        elif op == "cmul":
            v = np.float32(a) * vals[b]
        elif op == "cadd":
            v = np.float32(a) + vals[b]
        else:
            raise ValueError(op)
        vals.append(v.astype(np.float32, copy=False))
    ns = len(model["outputs"])
    out = np.zeros((ns, 4, n), np.float32)
    for s, o in enumerate(model["outputs"]):
        for c, (kind, val) in enumerate(o):
            out[s, c] = vals[val] if kind == "op" else np.float32(val)
        out[s, 3] = np.float32(model["radii"][s])
    return out


def check(name, n=20000, seed=1, verbose=True):
    with open(os.path.join(ROBOT_DIR, name + ".json")) as f:
        model = json.load(f)
    rv = R.RefVector()
    prog = R.load_program(name)
    consts = R.robot_constants(name)
    ok = True
    for key in ("dimension", "n_spheres", "resolution"):
        if model[key] != consts[key]:
            print(f"  {key}: model {model[key]} != reference {consts[key]}")
            ok = False
    if model["joint_names"] != consts["joint_names"]:
        print("  joint names differ")
        ok = False
    for key in ("min_radius", "max_radius"):
        if np.float32(model[key]) != np.float32(consts[key]):
            print(f"  {key}: model {model[key]} != reference {consts[key]}")
            ok = False
    rng = np.random.default_rng(seed)
    lo = np.array(consts["s_a"], np.float32)
    span = np.array(consts["s_m"], np.float32)
    if not np.array_equal(np.array(model["lower"], np.float32), lo):
        print("  lower bounds differ", model["lower"], lo)
        ok = False
    if not np.array_equal(np.array(model["span"], np.float32), span):
        print("  spans differ", model["span"], span)
        ok = False
    if not np.array_equal(np.array(model["descale"], np.float32), np.array(consts["d_m"], np.float32)):
        print("  descale differs", model["descale"], consts["d_m"])
        ok = False
    q = (rng.random((n, len(lo)), dtype=np.float32) * span + lo).astype(np.float32)
    # special configurations: exact zeros / quarter turns, where FK terms cancel and 1e-18 constants become visible
    special = [np.zeros(len(lo))]
    for j in range(len(lo)):
        for v in (np.pi / 2, -np.pi / 2, np.pi, 1e-3):
            s_ = np.zeros(len(lo))
            s_[j] = v
            special.append(s_)
    for _ in range(400):
        special.append(rng.choice([0, np.pi / 2, -np.pi / 2, np.pi, -np.pi, np.pi / 4], len(lo)))
    q = np.concatenate([np.array(special, np.float32), q]).astype(np.float32)
    n = q.shape[0]
    yref = R.Evaluator(prog, rv).run(q)  # [n_y, N]
    ymod = eval_model(model, q, rv)  # [S, 4, N]
    S = ymod.shape[0]
    if prog.n_y != 4 * S:
        print(f"  sphere count: model {S} vs reference {prog.n_y // 4}")
        ok = False
        S = min(S, prog.n_y // 4)
    yref = yref[:4 * S].reshape(S, 4, n)
    bad_spheres = []
    for s in range(S):
        for c in range(4):
            ne = int((yref[s, c].view(np.uint32) != ymod[s, c].view(np.uint32)).sum())
            if ne:
                bad_spheres.append((s, c, ne, float(np.abs(yref[s, c] - ymod[s, c]).max())))
    if bad_spheres:
        ok = False
        print(f"  {len(bad_spheres)} sphere coordinates differ; first few:")
        for s, c, ne, md in bad_spheres[:12]:
            print(f"    sphere {s} ({model['links'][model['sphere_link'][s]]}) coord {c}: {ne}/{n} lanes differ, max |d| = {md:.3e}")
    # structure
    ref_env = [(g[0], g[1] // 4, [i // 4 for i in g[2]]) for g in prog.env_groups]
    mod_env = [(g["link"], g["bound"], g["fine"]) for g in model["env_groups"]]
    if ref_env != mod_env:
        ok = False
        print("  env groups differ")
        for a, b in zip(ref_env, mod_env):
            if a != b:
                print("    ref", a, "\n    mod", b)
                break
    ref_self = [(g[0], g[1], g[2][0] // 4, g[2][1] // 4, [[a // 4, b // 4] for a, b in g[3]]) for g in prog.self_groups]
    mod_self = [(g["a"], g["b"], g["bound_a"], g["bound_b"], g["pairs"]) for g in model["self_groups"]]
    if ref_self != mod_self:
        ok = False
        print(f"  self groups differ: ref {len(ref_self)} groups, model {len(mod_self)}")
        for a, b in zip(ref_self, mod_self):
            if a != b:
                print("    ref", a[:4], len(a[4]), a[4][:4], "\n    mod", b[:4], len(b[4]), b[4][:4])
                break
    # end-effector frame and attachment structure of Robot::fkcc_attach
    progA = R.load_program(name, "fkcc_attach")
    yA = R.Evaluator(progA, rv).run(q)
    base = progA.n_y - 12
    ee = eval_model(dict(ops=model["ee_ops"], outputs=[model["ee_outputs"][3 * i:3 * i + 3] for i in range(4)],
                         radii=[0, 0, 0, 0]), q, rv)
    mine = np.concatenate([ee[i, :3] for i in range(4)])
    if not np.array_equal(mine.view(np.uint32), yA[base:base + 12].view(np.uint32)):
        ok = False
        print("  end-effector frame differs from fkcc_attach y[%d..%d]" % (base, base + 11))
    if not np.array_equal(yA[:base].view(np.uint32), R.Evaluator(prog, rv).run(q).view(np.uint32)):
        ok = False
        print("  fkcc_attach computes different sphere centres than fkcc")
    import re
    txt = open(R.ROBOT_HH.format(name=name)).read()
    seg = txt[txt.index("inline static bool fkcc_attach("):]
    seg = seg[seg.index("// attaching at"):]
    seg = seg[:seg.index("return true;")]
    if not seg.startswith("// attaching at " + model["end_effector"]):
        ok = False
        print("  attachment frame differs:", seg[:60])
    if re.findall(r"// Attachment vs\. (\S+)", seg) != model["attach_links"]:
        ok = False
        print("  attachment link list differs")
    max_err = float(np.abs(yref[:, :3] - ymod[:S, :3]).max())
    print(f"{name}: {'BIT-EXACT' if ok else 'MISMATCH'} on {n} configs; {S} spheres + end-effector frame; "
          f"max |centre diff| = {max_err:.3e}")
    return ok


if __name__ == "__main__":
    names = sys.argv[1:] or ["panda"]
    rc = 0
    for nm in names:
        if not check(nm):
            rc = 1
    sys.exit(rc)
