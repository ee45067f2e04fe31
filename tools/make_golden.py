#!/usr/bin/env python3
"""Generates tests/golden/* from the reference, in this container (needs /root/reference + oracle/_ref).

What is written is DATA ONLY — inputs and the reference's outputs on them:

  arith.npz        x, sin(x), cos(x) from the reference's vector/avx.hh + interface.hh (compiled in place as
                   oracle/_ref/libref_vector.so); l2_norm probes for dims 6/7/8/14
  fk_<robot>.npz   random configurations and the sphere centres/radii the reference's generated
                   `Robot::fkcc` FK block yields for them (statements evaluated by tools/ref_fk_eval.py)
  halton_panda.npz first 20,001 samples of the reference's default Halton sequence for the Panda
                   (random/halton.hh compiled in place)
  known_answers.json  whole-pipeline known answers recorded from the real reference in SURVEY.md §8c
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_fk_eval as R  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    rv = R.RefVector()
    rng = np.random.default_rng(20251004)
    x = np.concatenate([
        rng.uniform(-7, 7, 6000), rng.uniform(-0.01, 0.01, 500), rng.uniform(-100, 100, 500),
        [0.0, -0.0, 1e-8, 3.14159265, 1.5707964, -1.5707964, 6.2831855, 0.78539816, 2.3561945, 4.712389]
    ]).astype(np.float32)
    l2 = {}
    for d in (6, 7, 8, 14):
        v = rng.uniform(-3, 3, (64, d)).astype(np.float32)
        l2[f"l2_in_{d}"] = v
        l2[f"l2_out_{d}"] = np.array([rv.l2_norm(r) for r in v], np.float32)
    np.savez_compressed(os.path.join(OUT, "arith.npz"), x=x, sin=rv.sin(x), cos=rv.cos(x), **l2)

    for name in ("panda", "ur5", "fetch", "baxter"):
        prog = R.load_program(name)
        c = R.robot_constants(name)
        lo, span = np.array(c["s_a"], np.float32), np.array(c["s_m"], np.float32)
        q = (lo + span * rng.random((96, len(lo)), dtype=np.float32)).astype(np.float32)
        q[0] = 0.0  # exact zeros / quarter turns: FK terms cancel and the smallest constants become visible
        for i in range(1, 17):
            q[i] = rng.choice(np.array([0, np.pi / 2, -np.pi / 2, np.pi, np.pi / 4], np.float32), len(lo))
        y = R.Evaluator(prog, rv).run(q)  # [n_y, N]
        spheres = np.ascontiguousarray(y.reshape(-1, 4, q.shape[0]).transpose(2, 0, 1))  # [N][S][4]
        # end-effector frame of Robot::fkcc_attach (its last 12 outputs: translation, rotation column-major)
        ya = R.Evaluator(R.load_program(name, "fkcc_attach"), rv).run(q)
        ee = np.ascontiguousarray(ya[-12:].T)  # [N][12]
        np.savez_compressed(os.path.join(OUT, f"fk_{name}.npz"), q=q, spheres=spheres,
                            n_fine=np.int32(c["n_spheres"]), ee=ee)
        print(name, "fk golden", spheres.shape)

    c = R.robot_constants("panda")
    h = rv.halton(np.array(c["s_m"], np.float32), np.array(c["s_a"], np.float32), 20001)
    np.savez_compressed(os.path.join(OUT, "halton_panda.npz"), samples=h)

    known = {
        "source": "SURVEY.md §8c (measured with the real reference by the survey; reproduced here by the oracle)",
        "panda_probe_q": [0, -0.785, 0, -2.356, 0, 1.571, 0.785],
        "panda_probe_sphere58": [0.3069904, 0.0730000, 0.4878696, 0.012],
        "cage_start_valid": True, "cage_goal_valid": True, "cage_edge_valid": False,
        "cage_halton_20000_valid": 3533, "cage_halton_20000_edges_valid": 142,
        "cage_mt19937_seed0_100000_valid": 17708,
        "halton_first_sample_prefix": [-0.989033, -1.099560, -2.119357],
    }
    with open(os.path.join(OUT, "known_answers.json"), "w") as f:
        json.dump(known, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
