"""MotionBenchMaker problems of the reference (resources/<robot>/problems.tar.bz2 -> tests/golden/mbm_<robot>.npz by
tools/make_mbm_golden.py): the one reference-held fixture family with capsules and rotated cuboids.

Known answers (resources/README.md:146,81,210, the seven scenario families of README.md:24, validity rule
resources/problem_tar_to_pkl_json.py:79-84 = start valid and some goal valid, scene mapping src/vamp/__init__.py:140-186):
    Panda 699 / 700 - REPRODUCED exactly by the oracle and by the HIP path: the only one of the three that is a pin.
    UR5   608 / 700 published, 689 / 700 here;  Fetch 679 / 700 published, 671 / 700 here:
    **parity unpinned: reference-held answer not reproduced, cause unproven.**
For UR5 and Fetch, oracle == HIP agreement on these scenes shows self-consistency (on capsules and rotated cuboids of
real scenes), not reference parity.  What is recorded so that a reader with the real reference can check it
(tools/mbm_diagnostics.py -> tests/golden/mbm_diagnostics.json, re-derived by test_mbm_diagnostics_are_current):
  * per family: start-valid / goal-valid / both (PER_FAMILY below);
  * per invalid problem: endpoint, the predicate that fires (self-collision link pair or scene object) and the deepest
    penetration: every one of the 1 + 11 + 29 invalid problems is a graze of 0.03 - 8.9 mm, none is a deep overlap
    (UR5: ten goals with forearm x wrist_2/3 self-collisions of 0.7 - 8.9 mm + one start 2.4 mm into a cuboid; Fetch: 22
    goals 0.03 - 5.9 mm into a shelf board or pole, two head_pan x upperarm_roll goals at 1.0 mm, five starts with
    base_link self-collisions of 0.5 - 5.8 mm);
  * the near misses among the problems valid here: 81 UR5 problems have an environment gap below 5 mm (6 below 2 mm) -
    exactly the size of the UR5 gap (689 - 608 = 81) - and 9 of Fetch's 29 invalid problems are grazes of <= 1 mm
    (679 - 671 = 8).  Both gaps are thus of the size that millimetre-level differences in the scene data or the sphere
    models produce; which of the two it is cannot be told offline (the archive in this checkout holds 13 scenario
    families where the README's tables speak of 7 x 100 problems; FK and group tables here are bit-identical to the
    shipped headers; joint order equals the reference's `joint_names`; a joint-bounds check changes nothing for UR5).

This is a tolerance-level pin: Euler angles are recomputed from the scene quaternions and the euler -> axes
constructors are this package's fp32 restatement of collision/factory.hh (the reference uses Eigen)."""
import json
import os

import numpy as np
import pytest

STANDARD = ["bookshelf_small", "bookshelf_tall", "bookshelf_thin", "box", "cage", "table_pick", "table_under_pick"]
PUBLISHED = {"panda": 699, "ur5": 608, "fetch": 679}   # resources/README.md:146,81,210
HERE = {"panda": 699, "ur5": 689, "fetch": 671}        # oracle == HIP path; UR5 / Fetch: parity unpinned (module docstring)
# valid starts / valid goals / both, of 100 problems per family (oracle; the reference's rule counts "both")
PER_FAMILY = {
    "panda": {"bookshelf_small": (100, 100, 100), "bookshelf_tall": (100, 100, 100), "bookshelf_thin": (100, 100, 100),
              "box": (100, 100, 100), "cage": (100, 100, 100), "table_pick": (100, 99, 99), "table_under_pick": (100, 100, 100)},
    "ur5": {"bookshelf_small": (100, 96, 96), "bookshelf_tall": (100, 95, 95), "bookshelf_thin": (100, 99, 99),
            "box": (100, 100, 100), "cage": (100, 100, 100), "table_pick": (100, 100, 100), "table_under_pick": (99, 100, 99)},
    "fetch": {"bookshelf_small": (100, 98, 98), "bookshelf_tall": (100, 96, 96), "bookshelf_thin": (100, 84, 84),
              "box": (100, 99, 99), "cage": (100, 99, 99), "table_pick": (100, 100, 100), "table_under_pick": (95, 100, 95)},
}


def _load(golden_dir, robot):
    return np.load(os.path.join(golden_dir, f"mbm_{robot}.npz"))


def problem_primitives(vamp, g, i):
    """-> list of ("sphere" | "cuboid" | "capsule", canonical params) per src/vamp/__init__.py:140-186"""
    spec = []
    box_problem = str(g["names"][i]) == "box"
    for s in g["spheres"][g["sphere_off"][i]:g["sphere_off"][i + 1]]:
        sp = vamp.Sphere(s[:3], s[3])
        spec.append(("sphere", np.array([sp.x, sp.y, sp.z, sp.r], np.float32)))
    for c in g["cylinders"][g["cyl_off"][i]:g["cyl_off"][i + 1]]:
        if box_problem:  # the "box" scenario over-approximates its cylinders with boxes
            spec.append(("cuboid", vamp.Cuboid(c[:3], c[3:6], [c[6], c[6], c[7] / 2]).params))
        else:
            spec.append(("capsule", vamp.Cylinder(c[:3], c[3:6], c[6], c[7]).params))
    for b in g["boxes"][g["box_off"][i]:g["box_off"][i + 1]]:
        spec.append(("cuboid", vamp.Cuboid(b[:3], b[3:6], b[6:9]).params))
    return spec


@pytest.mark.parametrize("robot", ["panda", "ur5", "fetch"])
def test_oracle_mbm_validity_counts(vamp, oracle, golden_dir, robot):
    from envs import build_oracle_env

    g = _load(golden_dir, robot)
    rid = oracle.robot(robot)
    kinds, valid = set(), 0
    for i in np.flatnonzero(np.isin(g["names"], STANDARD)):
        spec = problem_primitives(vamp, g, i)
        kinds.update(k for k, _ in spec)
        env = build_oracle_env(oracle, spec)
        valid += oracle.validate(rid, env, g["start"][i].astype(np.float32)) and \
            oracle.validate(rid, env, g["goal"][i].astype(np.float32))
    assert kinds == {"sphere", "cuboid", "capsule"} or kinds == {"cuboid", "capsule"}
    assert valid == HERE[robot]
    if robot == "panda":
        assert valid == PUBLISHED[robot]


@pytest.mark.parametrize("robot", ["panda", "ur5", "fetch"])
def test_mbm_diagnostics_are_current(vamp, oracle, golden_dir, robot):
    """tests/golden/mbm_diagnostics.json (per-family table, the predicate and penetration depth of every invalid problem,
    near-miss counts) is what tools/mbm_diagnostics.py derives from the oracle today, and says what the module docstring
    says: every invalid problem is a graze below 1 cm, and 81 UR5 problems sit within 5 mm of the environment."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from mbm_diagnostics import diagnose

    committed = json.load(open(os.path.join(golden_dir, "mbm_diagnostics.json")))
    now = json.loads(json.dumps(diagnose(vamp, oracle, golden_dir, robot)))
    assert now == committed[robot]
    assert now["valid"] == HERE[robot] and committed["published"][robot] == PUBLISHED[robot]
    for fam, (s_ok, g_ok, both) in PER_FAMILY[robot].items():
        row = now["families"][fam]
        assert (row["start_valid"], row["goal_valid"], row["both"], row["problems"]) == (s_ok, g_ok, both, 100)
    assert len(now["invalid"]) >= 700 - HERE[robot]
    for e in now["invalid"]:
        assert e["cause"] in ("self", "environment", "self+environment")
        assert 0.0 < max(e["self_depth_m"], e["env_depth_m"]) < 0.01, e  # grazes, not deep overlaps
    if robot == "ur5":
        assert now["near_misses"]["environment_gap_below_5mm"] == HERE["ur5"] - PUBLISHED["ur5"] == 81


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["panda", "ur5", "fetch"])
def test_gpu_mbm_validity_matches_oracle_problem_by_problem(vamp, oracle, golden_dir, robot):
    """all 1,300 scenes of the archive (13 families): start and goal validity from the HIP path == the oracle's, and the
    count over the seven standard families is the pinned one"""
    from envs import build_oracle_env, build_product_env

    vamp.set_device(0)
    g = _load(golden_dir, robot)
    rid = oracle.robot(robot)
    mod = getattr(vamp, robot)
    valid = 0
    for i in range(len(g["names"])):
        spec = problem_primitives(vamp, g, i)
        q = np.stack([g["start"][i], g["goal"][i]]).astype(np.float32)
        got = mod.validate_batch(q, build_product_env(spec))
        want = oracle.validate_batch(rid, build_oracle_env(oracle, spec), q)
        assert np.array_equal(got, want), f"{robot} problem {g['names'][i]}/{g['index'][i]}"
        if str(g["names"][i]) in STANDARD:
            valid += bool(got.all())
    assert valid == HERE[robot]
