"""The robot models (vamp_mvt_amd/robots/*.json, from which BOTH the oracle's and the HIP kernels' tables are generated)
against the collision structure of the reference's generated checkers (tests/golden/groups_<robot>.json, extracted by
tools/make_groups_golden.py).  CPU only; needs neither a GPU nor /root/reference."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
COUNTS = {"panda": (11, 21, 690), "ur5": (17, 55, 383), "fetch": (15, 48, 2586), "baxter": (33, 349, 1845)}


@pytest.mark.parametrize("name", ["panda", "ur5", "fetch", "baxter"])
def test_model_structure_equals_the_references(golden_dir, name):
    ref = json.load(open(os.path.join(golden_dir, f"groups_{name}.json")))
    model = json.load(open(os.path.join(ROOT, "vamp_mvt_amd", "robots", f"{name}.json")))
    for key in ("dimension", "n_spheres", "resolution", "joint_names"):
        assert model[key] == ref[key], key
    for key in ("min_radius", "max_radius"):
        assert np.float32(model[key]) == np.float32(ref[key]), key
    assert len(model["outputs"]) == ref["n_total_spheres"]
    assert [[g["link"], g["bound"], g["fine"]] for g in model["env_groups"]] == ref["env_groups"]
    assert [[g["a"], g["b"], g["bound_a"], g["bound_b"], g["pairs"]] for g in model["self_groups"]] == ref["self_groups"]
    assert model["end_effector"] == ref["attach_frame"] and model["attach_links"] == ref["attach_links"]
    n_env, n_self, n_pairs = COUNTS[name]
    assert (len(ref["env_groups"]), len(ref["self_groups"]), sum(len(g[4]) for g in ref["self_groups"])) == (n_env, n_self, n_pairs)


@pytest.mark.parametrize("name", ["panda", "ur5", "fetch", "baxter"])
def test_oracle_and_library_tables_equal_the_references(oracle, vamp, golden_dir, name):
    """what was actually compiled in: the oracle's and the C-ABI library's exported fine-pair tables"""
    import ctypes
    ref = json.load(open(os.path.join(golden_dir, f"groups_{name}.json")))
    pairs = [p for g in ref["self_groups"] for p in g[4]]
    n = ctypes.c_size_t(0)
    rid = vamp.lib.vmv_robot_id(name.encode())
    assert vamp.lib.vmv_robot_self_pairs(rid, ctypes.byref(n), None) == 0 and n.value == len(pairs)
    got = np.zeros((len(pairs), 2), np.uint16)
    vamp.lib.vmv_robot_self_pairs(rid, ctypes.byref(n), got.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)))
    assert got.tolist() == pairs
    orid = oracle.robot(name)
    assert oracle.n_spheres(orid) == ref["n_spheres"] and oracle.n_total_spheres(orid) == ref["n_total_spheres"]
    assert oracle.dimension(orid) == ref["dimension"] and oracle.resolution(orid) == ref["resolution"]


@pytest.mark.parametrize("name", ["panda", "ur5", "fetch", "baxter"])
def test_merged_gates_enclose_their_fine_spheres(oracle, name):
    """The primitive-only kernel variants gate several links at once (tools/gen_hip.py: merged_groups): one sphere,
    centred at one member's bounding centre, for all their fine spheres.  Their pruning is exact only if that sphere
    ENCLOSES every fine sphere of the group in every configuration — checked here on the oracle's fp32 FK (the arithmetic
    the kernels run), inside and far outside the joint bounds, to 2e-6 m (the gate's candidate margin is 1e-4 m)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_code
    import gen_hip

    m = gen_code.load(name)
    groups = gen_hip.merged_groups(m)
    merged = [g for g in groups if len(g["members"]) > 1]
    assert merged, "no rigid cluster found"
    # every fine sphere of the robot sits in exactly one gate, in the reference's order inside its link
    assert sorted(s for g in groups for s in g["fine"]) == list(range(m["n_spheres"]))
    assert [ln for g in groups for ln in g["members"]] == m["links"]
    rid = oracle.robot(name)
    lo, span = oracle.bounds(rid)
    rng = np.random.default_rng(7)
    q = (lo + span * rng.random((1500, len(lo)), dtype=np.float32)).astype(np.float32)
    q[::3] = (q[::3] * np.float32(2.7)).astype(np.float32)
    worst = -1.0
    for cfg in q:
        s = oracle.fk_all(rid, cfg).astype(np.float64)
        for g in merged:
            c = s[g["bound"], :3]
            reach = np.linalg.norm(s[g["fine"], :3] - c, axis=1) + s[g["fine"], 3]
            worst = max(worst, float(reach.max() - g["radius"]))
    assert worst <= 2e-6, worst
