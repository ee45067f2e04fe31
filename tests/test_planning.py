"""Planner harness: Halton restatement on CPU; RRT-Connect / batched PRM on the GPU path."""
import os

import numpy as np
import pytest

from envs import make_env
from oracle_lib import CAGE_GOAL, CAGE_START


def test_halton_bit_exact_vs_reference(vamp, golden_dir):
    from vamp_mvt_amd.planning import Halton

    want = np.load(os.path.join(golden_dir, "halton_panda.npz"))["samples"]
    h = Halton(vamp.panda)
    got = h.batch(4096)
    assert np.array_equal(got.view(np.uint32), want[:4096].view(np.uint32))
    h.reset()
    h.skip(20000)
    assert np.array_equal(h.next().view(np.uint32), want[20000].view(np.uint32))


@pytest.mark.gpu
def test_rrtc_solves_the_sphere_cage(vamp, oracle):
    """BASELINE config 1 plumbing: sphere_cage_example start -> goal (straight edge is invalid, so it must plan)."""
    from vamp_mvt_amd.planning import Halton, RRTCSettings, rrtc, validate_path

    vamp.set_device(0)
    env, oenv = make_env("cage", oracle)
    res = rrtc(vamp.panda, CAGE_START, CAGE_GOAL, env, RRTCSettings(range=1.0), Halton(vamp.panda))
    assert res.solved and res.iterations > 0
    assert np.allclose(res.path[0], CAGE_START) and np.allclose(res.path[-1], CAGE_GOAL)
    assert validate_path(vamp.panda, res.path, env)
    rid = oracle.robot("panda")
    for a, b in zip(res.path[:-1], res.path[1:]):  # every segment is valid for the oracle too
        assert oracle.validate_motion(rid, oenv, a, b)


@pytest.mark.gpu
def test_fcit_batch_loop_solves_the_sphere_cage(vamp, oracle):
    """lazy complete-graph search (the FCIT* loop in batch form): solves the cage, every path edge holds for the oracle,
    and optimisation never makes the path longer"""
    from vamp_mvt_amd.planning import FCITSettings, Halton, fcit

    vamp.set_device(0)
    env, oenv = make_env("cage", oracle)
    res = fcit(vamp.panda, CAGE_START, CAGE_GOAL, env, FCITSettings(batch_size=400, max_samples=4000), Halton(vamp.panda))
    assert res.solved and res.edges_checked > 0 and len(res.path) >= 3
    rid = oracle.robot("panda")
    for a, b in zip(res.path[:-1], res.path[1:]):
        assert oracle.validate_motion(rid, oenv, a, b)
    better = fcit(vamp.panda, CAGE_START, CAGE_GOAL, env,
                  FCITSettings(batch_size=400, max_samples=2400, optimize=True, max_iterations=8), Halton(vamp.panda))
    assert better.solved and better.cost <= res.cost + 1e-6
    for a, b in zip(better.path[:-1], better.path[1:]):
        assert oracle.validate_motion(rid, oenv, a, b)
    # a blocked goal is reported as unsolved, not as an exception
    blocked = vamp.Environment()
    blocked.add_sphere(vamp.Sphere([0.3, 0.0, 0.5], 0.6))
    assert not fcit(vamp.panda, CAGE_START, CAGE_GOAL, blocked, FCITSettings(max_iterations=2)).solved


@pytest.mark.gpu
def test_batched_roadmap_edges_match_oracle(vamp, oracle):
    from vamp_mvt_amd.planning import Halton, build_roadmap

    vamp.set_device(0)
    env, oenv = make_env("cage", oracle)
    rm = build_roadmap(vamp.panda, env, n_samples=1500, k=6, sampler=Halton(vamp.panda),
                       extra_vertices=[CAGE_START, CAGE_GOAL])
    rid = oracle.robot("panda")
    assert len(rm.vertices) > 100 and rm.candidate_edges > len(rm.edges) > 0
    assert oracle.validate_batch(rid, oenv, rm.vertices).all()
    sel = rm.edges[:: max(1, len(rm.edges) // 300)]
    assert oracle.validate_motion_batch(rid, oenv, rm.vertices[sel[:, 0]], rm.vertices[sel[:, 1]]).all()
    path = rm.shortest_path(0, 1)
    if path is not None:
        for a, b in zip(path[:-1], path[1:]):
            assert oracle.validate_motion(rid, oenv, rm.vertices[a], rm.vertices[b])


@pytest.mark.gpu
def test_device_halton_bit_exact_vs_reference(vamp, golden_dir):
    pytest.importorskip("torch")
    vamp.set_device(0)
    want = np.load(os.path.join(golden_dir, "halton_panda.npz"))["samples"]
    got = vamp.panda.halton_device(20001).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    part = vamp.panda.halton_device(100, skip=12345).cpu().numpy()
    assert np.array_equal(part.view(np.uint32), want[12345:12445].view(np.uint32))


# ---- the reference's own entry points (scripts/sphere_cage_example.py:45-90 with `import vamp_mvt_amd as vamp`) ----------
def _sphere_cage_example_body(vamp, n_trials, planner="rrtc", variation=0.01, radius=0.2, sampler_name="halton",
                              skip_rng_iterations=0, **kwargs):
    """The body of the reference example's main(), benchmark branch, statement for statement (problem data restated:
    start a, goal b, 14 sphere centres).  Returns the per-trial dicts and the summary frame."""
    import copy
    import random

    import pandas as pd

    a = [0., -0.785, 0., -2.356, 0., 1.571, 0.785]
    b = [2.35, 1., 0., -0.8, 0, 2.5, 0.785]
    from oracle_lib import SPHERE_CAGE as problem

    (vamp_module, planner_func, plan_settings,
     simp_settings) = vamp.configure_robot_and_planner_with_kwargs("panda", planner, **kwargs)

    sampler = getattr(vamp_module, sampler_name)()
    sampler.skip(skip_rng_iterations)

    random.seed(0)
    np.random.seed(0)

    results = []
    spheres = [np.array(sphere) for sphere in problem]
    for _ in range(n_trials):
        random.shuffle(spheres)
        spheres_copy = copy.deepcopy(spheres)

        e = vamp.Environment()
        for sphere in spheres_copy:
            sphere += np.random.uniform(low=-variation, high=variation, size=(3, ))
            e.add_sphere(vamp.Sphere(sphere, radius))

        if vamp.panda.validate(a, e) and vamp.panda.validate(b, e):
            result = planner_func(a, b, e, plan_settings, sampler)
            simple = vamp_module.simplify(result.path, e, simp_settings, sampler)
            results.append(vamp.results_to_dict(result, simple))
            assert result.solved and result.path.validate(e) and simple.path.validate(e)
            assert simple.path.cost() <= result.path.cost() + 1e-5

    df = pd.DataFrame.from_dict(results)
    df["planning_time"] = df["planning_time"].dt.microseconds
    df["simplification_time"] = df["simplification_time"].dt.microseconds
    stats = df[["planning_time", "simplification_time", "initial_path_cost", "simplified_path_cost",
                "planning_iterations"]].describe()
    return results, stats


@pytest.mark.gpu
def test_sphere_cage_example_runs_unchanged_and_solves_every_trial(vamp):
    """BASELINE config 1 through the reference's API shape: configure_robot_and_planner_with_kwargs, <robot>.halton(),
    planner_func(start, goal, env, settings, sampler) -> PlanningResult, simplify, results_to_dict"""
    vamp.set_device(0)
    n_trials = 12
    results, stats = _sphere_cage_example_body(vamp, n_trials)
    assert len(results) == n_trials and all(r["solved"] for r in results)
    assert float(stats.loc["count", "planning_iterations"]) == n_trials
    assert all(r["initial_path_vertices"] >= 3 for r in results)  # the straight edge is invalid: it really plans


@pytest.mark.gpu
@pytest.mark.parametrize("planner", ["prm", "fcit"])
def test_prm_and_fcit_entry_points(vamp, planner):
    vamp.set_device(0)
    results, _ = _sphere_cage_example_body(vamp, 3, planner=planner, **({"batch_size": 400} if planner == "fcit" else {}))
    assert len(results) == 3 and all(r["solved"] for r in results)


def test_reference_shaped_names_exist(vamp):
    """everything scripts/sphere_cage_example.py and src/vamp/__init__.py:1-51 name (no GPU needed)"""
    for name in ("configure_robot_and_planner_with_kwargs", "problem_dict_to_vamp", "results_to_dict", "Environment",
                 "Attachment", "Sphere", "Cuboid", "Cylinder", "RRTCSettings", "PRMSettings", "PRMNeighborParams",
                 "FCITSettings", "FCITNeighborParams", "AORRTCSettings", "SimplifySettings", "SimplifyRoutine",
                 "filter_pointcloud", "png_to_heightfield", "robots"):
        assert hasattr(vamp, name), name
    assert list(vamp.robots) == ["panda", "ur5", "fetch", "baxter"] and "panda" in vamp.robots
    for robot in vamp.robots:
        m = getattr(vamp, robot)
        for name in ("dimension", "resolution", "n_spheres", "space_measure", "min_max_radii", "joint_names", "end_effector",
                     "halton", "xorshift", "Path", "PlanningResult", "Roadmap", "fk", "eefk", "debug", "validate",
                     "simplify", "roadmap", "rrtc", "prm", "fcit"):
            assert hasattr(m, name), (robot, name)
        rng = m.halton()
        first = rng.next()
        rng.skip(5)
        rng.reset()
        assert np.array_equal(rng.next(), first) and first.shape == (m.dimension(),)
        with pytest.raises(RuntimeError):
            m.xorshift()
    mod, fn, ps, ss = vamp.configure_robot_and_planner_with_kwargs("panda", "rrtc", range=0.7)
    assert mod is vamp.panda and fn is vamp.panda.rrtc and ps.range == 0.7 and ps.max_iterations == 1000000
    _, _, ps, _ = vamp.configure_robot_and_planner_with_kwargs("ur5", "prm")
    assert ps.max_neighbors(1000) == int(np.ceil((np.e + np.e / 6) * np.log(1000.0)))
    with pytest.raises(ValueError):
        vamp.configure_robot_and_planner_with_kwargs("panda", "nonsense")
    p = vamp.panda.Path()
    assert p.cost() == float("inf")
    p.append([0.0] * 7)
    p.append([0.5] * 7)
    p.insert(1, [0.25] * 7)
    assert len(p) == 3 and abs(p.cost() - np.sqrt(7) * 0.5) < 1e-6 and p.numpy().shape == (3, 7)
    p.interpolate_to_resolution(32)
    assert len(p) > 3 and np.allclose(p[0], 0.0) and np.allclose(p[-1], 0.5)
    res = vamp.panda.PlanningResult()
    assert not res.solved and res.nanoseconds == 0
