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
