"""The oracle against the reference's golden vectors and known answers (CPU; no GPU, no /root/reference)."""
import json
import os

import numpy as np
import pytest

from oracle_lib import CAGE_GOAL, CAGE_START, SPHERE_CAGE

ROBOTS = ["panda", "ur5", "fetch", "baxter"]


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_sin_cos_bit_exact_vs_reference_vector_hh(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "arith.npz"))
    assert np.array_equal(_bits(oracle.sin(g["x"])), _bits(g["sin"]))
    assert np.array_equal(_bits(oracle.cos(g["x"])), _bits(g["cos"]))


@pytest.mark.parametrize("dim", [6, 7, 8, 14])
def test_l2_norm_bit_exact(oracle, golden_dir, dim):
    g = np.load(os.path.join(golden_dir, "arith.npz"))
    got = np.array([oracle.l2_norm(v) for v in g[f"l2_in_{dim}"]], np.float32)
    assert np.array_equal(_bits(got), _bits(g[f"l2_out_{dim}"]))


@pytest.mark.parametrize("name", ROBOTS)
def test_fk_bit_exact_vs_reference_fkcc(oracle, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"fk_{name}.npz"))
    rid = oracle.robot(name)
    n_fine = int(g["n_fine"])
    assert oracle.n_spheres(rid) == n_fine
    assert oracle.n_total_spheres(rid) == g["spheres"].shape[1]
    for q, want in zip(g["q"], g["spheres"]):
        assert np.array_equal(_bits(oracle.fk_all(rid, q)), _bits(want))
        assert np.array_equal(_bits(oracle.fk(rid, q)), _bits(want[:n_fine]))  # sphere_fk == fkcc FK block


def test_panda_probe_sphere(oracle, golden_dir):
    k = json.load(open(os.path.join(golden_dir, "known_answers.json")))
    s = oracle.fk(oracle.robot("panda"), np.array(k["panda_probe_q"], np.float32))[58]
    assert np.allclose(s, k["panda_probe_sphere58"], atol=5e-7)


@pytest.fixture(scope="module")
def cage(oracle):
    e = oracle.env()
    for c in SPHERE_CAGE:
        e.add_sphere(*c, 0.2)
    return e


def test_cage_start_goal_edge(oracle, cage, golden_dir):
    k = json.load(open(os.path.join(golden_dir, "known_answers.json")))
    rid = oracle.robot("panda")
    assert oracle.validate(rid, cage, CAGE_START) == k["cage_start_valid"]
    assert oracle.validate(rid, cage, CAGE_GOAL) == k["cage_goal_valid"]
    assert oracle.validate_motion(rid, cage, CAGE_START, CAGE_GOAL) == k["cage_edge_valid"]


def test_cage_halton_known_answers(oracle, cage, golden_dir):
    """3,533 of the first 20,000 Halton samples valid; 142 of the 20,000 consecutive edges valid (reference)."""
    k = json.load(open(os.path.join(golden_dir, "known_answers.json")))
    h = np.load(os.path.join(golden_dir, "halton_panda.npz"))["samples"]
    assert np.allclose(h[0, :3], k["halton_first_sample_prefix"], atol=2e-6)
    rid = oracle.robot("panda")
    assert int(oracle.validate_batch(rid, cage, h[:20000]).sum()) == k["cage_halton_20000_valid"]
    assert int(oracle.validate_motion_batch(rid, cage, h[:20000], h[1:20001]).sum()) == k["cage_halton_20000_edges_valid"]


def mt19937_uniform_configs(lo, span, n):
    """std::mt19937(0) + uniform_real_distribution<float>: u = float(draw) / 2^32, per joint in order."""
    dim = len(lo)
    raw = np.random.RandomState(0).randint(0, 2 ** 32, n * dim, dtype=np.uint64)
    u = (raw.astype(np.float32) / np.float32(4294967296.0)).astype(np.float32)
    u[u >= 1] = np.nextafter(np.float32(1), np.float32(0))
    return (lo + span * u.reshape(n, dim)).astype(np.float32)


def test_cage_mt19937_known_answer(oracle, cage, golden_dir):
    """17,708 of 100,000 uniform mt19937(0) configurations valid in the sphere cage (reference)."""
    k = json.load(open(os.path.join(golden_dir, "known_answers.json")))
    rid = oracle.robot("panda")
    lo, span = oracle.bounds(rid)
    q = mt19937_uniform_configs(lo, span, 100000)
    assert int(oracle.validate_batch(rid, cage, q, threads=4).sum()) == k["cage_mt19937_seed0_100000_valid"]


@pytest.mark.parametrize("name", ROBOTS)
def test_validate_equals_replicated_rake(oracle, name):
    """validate(q) == fkcc on a rake holding q in all 8 lanes == validate_motion(q, q) at resolution 1."""
    from envs import make_env

    rid = oracle.robot(name)
    lo, span = oracle.bounds(rid)
    rng = np.random.default_rng(3)
    _, env = make_env("mixed", oracle, name)
    q = (lo + span * rng.random((200, len(lo)), dtype=np.float32)).astype(np.float32)
    for c in q:
        block = np.repeat(c[:, None], 8, axis=1)
        assert oracle.validate(rid, env, c) == oracle.fkcc_rake(rid, env, block)


def test_check_bounds(oracle):
    rid = oracle.robot("panda")
    e = oracle.env()
    lo, span = oracle.bounds(rid)
    mid = (lo + span * np.float32(0.5)).astype(np.float32)
    inside = oracle.validate(rid, e, mid, check_bounds=True)
    out = mid.copy()
    out[3] = lo[3] - np.float32(0.1)
    assert oracle.validate(rid, e, out, check_bounds=True) is False
    assert oracle.validate(rid, e, mid, check_bounds=False) == inside


@pytest.mark.parametrize("name", ROBOTS)
def test_avx2_rake_of_eight_build_equals_the_scalar_port(oracle, name):
    """bench.py's AVX2 cpu_baseline (8 distinct configurations per rake, per-lane masks) answers bit-identically to the
    scalar restatement, ragged tail and out-of-range joint values included"""
    from envs import build_oracle_env, spec_for

    if not oracle.has_avx2():
        pytest.skip("host CPU without AVX2")
    rid = oracle.robot(name)
    lo, span = oracle.bounds(rid)
    rng = np.random.default_rng(17)
    for kind in ("shell64", "mixed", "empty"):
        env = build_oracle_env(oracle, spec_for(kind, name))
        q = (lo + span * rng.random((20003, len(lo)), dtype=np.float32)).astype(np.float32)
        q[:16] *= np.float32(3.0)
        want = oracle.validate_batch(rid, env, q, threads=4)
        assert 0 < want.sum() < len(q)
        assert np.array_equal(oracle.validate_batch_avx2(rid, env, q, threads=1), want)
        assert np.array_equal(oracle.validate_batch_avx2(rid, env, q, threads=3), want)
    capt = build_oracle_env(oracle, spec_for("capt", name))
    with pytest.raises(ValueError):
        oracle.validate_batch_avx2(rid, capt, q[:64])


def test_batch_entry_points_call_non_finite_units_invalid(oracle):
    """The boundary rule of include/vamp_mvt_amd.h, mirrored by the oracle's BATCH entry points (vo_validate_batch,
    vo_validate_motion_batch, the AVX2 build): a NaN / +-inf joint makes the unit invalid.  vo_validate itself stays the
    plain restatement of the reference (whose answer for such input is an artefact of NaN sign propagation)."""
    rid = oracle.robot("panda")
    env = oracle.env()
    q = np.tile(np.asarray(CAGE_START, np.float32), (16, 1))
    assert oracle.validate_batch(rid, env, q).all()
    for i, s in enumerate([np.nan, -np.nan, np.inf, -np.inf]):
        q[2 * i + 1, (3 * i) % 7] = s
    want = np.ones(16, bool)
    want[[1, 3, 5, 7]] = False
    assert np.array_equal(oracle.validate_batch(rid, env, q), want)
    assert np.array_equal(oracle.validate_batch(rid, env, q, threads=4), want)
    clean = np.tile(np.asarray(CAGE_START, np.float32), (16, 1))
    assert np.array_equal(oracle.validate_motion_batch(rid, env, q, clean), want)
    assert np.array_equal(oracle.validate_motion_batch(rid, env, clean, q), want)
    if hasattr(oracle, "validate_batch_avx2") and oracle.has_avx2():
        assert np.array_equal(oracle.validate_batch_avx2(rid, env, q), want)
