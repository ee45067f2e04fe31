"""HIP path vs oracle, through the C ABI, on a real MI355X.  Bit-exact on every boolean and every FK word."""
import json
import os

import numpy as np
import pytest

from envs import make_env
from oracle_lib import CAGE_GOAL, CAGE_START
from test_oracle_pins import mt19937_uniform_configs
from workmix import case_seed, mixed_configs, mixed_edges

pytestmark = pytest.mark.gpu
ROBOTS = ["panda", "ur5", "fetch", "baxter"]
KINDS = ["empty", "cage", "shell64", "mixed", "capt", "clouds", "heightfield", "attach", "attach_free"]


def uniform_configs(oracle, name, n, seed):
    rid = oracle.robot(name)
    lo, span = oracle.bounds(rid)
    rng = np.random.default_rng(seed)
    return rid, (lo + span * rng.random((n, len(lo)), dtype=np.float32)).astype(np.float32)


@pytest.fixture(scope="module", autouse=True)
def _device(vamp):
    assert vamp.device_count() >= 1, "no HIP device visible"
    vamp.set_device(0)


def _non_degenerate(want, n):
    """every case must hold a real share of valid AND of invalid answers: an all-zero (or all-one) kernel fails it"""
    assert 0.05 * n < int(want.sum()) < 0.95 * n, f"degenerate case: {int(want.sum())} of {n} valid"


@pytest.mark.parametrize("name", ROBOTS)
@pytest.mark.parametrize("kind", KINDS)
def test_validate_batch_bit_exact(vamp, oracle, name, kind):
    """uniform configurations (whatever their validity) + configurations searched around valid postures (workmix)"""
    env, oenv = make_env(kind, oracle, name)
    n = 12000 if kind not in ("capt", "clouds") else 4000
    rid, q = uniform_configs(oracle, name, n, seed=case_seed(name, kind, "uniform") % 100000)
    got = getattr(vamp, name).validate_batch(q, env)
    want = oracle.validate_batch(rid, oenv, q, threads=8)
    assert got.dtype == bool and got.shape == (n,)
    assert np.array_equal(got, want)
    rid, q, want = mixed_configs(oracle, name, oenv, n, case_seed(name, kind, "mixed"))
    _non_degenerate(want, n)
    assert np.array_equal(getattr(vamp, name).validate_batch(q, env), want)


@pytest.mark.parametrize("name", ROBOTS)
@pytest.mark.parametrize("kind", KINDS)
def test_validate_motion_batch_bit_exact(vamp, oracle, name, kind):
    """Edges with distinct configurations per rake lane: exercises the 8-lane "any lane" gating.  Starts near valid
    postures, steps of several lengths, every 7th edge zero-length (n = 1, block = start)."""
    env, oenv = make_env(kind, oracle, name)
    n = 1500 if kind not in ("capt", "clouds") else 600
    rid, a, b, want = mixed_edges(oracle, name, oenv, n, case_seed(name, kind, "edges"), zero_every=7)
    _non_degenerate(want, n)
    assert np.array_equal(getattr(vamp, name).validate_motion_batch(a, b, env), want)
    # long random edges between uniform configurations (mostly invalid; early-outs at every rake index)
    rid, a = uniform_configs(oracle, name, n, seed=7)
    b = (a + np.random.default_rng(8).normal(0, 0.35, a.shape)).astype(np.float32)
    assert np.array_equal(getattr(vamp, name).validate_motion_batch(a, b, env),
                          oracle.validate_motion_batch(rid, oenv, a, b, threads=8))


@pytest.mark.parametrize("name", ROBOTS)
@pytest.mark.parametrize("kind", ["shell64", "mixed", "capt", "clouds", "attach", "heightfield"])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_edge_schedules_are_bit_exact(vamp, oracle, monkeypatch, name, kind, mode):
    """vmv_validate_motion_batch has four schedules (csrc/vmv_robot_tu.inc: launch_validate_motion): 0 = one rake group
    walks one edge, 1 = (edge, rake) tasks in two passes, 2 = tasks in doubling passes, 3 = two-pass tasks through the
    fused one-FK kernel (Panda / UR5 vs primitives; elsewhere it is schedule 1).  Each is forced here on edges of every length — zero-length ones, one-rake ones, edges of dozens of rakes — with
    ragged batch sizes around the 8-edge waves and 64-edge words, and must give the oracle's booleans."""
    monkeypatch.setenv("VMV_EDGE_TASKS", str(mode))
    env, oenv = make_env(kind, oracle, name)
    mod = getattr(vamp, name)
    n = 1100 if kind not in ("capt", "clouds") else 500
    rid, a, b, want = mixed_edges(oracle, name, oenv, n, case_seed(name, kind, "schedules"), zero_every=5)
    _non_degenerate(want, n)
    assert np.array_equal(mod.validate_motion_batch(a, b, env), want)
    for m in (1, 7, 8, 9, 63, 64, 65, 100):
        assert np.array_equal(mod.validate_motion_batch(a[:m], b[:m], env), want[:m])
    # long edges between valid postures (many rakes, collisions at every rake index) and between uniform configurations
    rid, q, ok = mixed_configs(oracle, name, oenv, 600, case_seed(name, kind, "schedule-ends"))
    ends = q[ok][:256] if ok.sum() >= 32 else q[:256]
    la, lb = ends[: len(ends) // 2], ends[len(ends) // 2: 2 * (len(ends) // 2)]
    assert np.array_equal(mod.validate_motion_batch(la, lb, env), oracle.validate_motion_batch(rid, oenv, la, lb, threads=8))
    rid, ua = uniform_configs(oracle, name, 300, seed=17)
    ub = (ua + np.random.default_rng(18).normal(0, 0.5, ua.shape)).astype(np.float32)
    assert np.array_equal(mod.validate_motion_batch(ua, ub, env), oracle.validate_motion_batch(rid, oenv, ua, ub, threads=8))


def test_edge_batches_from_two_host_threads_and_streams(vamp, oracle):
    """vmv_validate_motion_batch is a SEQUENCE of kernels that share per-(device, stream) scratch (csrc/vmv_edge_tasks.hip):
    two host threads, each on its own stream, and two threads on ONE stream, hammer it with batches of different sizes;
    every result must be the oracle's (ctypes releases the GIL during the call, so the sequences really race)."""
    import threading
    torch = pytest.importorskip("torch")
    env, oenv = make_env("shell64", oracle, "ur5")
    mod = vamp.ur5
    jobs = []
    for i, n in enumerate((300, 5000, 70000, 1100)):
        rid, a, b, want = mixed_edges(oracle, "ur5", oenv, min(n, 3000), case_seed("threads", str(i)), zero_every=7)
        reps = (n + len(a) - 1) // len(a)
        a, b, want = np.tile(a, (reps, 1))[:n], np.tile(b, (reps, 1))[:n], np.tile(want, reps)[:n]
        jobs.append((torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), want))
    errors = []

    def worker(stream, order):
        try:
            with torch.cuda.stream(stream):
                for _ in range(15):
                    for j in order:
                        ta, tb, want = jobs[j]
                        got = mod.validate_motion_batch(ta, tb, env).cpu().numpy()
                        if not np.array_equal(got, want):
                            errors.append(("mismatch", j, int((got != want).sum())))
        except Exception as e:  # noqa: BLE001
            errors.append(("exception", repr(e)))

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for streams in ((s1, s2), (s1, s1)):
        threads = [threading.Thread(target=worker, args=(streams[0], (0, 1, 2, 3))),
                   threading.Thread(target=worker, args=(streams[1], (3, 2, 1, 0)))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        assert not errors, errors[:3]
    assert vamp.lib.vmv_release_staging() == 0  # frees the streams' scratch; the next call allocates again
    ta, tb, want = jobs[0]
    assert np.array_equal(mod.validate_motion_batch(ta, tb, env).cpu().numpy(), want)


@pytest.mark.parametrize("name,kind", [("fetch", "config3"), ("baxter", "config5"), ("panda", "config3"), ("ur5", "config5")])
def test_baseline_point_cloud_configs_at_cloud_size(vamp, oracle, name, kind):
    """BASELINE config 3 (Fetch vs a 10,000-point CAPT cloud) and config 5 (Baxter edges vs 32 primitives + a
    10,000-point CAPT cloud) at the BASELINE cloud size, configurations AND edges, each with a real valid share."""
    env, oenv = make_env(kind, oracle, name)
    n = 6000
    rid, q, want = mixed_configs(oracle, name, oenv, n, case_seed(name, kind, "mixed"))
    _non_degenerate(want, n)
    assert np.array_equal(getattr(vamp, name).validate_batch(q, env), want)
    m = 1200
    rid, a, b, want_e = mixed_edges(oracle, name, oenv, m, case_seed(name, kind, "edges"), zero_every=11)
    _non_degenerate(want_e, m)
    assert np.array_equal(getattr(vamp, name).validate_motion_batch(a, b, env), want_e)


@pytest.mark.parametrize("kind", ["shell64", "mixed", "capt", "heightfield", "mvt", "many"])
def test_free_spheres_against_the_environment(vamp, oracle, kind):
    """sphere_environment_in_collision (validity.hh:47-158) for free spheres, one predicate at a time: every primitive
    type, the sorted early-break, heightfields and both point-cloud structures, without a robot in between."""
    import ctypes
    env, oenv = make_env(kind, oracle, "panda")
    rng = np.random.default_rng(case_seed("spheres", kind) % 100000)
    n = 20000
    s = np.concatenate([rng.uniform([-1.3, -1.3, -0.3], [1.3, 1.3, 1.6], (n, 3)), rng.uniform(0.005, 0.25, (n, 1))], 1)
    s = s.astype(np.float32)
    s[:64, :3] = 0.0  # the origin: max_extent = r exactly (where the reference's v * rsqrt(v) is NaN)
    got = env.spheres_in_collision(s)
    f = ctypes.POINTER(ctypes.c_float)
    want = np.array([bool(oracle.L.vo_sphere_environment_in_collision(oenv.h, s[i, :3].ctypes.data_as(f),
                                                                      ctypes.c_float(float(s[i, 3])))) for i in range(n)])
    assert np.array_equal(got, want)
    _non_degenerate(want, n)


@pytest.mark.parametrize("name", ROBOTS)
@pytest.mark.parametrize("n_points", [2, 37, 3000, 10000])
def test_capt_query_copy_changes_no_answer(vamp, oracle, monkeypatch, name, n_points):
    """The device walks a derived copy of the affordance arrays (each leaf's points sorted by their distance to the
    leaf's cell, cut per radius bucket: vmv_capt_build.h).  Radii from far below r_min to beyond r_max (the last bucket
    = the whole list), centres in and around the cloud, duplicated points: the answers must be the oracle's, and the
    same as with VMV_CAPT_NO_PREFIX=1 (every query walks its leaf's whole list, read at finalize) and with
    VMV_CAPT_NO_DIST_GRID=1 (no "farther than r + r_point from every cloud point" rejection in front of the descent)."""
    import ctypes
    from envs import build_oracle_env, build_product_env
    from vamp_mvt_amd.workloads import POINT_RADIUS, RADII, shell_cloud
    r_min, r_max = RADII[name]
    k = 1.6 if name == "baxter" else 1.0
    pts = shell_cloud(n_points, case_seed("captcopy", name, n_points) % 100000, 0.5 * k, 1.2 * k, 0.0, 1.5)
    if n_points >= 37:
        m = min(len(pts[::7]), len(pts[3::7]))
        pts[::7][:m] = pts[3::7][:m]  # equal points: equal keys in the sort
    spec = [("capt", (pts, r_min, r_max, POINT_RADIUS))]
    oenv = build_oracle_env(oracle, spec)
    rng = np.random.default_rng(case_seed("captcopy-q", name, n_points) % 100000)
    n = 30000
    c = pts[rng.integers(len(pts), size=n)] + rng.normal(0, 0.5 * r_max, (n, 3))
    c[: n // 10] = rng.uniform([-2, -2, -0.5], [2, 2, 2], (n // 10, 3))
    r = np.concatenate([rng.uniform(0.2 * r_min, r_max, n - 2000), rng.uniform(r_max, 1.6 * r_max, 2000)])
    s = np.concatenate([c, r[:, None]], 1).astype(np.float32)
    s[5, 3], s[6, :3] = np.nan, np.inf
    f = ctypes.POINTER(ctypes.c_float)
    want = np.array([bool(oracle.L.vo_sphere_environment_in_collision(oenv.h, s[i, :3].ctypes.data_as(f),
                                                                      ctypes.c_float(float(s[i, 3])))) for i in range(n)])
    rid, q = uniform_configs(oracle, name, 6000, seed=case_seed("captcopy-cfg", name, n_points) % 100000)
    want_q = oracle.validate_batch(rid, oenv, q, threads=8)
    for no_prefix, no_grid in (("0", "0"), ("0", "1"), ("1", "0")):
        monkeypatch.setenv("VMV_CAPT_NO_PREFIX", no_prefix)
        if no_grid == "1":
            monkeypatch.setenv("VMV_CAPT_NO_DIST_GRID", "1")  # no distance grid in front of the descent
        else:
            monkeypatch.delenv("VMV_CAPT_NO_DIST_GRID", raising=False)
        env = build_product_env(spec)
        assert np.array_equal(env.spheres_in_collision(s), want), (no_prefix, no_grid)
        assert np.array_equal(getattr(vamp, name).validate_batch(q, env), want_q), (no_prefix, no_grid)
    if n_points >= 3000:
        _non_degenerate(want, n)


@pytest.mark.parametrize("name", ROBOTS)
@pytest.mark.parametrize("kind", ["shell64", "cage", "capt", "attach", "empty"])
def test_grouped_self_collision_kernel_is_bit_exact(vamp, oracle, monkeypatch, name, kind):
    """The self-collision kernel gives every wave 1..8 validity words and works through the configurations still valid in
    them, 64 per pass (the launcher picks the group size from the batch size; VMV_SELF_GROUP forces it here).  Ragged
    sizes, every group size: the oracle's answers."""
    env, oenv = make_env(kind, oracle, name)
    for n in (1, 63, 64, 65, 4097, 20000):
        rid, q = uniform_configs(oracle, name, n, seed=case_seed(name, kind, "grouped", n) % 100000)
        q[::13] = (q[::13] * np.float32(1.6)).astype(np.float32)
        want = oracle.validate_batch(rid, oenv, q, threads=8)
        for group in ("1", "2", "3", "5", "8"):
            monkeypatch.setenv("VMV_SELF_GROUP", group)
            assert np.array_equal(getattr(vamp, name).validate_batch(q, env), want), (n, group)
        monkeypatch.delenv("VMV_SELF_GROUP")
        assert np.array_equal(getattr(vamp, name).validate_batch(q, env), want), n


@pytest.mark.parametrize("name", ["panda", "ur5", "baxter"])
def test_reach_certificates_change_no_answer(vamp, oracle, monkeypatch, name):
    """vmv_env_finalize skips the first links of the chain for environments none of whose primitives they can ever touch
    (sample centres + slack per link, tools/gen_hip.py: link_samples).  Spheres and cuboids pushed towards the base from
    far outside until they cross the links' reach — certified, borderline and touching environments — and configurations
    beyond the joint bounds: the oracle's answers, with the certificates and with VMV_NO_LINK_SKIP=1."""
    from envs import build_oracle_env, build_product_env
    from vamp_mvt_amd.workloads import yaw_cuboid
    rid, q = uniform_configs(oracle, name, 4000, seed=case_seed("reach", name) % 100000)
    q[::7] = (q[::7] * np.float32(2.3)).astype(np.float32)  # any joint value: the certificates do not assume the bounds
    z0 = {"panda": 0.33, "ur5": 1.0, "baxter": 0.4}[name]
    rng = np.random.default_rng(5)
    some_valid = False
    for radial in (0.9, 0.6, 0.45, 0.36, 0.31, 0.27, 0.22, 0.15):
        spec = []
        for k in range(10):
            a = rng.uniform(0, 2 * np.pi)
            c = np.array([radial * np.cos(a), radial * np.sin(a), z0 + rng.uniform(-0.15, 0.15)], np.float32)
            if name == "baxter":
                c[:2] += np.array([0.064, -0.259 if k % 2 else 0.259], np.float32)  # around the shoulders
            if k % 2:
                spec.append(("sphere", np.array([*c, 0.05], np.float32)))
            else:
                spec.append(("cuboid", yaw_cuboid(c, rng.uniform(0, 6.28), np.array([0.05, 0.03, 0.04], np.float32))))
        oenv = build_oracle_env(oracle, spec)
        want = oracle.validate_batch(rid, oenv, q, threads=8)
        some_valid = some_valid or bool(want.any())
        for off in (None, "1"):
            if off:
                monkeypatch.setenv("VMV_NO_LINK_SKIP", off)
            else:
                monkeypatch.delenv("VMV_NO_LINK_SKIP", raising=False)
            env = build_product_env(spec)
            assert np.array_equal(getattr(vamp, name).validate_batch(q, env), want), (radial, off)
            a, b = q[:600], (q[:600] + rng.normal(0, 0.2, q[:600].shape)).astype(np.float32)
            assert np.array_equal(getattr(vamp, name).validate_motion_batch(a, b, env),
                                  oracle.validate_motion_batch(rid, oenv, a, b, threads=8)), (radial, off)
    assert some_valid


@pytest.mark.parametrize("name", ROBOTS)
@pytest.mark.parametrize("kind", ["empty", "shell64"])
def test_clustered_waves_bit_exact(vamp, oracle, name, kind):
    """Adversarial for the kernels' in-wave work lists: every lane of a wave holds (nearly) the same configuration,
    so whichever gate fires, fires on all 64 lanes at once (fullest item lists, all rounds, all flush paths)."""
    env, oenv = make_env(kind, oracle, name)
    rid, base = uniform_configs(oracle, name, 96, seed=11)
    rng = np.random.default_rng(12)
    q = np.repeat(base, 64, axis=0)
    q[64 * 32:] += rng.normal(0, 2e-3, q[64 * 32:].shape).astype(np.float32)  # second third: tiny jitter per lane
    got = getattr(vamp, name).validate_batch(q, env)
    want = oracle.validate_batch(rid, oenv, q, threads=8)
    assert np.array_equal(got, want)
    # the same clusters as edges: 8 identical rakes per wave, moving together
    a = q[::8][:768]
    b = (a + np.repeat(rng.normal(0, 0.3, (a.shape[0] // 8, a.shape[1])), 8, axis=0)).astype(np.float32)
    got_e = getattr(vamp, name).validate_motion_batch(a, b, env)
    want_e = oracle.validate_motion_batch(rid, oenv, a, b)
    assert np.array_equal(got_e, want_e)


@pytest.mark.parametrize("name", ROBOTS)
def test_fk_bit_exact_vs_golden_and_oracle(vamp, oracle, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"fk_{name}.npz"))
    n_fine = int(g["n_fine"])
    got = getattr(vamp, name).fk_batch(g["q"])
    assert got.shape == (len(g["q"]), n_fine, 4)
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(g["spheres"][:, :n_fine]).view(np.uint32))
    assert np.abs(got[..., :3] - g["spheres"][:, :n_fine, :3]).max() <= 1e-5  # north star tolerance (met at 0)
    spheres = getattr(vamp, name).fk(g["q"][3])
    assert len(spheres) == n_fine and abs(spheres[5].x - float(g["spheres"][3, 5, 0])) == 0.0


@pytest.mark.parametrize("name", ROBOTS)
def test_eefk_bit_exact_vs_reference_outputs(vamp, oracle, golden_dir, name):
    """<robot>.eefk against the end-effector frame of the reference's generated fkcc_attach (tests/golden, `ee`)."""
    g = np.load(os.path.join(golden_dir, f"fk_{name}.npz"))
    got = getattr(vamp, name).eefk_batch(g["q"])
    want = np.zeros((len(g["q"]), 4, 4), np.float32)
    want[:, :3, 3] = g["ee"][:, :3]
    want[:, :3, :3] = g["ee"][:, 3:].reshape(-1, 3, 3).transpose(0, 2, 1)  # the reference stores it column-major
    want[:, 3, 3] = 1.0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(getattr(vamp, name).eefk(g["q"][5]), want[5])


@pytest.mark.parametrize("name", ROBOTS)
def test_attachment_changes_answers_and_detach_restores_them(vamp, oracle, name):
    env, oenv = make_env("attach", oracle, name)
    rid, q = uniform_configs(oracle, name, 8000, seed=21)
    with_att = getattr(vamp, name).validate_batch(q, env)
    assert np.array_equal(with_att, oracle.validate_batch(rid, oenv, q, threads=8))
    env.detach()
    oenv.detach()
    without = getattr(vamp, name).validate_batch(q, env)
    assert np.array_equal(without, oracle.validate_batch(rid, oenv, q, threads=8))
    assert (without & ~with_att).sum() > 0 and (with_att & ~without).sum() == 0  # the attachment only removes validity


@pytest.mark.parametrize("name", ["panda", "fetch"])
def test_contact_report_matches_oracle_predicates(vamp, oracle, name):
    """<robot>.debug (Robot::fkcc_debug): per-sphere object lists = the oracle's exact predicate against each object on
    its own, self pairs = overlapping fine pairs; consistent with validate()."""
    import ctypes
    env, oenv = make_env("mixed", oracle, name)
    tables = env.host_tables()
    rid, q = uniform_configs(oracle, name, 40, seed=33)
    robot = getattr(vamp, name)
    f = ctypes.POINTER(ctypes.c_float)
    singles = {}
    for lst, adder in (("spheres", lambda e, p: e.add_sphere(*[float(v) for v in p[:4]])),
                       ("capsules", lambda e, p: e.add_capsule(p[:8])), ("z_capsules", lambda e, p: e.add_capsule(p[:8])),
                       ("cuboids", lambda e, p: e.add_cuboid(p[:15])), ("z_cuboids", lambda e, p: e.add_cuboid(p[:15]))):
        for i, p in enumerate(tables[lst]):
            e = oracle.env()
            adder(e, np.ascontiguousarray(p, np.float32))
            singles[(lst, i)] = e
    any_env = any_self = 0
    for c in q:
        per_sphere, pairs = robot.debug(c, env)
        spheres = robot.fk_batch(c[None, :])[0]
        assert len(per_sphere) == robot.n_spheres()
        for s, hits in enumerate(per_sphere):
            want = [k for k, e in singles.items()
                    if oracle.L.vo_sphere_environment_in_collision(e.h, np.ascontiguousarray(spheres[s, :3]).ctypes.data_as(f),
                                                                   ctypes.c_float(float(spheres[s, 3])))]
            assert sorted(hits) == sorted(want)
            any_env += len(hits)
        d = np.linalg.norm(spheres[:, None, :3] - spheres[None, :, :3], axis=2)
        for a, b in pairs:
            assert d[a, b] < spheres[a, 3] + spheres[b, 3] + 1e-6
        any_self += len(pairs)
        if robot.validate(c, env):
            assert not pairs and not any(per_sphere)
    assert any_env > 0 and any_self > 0


def test_reference_known_answers_on_gpu(vamp, oracle, golden_dir):
    """The reference's own known answers, computed by the HIP path."""
    k = json.load(open(os.path.join(golden_dir, "known_answers.json")))
    env, _ = make_env("cage", oracle)
    p = vamp.panda
    assert p.validate(CAGE_START, env) is True and p.validate(CAGE_GOAL, env) is True
    assert p.validate_motion(CAGE_START, CAGE_GOAL, env) is False
    h = np.load(os.path.join(golden_dir, "halton_panda.npz"))["samples"]
    assert int(p.validate_batch(h[:20000], env).sum()) == k["cage_halton_20000_valid"]
    assert int(p.validate_motion_batch(h[:20000], h[1:20001], env).sum()) == k["cage_halton_20000_edges_valid"]
    q = mt19937_uniform_configs(p.lower_bounds(), p.upper_bounds() - p.lower_bounds(), 100000)
    assert int(p.validate_batch(q, env).sum()) == k["cage_mt19937_seed0_100000_valid"]


def test_edge_cases(vamp, oracle):
    env, oenv = make_env("shell64", oracle)
    p = vamp.panda
    rid, q = uniform_configs(oracle, "panda", 1000, seed=1)
    assert p.validate_batch(q[:0], env).shape == (0,)                      # empty batch
    for n in (1, 7, 63, 64, 65, 127, 129, 1000):                            # ragged tails of the 64-lane waves
        assert np.array_equal(p.validate_batch(q[:n], env), oracle.validate_batch(rid, oenv, q[:n]))
    for n in (1, 7, 8, 9, 31, 33):                                          # ragged tails of the 8-edge waves
        a, b = q[:n], q[n:2 * n]
        assert np.array_equal(p.validate_motion_batch(a, b, env), oracle.validate_motion_batch(rid, oenv, a, b))
    # check_bounds (robot_helper.hh:258-262)
    out = q[0].copy()
    out[3] = p.lower_bounds()[3] - np.float32(0.2)
    assert p.validate(out, vamp.Environment(), check_bounds=True) is False
    assert p.validate(out, vamp.Environment(), check_bounds=True) == oracle.validate(rid, oracle.env(), out, True)
    # configurations outside the bounds are still evaluated exactly
    wild = (q[:256] * np.float32(3.0)).astype(np.float32)
    assert np.array_equal(p.validate_batch(wild, env), oracle.validate_batch(rid, oenv, wild))
    # default environment argument = empty environment (self-collision only)
    assert np.array_equal(p.validate_batch(q), oracle.validate_batch(rid, oracle.env(), q))
    with pytest.raises(TypeError):
        p.validate_batch(q[:, :6], env)


@pytest.mark.parametrize("name", ROBOTS)
@pytest.mark.parametrize("kind", ["empty", "shell64", "capt"])
def test_non_finite_joints_are_invalid_by_rule(vamp, oracle, name, kind):
    """include/vamp_mvt_amd.h: a configuration with a NaN / +-inf joint is INVALID, an edge with such an endpoint too; the
    finite units of the same waves / rakes keep the reference's answers (the oracle's batch entry points carry the same
    rule; VERDICT r2 item 6 — before it a third of such inputs differed, profiles/r02_nonfinite_inputs_survey.txt)."""
    env, oenv = make_env(kind, oracle, name)
    mod = getattr(vamp, name)
    n = 3000
    rid, q, want_clean = mixed_configs(oracle, name, oenv, n, case_seed(name, kind, "nonfinite"))
    rng = np.random.default_rng(case_seed(name, kind, "nonfinite-where") % 100000)
    specials = np.array([np.nan, -np.nan, np.inf, -np.inf], np.float32)
    dirty = q.copy()
    rows = rng.choice(n, n // 3, replace=False)
    dirty[rows, rng.integers(0, q.shape[1], rows.size)] = specials[rng.integers(0, 4, rows.size)]
    dirty[rows[:50]] = np.float32(np.nan)  # every joint
    got = mod.validate_batch(dirty, env)
    poisoned = np.zeros(n, bool)
    poisoned[rows] = True
    assert not got[poisoned].any()
    assert np.array_equal(got[~poisoned], want_clean[~poisoned]) and want_clean[~poisoned].any()
    assert np.array_equal(got, oracle.validate_batch(rid, oenv, dirty))
    assert mod.validate(dirty[rows[0]], env) is False
    # edges: start, goal or both
    m = 1200
    rid, a, b, want_e = mixed_edges(oracle, name, oenv, m, case_seed(name, kind, "nonfinite-edges"), zero_every=9)
    da, db = a.copy(), b.copy()
    e_rows = rng.choice(m, m // 3, replace=False)
    for i, r in enumerate(e_rows):
        tgt = (da, db, da)[i % 3]
        tgt[r, rng.integers(0, a.shape[1])] = specials[i % 4]
        if i % 3 == 2:
            db[r, rng.integers(0, a.shape[1])] = specials[(i + 1) % 4]
    got_e = mod.validate_motion_batch(da, db, env)
    bad_e = np.zeros(m, bool)
    bad_e[e_rows] = True
    assert not got_e[bad_e].any()
    assert np.array_equal(got_e[~bad_e], want_e[~bad_e]) and want_e[~bad_e].any()
    assert np.array_equal(got_e, oracle.validate_motion_batch(rid, oenv, da, db))


@pytest.mark.parametrize("name", ROBOTS)
@pytest.mark.parametrize("n", [1, 37, 64, 100, 4097])
def test_self_collision_stage_alone_ignores_bits_beyond_n(vamp, oracle, name, n):
    """vmv_validate_batch_self is an entry point of its own and ANDs into the caller's words.  With all-ones
    words and n % 64 != 0 it must neither read configurations past the end of d_q nor leave bits >= n set: the result is
    the self-collision answer (= validity in the empty environment) of the n configurations, tail bits 0."""
    import ctypes
    torch = pytest.importorskip("torch")
    rid, q = uniform_configs(oracle, name, n, seed=n)
    tq = torch.from_numpy(q).cuda()  # exactly n rows: anything read beyond them is outside the allocation's payload
    words = torch.full(((n + 63) // 64,), -1, dtype=torch.int64, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = vamp.lib.vmv_validate_batch_self(rid_of(vamp, name), ctypes.c_void_p(tq.data_ptr()), n,
                                          ctypes.c_void_p(words.data_ptr()), stream)
    assert rc == 0
    torch.cuda.synchronize()
    bits = np.unpackbits(words.cpu().numpy().view(np.uint8), bitorder="little")
    assert not bits[n:].any()
    assert np.array_equal(bits[:n].astype(bool), oracle.validate_batch(rid, oracle.env(), q))


def rid_of(vamp, name):
    return vamp.lib.vmv_robot_id(name.encode())


def test_large_primitive_lists_and_capacity(vamp, oracle):
    env, oenv = make_env("many", oracle)
    rid, q = uniform_configs(oracle, "panda", 8000, seed=4)
    assert np.array_equal(vamp.panda.validate_batch(q, env), oracle.validate_batch(rid, oenv, q, threads=8))
    a, b = q[:800], q[800:1600]
    assert np.array_equal(vamp.panda.validate_motion_batch(a, b, env), oracle.validate_motion_batch(rid, oenv, a, b))
    # more primitive records than the 48 KiB on-chip staging budget: a status code, not a crash or a fallback
    big = vamp.Environment()
    for i in range(2000):
        big.add_sphere(vamp.Sphere([2.0 + 0.01 * i, 0.0, 0.0], 0.01))
    with pytest.raises(vamp.VmvError) as ei:
        vamp.panda.validate_batch(q[:64], big)
    assert ei.value.status == 4  # VMV_ERR_CAPACITY


@pytest.mark.parametrize("name", ["panda", "baxter"])
@pytest.mark.parametrize("counts", [
    (5, 5, 5, 5, 5),        # five short lists: one shared word instead of five
    (30, 30, 30, 20, 12),   # packs into exactly four words, the last list ends on bit 31 of a shared word
    (32, 1, 31, 3, 29),     # a list of exactly one word, then lists that fill words to the last bit
    (20, 20, 5, 40, 10),    # a two-word list between short ones: five words even when shared -> the counted loops
    (0, 17, 0, 16, 40),     # empty lists take no bits; the two-word list is last
    (33, 9, 9, 9, 9),       # a two-word list first, four short lists sharing the next word
])
def test_shared_candidate_word_layouts(vamp, oracle, name, counts, monkeypatch):
    """Environments with more lists than candidate words (vmv_device.h kMaskWords = 4): lists of at most 32 primitives share
    words (EnvDev::wshift_*, vmv_api.hip finalize).  Every layout — shared, word-aligned, too long for either (counted
    loops) — with the broad-phase grid and without it (VMV_NO_GRID: the counted-loop gate writes the shared words) gives
    the oracle's booleans, configurations and edges."""
    from envs import build_oracle_env, build_product_env, counted_spec

    spec = counted_spec(name, counts, seed=case_seed(name, "counted", str(counts)) % 100000)
    oenv = build_oracle_env(oracle, spec)
    rid, q, want = mixed_configs(oracle, name, oenv, 6000, case_seed(name, "counted", "configs"))
    _non_degenerate(want, len(q))
    rid, a, b, want_e = mixed_edges(oracle, name, oenv, 500, case_seed(name, "counted", "edges"), zero_every=9)
    for no_grid in (False, True):
        if no_grid:
            monkeypatch.setenv("VMV_NO_GRID", "1")
        env = build_product_env(spec)
        mod = getattr(vamp, name)
        assert np.array_equal(mod.validate_batch(q, env), want), f"configurations, no_grid={no_grid}"
        assert np.array_equal(mod.validate_motion_batch(a, b, env), want_e), f"edges, no_grid={no_grid}"


@pytest.mark.parametrize("name", ["panda", "baxter"])
def test_counted_loop_gate_without_the_grid(vamp, oracle, name, monkeypatch):
    """VMV_NO_GRID=1: the bounding-sphere pass falls back to the counted sorted loops (the path environments take whose
    grid cannot be built); same answers."""
    monkeypatch.setenv("VMV_NO_GRID", "1")
    env, oenv = make_env("shell64", oracle, name, seed=3)  # a fresh environment: the robot part is built on first use
    rid, q = uniform_configs(oracle, name, 6000, seed=8)
    assert np.array_equal(getattr(vamp, name).validate_batch(q, env), oracle.validate_batch(rid, oenv, q, threads=8))
    a, b = q[:400], q[400:800]
    assert np.array_equal(getattr(vamp, name).validate_motion_batch(a, b, env), oracle.validate_motion_batch(rid, oenv, a, b))


def test_ill_formed_primitives_take_the_full_loops(vamp, oracle):
    """canonical parameters are the caller's: cuboid axes that are not unit vectors (or not orthogonal) and a capsule
    whose rdv is not 1 / |v|^2 make the distance expressions non-1-Lipschitz, which the candidate pruning assumes; such
    environments must give the reference's full-loop answers all the same"""
    from envs import build_oracle_env, build_product_env, spec_for
    spec = spec_for("mixed", "panda", seed=5)
    out = []
    for k, (kind, p) in enumerate(spec):
        p = np.array(p, np.float32)
        if kind == "cuboid" and k % 2 == 0:
            p[3:6] *= np.float32(1.7)           # a stretched axis
            p[6:9] += np.float32(0.4) * p[9:12]  # and a sheared one
        if kind == "capsule" and k % 3 == 0:
            p[7] *= np.float32(0.45)
        out.append((kind, p))
    env, oenv = build_product_env(out), build_oracle_env(oracle, out)
    rid, q, want = mixed_configs(oracle, "panda", oenv, 8000, case_seed("ill-formed"))
    _non_degenerate(want, 8000)
    assert np.array_equal(vamp.panda.validate_batch(q, env), want)
    rid, a, b, want_e = mixed_edges(oracle, "panda", oenv, 800, case_seed("ill-formed", "edges"))
    assert np.array_equal(vamp.panda.validate_motion_batch(a, b, env), want_e)


def test_environment_rebuild_after_mutation(vamp, oracle):
    env, oenv = make_env("cage", oracle)
    rid, q = uniform_configs(oracle, "panda", 4000, seed=2)
    first = vamp.panda.validate_batch(q, env)
    env.add_sphere(vamp.Sphere([0.0, 0.0, 1.1], 0.25))
    oenv.add_sphere(0.0, 0.0, 1.1, 0.25)
    second = vamp.panda.validate_batch(q, env)
    assert np.array_equal(second, oracle.validate_batch(rid, oenv, q)) and second.sum() < first.sum()


def test_full_size_properties(vamp, oracle):
    """BASELINE config 2 at full size (1,048,576 configs): size-independent properties + sampled oracle parity."""
    torch = pytest.importorskip("torch")
    env, oenv = make_env("shell64", oracle)
    p = vamp.panda
    n = 1 << 20
    rid, q = uniform_configs(oracle, "panda", n, seed=11)
    tq = torch.from_numpy(q).cuda()
    v = p.validate_batch(tq, env)
    assert v.dtype == torch.bool and v.shape == (n,)
    v = v.cpu().numpy()
    assert 0.55 < v.mean() < 0.70                                  # ~63 % valid
    # permutation equivariance: the answer for a configuration does not depend on its position / wave
    perm = np.random.default_rng(0).permutation(n)
    assert np.array_equal(p.validate_batch(tq[torch.from_numpy(perm).cuda()].contiguous(), env).cpu().numpy(), v[perm])
    # batch-split invariance
    assert np.array_equal(p.validate_batch(tq[123457:654321].contiguous(), env).cpu().numpy(), v[123457:654321])
    # a zero-length edge is the configuration check (validate_motion<.., 1>(q, q) is what validate() calls)
    sub = tq[: 1 << 16].contiguous()
    assert np.array_equal(p.validate_motion_batch(sub, sub, env).cpu().numpy(), v[: 1 << 16])
    # monotonicity: removing obstacles can only make configurations valid
    empty = p.validate_batch(tq, vamp.Environment()).cpu().numpy()
    assert not np.any(v & ~empty)
    # sampled oracle parity at full size
    idx = np.random.default_rng(1).choice(n, 50000, replace=False)
    assert np.array_equal(v[idx], oracle.validate_batch(rid, oenv, q[idx], threads=8))
    # host-buffer path == device-buffer path
    assert np.array_equal(p.validate_batch(q[:200000], env), v[:200000])


def test_fused_one_fk_kernel_is_bit_exact_too(oracle):
    """VMV_FUSED_KERNEL=1 (opt-in; measured slower): both halves of fkcc along one walk of the chain, one
    FK per configuration.  Read once per process, so the fused path runs in a child process."""
    import subprocess
    import sys
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import vamp_mvt_amd as vamp
from oracle_lib import Oracle
from envs import make_env
from workmix import case_seed, mixed_configs
vamp.set_device(0)
o = Oracle()
for name in ("panda", "ur5"):
    for kind in ("shell64", "mixed", "cage", "empty"):
        env, oenv = make_env(kind, o, name)
        rid, q, want = mixed_configs(o, name, oenv, 8000, case_seed(name, kind, "fused"))
        assert np.array_equal(getattr(vamp, name).validate_batch(q, env), want), (name, kind)
        assert np.array_equal(getattr(vamp, name).validate_batch(q[:77], env), want[:77]), (name, kind)
print("fused ok")
""" % (os.path.join(os.path.dirname(__file__), ".."), os.path.dirname(__file__))
    env = dict(os.environ, VMV_FUSED_KERNEL="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fused ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("cfg", ["config3", "config4", "config5", "config4_uniform_starts"])
def test_full_size_baseline_configs(vamp, oracle, cfg):
    """BASELINE configs 3, 4, 5 at their full sizes (1M Fetch configurations vs a 10k-point CAPT cloud; 1M UR5 edges vs
    64 primitives; 262,144 Baxter edges vs 32 primitives + a 10k-point CAPT cloud): size-independent properties and
    sampled oracle parity, on the generators of tools/bench_configs.py.  Edge batches are roadmap-shaped (SURVEY.md
    §8d-4/5; planning/prm.hh:109-145, fcit.hh:137-260 only ever validate edges between VALID samples): config 4 joins
    valid Halton samples to valid neighbours at U[0.2, 1.5] rad, config 5 is the 8-nearest-neighbour graph over valid
    Halton samples; `config4_uniform_starts` keeps the generator of rounds 1-2 (uniform, mostly invalid starts)."""
    torch = pytest.importorskip("torch")
    from envs import build_oracle_env, build_product_env, spec_for
    from vamp_mvt_amd.workloads import knn_shaped_edges, prm_shaped_edges, shell_spec

    base = cfg.split("_")[0]
    name, n, edges = {"config3": ("fetch", 1 << 20, False), "config4": ("ur5", 1 << 20, True),
                      "config5": ("baxter", 1 << 18, True)}[base]
    spec = shell_spec(0) if base == "config4" else spec_for(base, name)
    env, oenv = build_product_env(spec), build_oracle_env(oracle, spec)
    mod = getattr(vamp, name)
    rid = oracle.robot(name)
    lo, span = oracle.bounds(rid)
    rng = np.random.default_rng(case_seed(cfg) % 100000)
    if cfg == "config4":
        ta, tb = prm_shaped_edges(mod, env, n, 0.2, 1.5, seed=case_seed(cfg) % 100000)
        a, b = ta.cpu().numpy(), tb.cpu().numpy()
    elif cfg == "config5":
        ta, tb = knn_shaped_edges(mod, env, n, 8)
        a, b = ta.cpu().numpy(), tb.cpu().numpy()
    else:
        a = (lo + span * rng.random((n, len(lo)), dtype=np.float32)).astype(np.float32)
        if edges:  # uniform starts, random directions
            d = rng.normal(size=a.shape).astype(np.float32)
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            b = (a + d * rng.uniform(0.2, 1.5, (n, 1)).astype(np.float32)).astype(np.float32)
    if cfg in ("config4", "config5"):  # roadmap-shaped: both endpoints valid, so edges are walked, not dropped at rake 0
        assert mod.validate_batch(ta, env).all() and mod.validate_batch(tb, env).all()
    ta = torch.from_numpy(a).cuda()
    tb = torch.from_numpy(b).cuda() if edges else None

    def run(x, y=None):
        return (mod.validate_motion_batch(x, y, env) if edges else mod.validate_batch(x, env)).cpu().numpy()

    v = run(ta, tb)
    assert v.shape == (n,) and 0.005 * n < v.sum() < 0.995 * n
    if cfg in ("config4", "config5"):
        assert v.mean() > 0.05  # a real share of edges is walked to its last rake
    # permutation equivariance and batch-split invariance: a unit's answer does not depend on its position / wave / rake
    perm = torch.from_numpy(np.random.default_rng(1).permutation(n)).cuda()
    assert np.array_equal(run(ta[perm].contiguous(), tb[perm].contiguous() if edges else None), v[perm.cpu().numpy()])
    lo_i, hi_i = 100003, 100003 + 333333 if n > 500000 else 100003 + 77777
    assert np.array_equal(run(ta[lo_i:hi_i].contiguous(), tb[lo_i:hi_i].contiguous() if edges else None), v[lo_i:hi_i])
    # monotonicity: without the environment only self-collisions remain, so nothing valid may become invalid
    empty = vamp.Environment()
    free = (mod.validate_motion_batch(ta, tb, empty) if edges else mod.validate_batch(ta, empty)).cpu().numpy()
    assert not np.any(v & ~free)
    if edges:  # an edge whose every rake is valid has a valid goal configuration (lane 7 of the first rake is the goal)
        goal_ok = mod.validate_batch(tb, env).cpu().numpy()
        assert not np.any(v & ~goal_ok)
        for mode in ("0", "1", "2", "3"):  # the edge schedules give the same words at full size (1M edges = one slice)
            os.environ["VMV_EDGE_TASKS"] = mode
            try:
                assert np.array_equal(run(ta, tb), v), f"VMV_EDGE_TASKS={mode}"
            finally:
                del os.environ["VMV_EDGE_TASKS"]
        if cfg == "config4":  # more than one slice of 2^20 edges through the task schedule
            os.environ["VMV_EDGE_TASKS"] = "1"
            try:
                xa, xb = torch.cat([ta, ta[:777]]).contiguous(), torch.cat([tb, tb[:777]]).contiguous()
                assert np.array_equal(run(xa, xb), np.concatenate([v, v[:777]]))
            finally:
                del os.environ["VMV_EDGE_TASKS"]
    # sampled oracle parity at full size
    m = 40000 if cfg == "config3" else (20000 if base == "config4" else 4000)
    idx = np.random.default_rng(2).choice(n, m, replace=False)
    want = oracle.validate_motion_batch(rid, oenv, a[idx], b[idx], threads=8) if edges else \
        oracle.validate_batch(rid, oenv, a[idx], threads=8)
    assert np.array_equal(v[idx], want)
