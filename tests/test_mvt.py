"""Multi-level Voxel Table (collision/mvt.hh): oracle consistency, product host build vs oracle,
and the HIP query vs the oracle on the GPU."""
import ctypes

import numpy as np
import pytest

from envs import WORKSPACE, make_env
from vamp_mvt_amd.workloads import shell_cloud

PANDA = (0.012, 0.08, *WORKSPACE["panda"], 0.0025)


def test_oracle_mvt_structure_and_queries(oracle):
    pts = shell_cloud(1500, seed=3, rmin=0.5, rmax=1.0, zmin=0.0, zmax=1.2)
    e = oracle.env()
    assert e.add_mvt(pts, *PANDA) == 0
    m = e.mvt(0)
    assert m["grid_width"] == int(np.floor(np.float32(2.38) / np.float32(0.08))) and m["capacity"] == 64
    assert np.array_equal(m["global_box"][:3], pts.min(0)) and np.array_equal(m["global_box"][3:], pts.max(0))
    rng = np.random.default_rng(5)
    q = (shell_cloud(6000, seed=11, rmin=0.4, rmax=1.1, zmin=-0.1, zmax=1.3) +
         rng.normal(0, 0.03, (6000, 3))).astype(np.float32)
    r = rng.uniform(0.012, 0.08, 6000).astype(np.float32)
    scalar = np.array([e.mvt_collides(q[i], r[i]) for i in range(len(q))])
    for i in range(0, 6000, 8):  # collides_simd == OR of its lanes' scalar answers
        assert e.mvt_collides_simd(q[i:i + 8, 0], q[i:i + 8, 1], q[i:i + 8, 2], r[i:i + 8]) == bool(scalar[i:i + 8].any())
    # for radii up to r_max the +-1-cell walk sees every point within reach: the table answers exactly like brute force
    d2 = ((q[:, None, :].astype(np.float32) - pts[None, :, :]) ** 2).sum(-1).min(1)
    brute = d2 <= (r + np.float32(0.0025)) ** 2
    margin = np.abs(np.sqrt(d2) - (r + 0.0025)) > 1e-5  # away from the fp32 knife edge
    assert np.array_equal(scalar[margin], brute[margin]) and scalar.sum() > 200
    # radii above r_max (bounding spheres) are clamped to one cell: misses are allowed, false hits are not
    big = np.array([e.mvt_collides(q[i], 0.3) for i in range(2000)])
    brute_big = np.sqrt(d2[:2000]) <= 0.3 + 0.0025
    assert not np.any(big & ~brute_big)


def test_mvt_pool_limits_match_between_product_and_oracle(vamp, oracle):
    """Where the reference would throw in its noexcept constructor, both sides report the same reason."""
    from vamp_mvt_amd._lib import lib

    def product(pts, *args):
        h = ctypes.c_void_p()
        lib.vmv_env_create(ctypes.byref(h))
        reason = ctypes.c_int(0)
        p = np.ascontiguousarray(pts, np.float32)
        lo, hi = np.array(args[2], np.float32), np.array(args[3], np.float32)
        rc = lib.vmv_env_add_mvt_pointcloud(h, p.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(p), args[0],
                                            args[1], lo.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                            hi.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), args[4], None,
                                            ctypes.byref(reason))
        info = None
        if rc == 0:
            gw, cap, nv = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
            isf = ctypes.c_float()
            box = (ctypes.c_float * 6)()
            lib.vmv_env_mvt_info(h, 0, ctypes.byref(gw), ctypes.byref(cap), ctypes.byref(nv), ctypes.byref(isf), box)
            info = (gw.value, cap.value, nv.value, np.float32(isf.value), np.array(list(box), np.float32))
        lib.vmv_env_destroy(h)
        return rc, reason.value, info

    cases = [
        (shell_cloud(1500, 3, 0.5, 1.0, 0.0, 1.2), PANDA),                     # fits
        (shell_cloud(10000, 3), PANDA),                                         # pools run out
        (np.random.default_rng(0).uniform(-1, 1, (5000, 3)).astype(np.float32), PANDA),   # > 10 % of voxels
        (np.tile(np.array([[0.5, 0.5, 0.5]], np.float32), (70, 1)) + np.float32(1e-4) * np.arange(70)[:, None], PANDA),
        (shell_cloud(300, 4), (0.012, 0.24, *WORKSPACE["fetch"], 0.0025)),
        (shell_cloud(50, 4), (0.012, 0.5, *WORKSPACE["baxter"], 0.0025)),
    ]
    outcomes = []
    for pts, args in cases:
        e = oracle.env()
        want = e.add_mvt(pts, *args)
        rc, reason, info = product(pts, *args)
        outcomes.append(want)
        assert (rc == 0) == (want == 0) and reason == want
        if want == 0:
            m = e.mvt(0)
            assert info[0] == m["grid_width"] and info[1] == m["capacity"] and info[2] == m["n_voxels"]
            assert info[3] == np.float32(m["inverse_scale_factor"]) and np.array_equal(info[4], m["global_box"])
        else:
            assert rc == 4  # VMV_ERR_CAPACITY instead of std::terminate
    assert 0 in outcomes and any(o != 0 for o in outcomes)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["panda", "ur5", "fetch", "baxter"])
def test_mvt_environment_bit_exact_on_gpu(vamp, oracle, name):
    vamp.set_device(0)
    env, oenv = make_env("mvt", oracle, name)
    rid = oracle.robot(name)
    lo, span = oracle.bounds(rid)
    rng = np.random.default_rng(12)
    q = (lo + span * rng.random((8000, len(lo)), dtype=np.float32)).astype(np.float32)
    got = getattr(vamp, name).validate_batch(q, env)
    want = oracle.validate_batch(rid, oenv, q, threads=8)
    assert np.array_equal(got, want)
    a = q[:800]
    b = (a + rng.normal(0, 0.3, a.shape)).astype(np.float32)
    assert np.array_equal(getattr(vamp, name).validate_motion_batch(a, b, env),
                          oracle.validate_motion_batch(rid, oenv, a, b))
    _, empty = make_env("empty", oracle, name)
    assert want.sum() < oracle.validate_batch(rid, empty, q, threads=8).sum()  # the cloud does something
