"""Known answers the survey measured with the reference's own headers, as pins
for the CPU oracle and — marked gpu — as twins for the HIP path on the SAME inputs (generators: tests/pins.py).

What these pin that the sphere-cage answers do not: z-aligned cuboids, the sorted early-break with 64 primitives, the
1M-boolean hash of BASELINE config 2's shape, the CAPT build (vector counts) and the CAPT query against brute force."""
import os

import numpy as np
import pytest

import pins
from envs import build_oracle_env, build_product_env

PRIM64_VALID, PRIM64_HASH = 634173, "5cbe9a5badce93fe"
CAPT_VECTORS = {"panda": 24169, "fetch": 177408, "baxter": 873895}
CAPT_RADII = {"panda": (0.012, 0.08), "fetch": (0.012, 0.24), "baxter": (0.012, 0.5)}
BRUTE_FALSE_NEGATIVES, BRUTE_FALSE_POSITIVES = 4039, 0          # SURVEY.md §A.5 (Panda radii, r_point 0.0025)


@pytest.fixture(scope="module")
def prim64(oracle):
    rid = oracle.robot("panda")
    lo, span = oracle.bounds(rid)
    spec, q = pins.prim64_problem(1_000_000, lo, span)
    return rid, spec, q


# ----------------------------------------------------------------------------------------------- CPU: the oracle
def test_oracle_prim64_one_million_known_answer(oracle, prim64):
    """32 spheres + 32 z-aligned cuboids, 1,000,000 mt19937(0) Panda configurations: 634,173 valid, same 1M booleans
    (hash) as the reference compiled five different ways by the survey."""
    rid, spec, q = prim64
    v = oracle.validate_batch(rid, build_oracle_env(oracle, spec), q, threads=8)
    assert int(v.sum()) == PRIM64_VALID
    assert pins.fnv_bytes(v) == PRIM64_HASH


def test_oracle_booleans_do_not_depend_on_the_sqrt_deviation(oracle, prim64):
    """The oracle (and the product) use a correctly rounded sqrt for `max_extent` where the reference uses
    v * rsqrt_ps(v) (validity.hh:59, vector/avx.hh:411-415).  With the REFERENCE'S OWN approximate sqrt (oracle/_ref,
    compiled from the reference's vector.hh in place) installed in the oracle, the 1M booleans are identical."""
    import ctypes
    ref = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libref_vector.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    R = ctypes.CDLL(ref)
    rid, spec, q = prim64
    env = build_oracle_env(oracle, spec)
    hook = ctypes.cast(R.ref_sqrt_approx, ctypes.c_void_p)
    oracle.L.vo_set_max_extent_sqrt.argtypes = [ctypes.c_void_p]
    oracle.L.vo_set_max_extent_sqrt(hook)
    try:
        v = oracle.validate_batch(rid, env, q, threads=8)
    finally:
        oracle.L.vo_set_max_extent_sqrt(None)
    assert int(v.sum()) == PRIM64_VALID and pins.fnv_bytes(v) == PRIM64_HASH


def _vectors(oracle, cloud, name):
    e = oracle.env()
    e.add_capt(cloud, *CAPT_RADII[name], 0.0025)
    return e, e.capt()


@pytest.mark.parametrize("name", ["panda", "fetch", "baxter"])
def test_oracle_capt_affordance_vector_counts(oracle, name):
    """10,000-point shell cloud -> 24,169 / 177,408 / 873,895 affordance vectors.

    The survey's driver was compiled by g++ -march=native with its default -ffp-contract=fast, so `0.6f + 0.6f*u` and
    `0.2f + 1.3f*u` were single FMAs there; that cloud (fma=True) reproduces all three counts.  With separately
    rounded mul/add (fma=False) ~3,300 points move by one ulp: Panda and Fetch counts are unchanged and Baxter's is
    873,894 — one leaf (6117) holds a point whose distance to the cell is one ulp above (r_max + r_point)^2.  Tie order
    of the three duplicate-coordinate pairs of the cloud does not change any count (tools/capt_tie_study.py)."""
    _, c = _vectors(oracle, pins.capt_cloud(0, fma=True), name)
    assert c["nlog2"] == 14 and c["aff"].shape[1] == CAPT_VECTORS[name]
    if name != "fetch":  # keep the CPU suite short: the two informative cases
        _, c2 = _vectors(oracle, pins.capt_cloud(0, fma=False), name)
        assert c2["aff"].shape[1] == CAPT_VECTORS[name] - (1 if name == "baxter" else 0)


@pytest.fixture(scope="module")
def brute_problem():
    rng = pins.Mt19937Uniform(1, 3 * 10000 + 4 * 200000)
    cloud = pins.capt_cloud(rng=rng)
    c, r = pins.capt_queries(rng, 200000, *CAPT_RADII["panda"])
    return cloud, c, r


def test_oracle_capt_query_vs_brute_force(oracle, brute_problem):
    """200,000 sphere queries: the CAPT misses 4,039 collisions brute force finds (the build/query quirks of
    SURVEY.md §8a-8), reports none that brute force does not, and scalar == simd.  The build must reproduce THESE
    answers, not brute force."""
    cloud, c, r = brute_problem
    e, _ = _vectors(oracle, cloud, "panda")
    got = np.array([e.capt_collides(c[i], float(r[i])) for i in range(len(c))])
    brute = pins.brute_force_collides(cloud, c, r, 0.0025)
    assert int((brute & ~got).sum()) == BRUTE_FALSE_NEGATIVES
    assert int((got & ~brute).sum()) == BRUTE_FALSE_POSITIVES
    for s in range(0, 8000, 8):  # collides_simd on rakes of 8 queries == OR of the scalar answers
        assert e.capt_collides_simd(c[s:s + 8, 0], c[s:s + 8, 1], c[s:s + 8, 2], r[s:s + 8]) == bool(got[s:s + 8].any())


# ------------------------------------------------------------------------------------- GPU twins: the HIP path
@pytest.fixture(scope="module")
def device(vamp):
    assert vamp.device_count() >= 1, "no HIP device visible"
    vamp.set_device(0)
    return vamp


@pytest.mark.gpu
def test_gpu_prim64_one_million_known_answer(device, prim64):
    """the same 1M-boolean hash from vmv_validate_batch"""
    _, spec, q = prim64
    v = device.panda.validate_batch(q, build_product_env(spec))
    assert int(v.sum()) == PRIM64_VALID
    assert pins.fnv_bytes(v) == PRIM64_HASH


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["panda", "fetch", "baxter"])
@pytest.mark.parametrize("build", ["gpu", "host"])
def test_gpu_capt_arrays_equal_the_oracles(device, oracle, name, build):
    """device-built (vmv_env_add_capt_pointcloud_gpu) and host-built CAPT arrays == the ORACLE's arrays, word for word,
    on the survey's cloud at the BASELINE cloud size, with the survey's vector counts."""
    cloud = pins.capt_cloud(0, fma=True)
    _, want = _vectors(oracle, cloud, name)
    e = device.Environment()
    e.add_capt_pointcloud(cloud, *CAPT_RADII[name], 0.0025, build=build)
    got = e.host_tables()["capt"][0]
    assert got["nlog2"] == want["nlog2"] and got["aff"].shape[1] == CAPT_VECTORS[name]
    for key in ("tests", "aff_starts", "aabbs", "aff", "aabb_top"):
        assert got[key].shape == want[key].shape, key
        assert np.array_equal(np.ascontiguousarray(got[key]).view(np.uint32),
                              np.ascontiguousarray(want[key]).view(np.uint32)), key


@pytest.mark.gpu
def test_gpu_capt_query_vs_brute_force(device, oracle, brute_problem):
    """the 4,039 / 0 answer from the HIP CAPT query (vmv_spheres_in_collision_batch), equal to the oracle query by query"""
    cloud, c, r = brute_problem
    e = device.Environment()
    e.add_capt_pointcloud(cloud, *CAPT_RADII["panda"], 0.0025, build="gpu")
    got = e.spheres_in_collision(np.concatenate([c, r[:, None]], 1))
    oe, _ = _vectors(oracle, cloud, "panda")
    want = np.array([oe.capt_collides(c[i], float(r[i])) for i in range(len(c))])
    assert np.array_equal(got, want)
    brute = pins.brute_force_collides(cloud, c, r, 0.0025)
    assert int((brute & ~got).sum()) == BRUTE_FALSE_NEGATIVES and int((got & ~brute).sum()) == BRUTE_FALSE_POSITIVES
