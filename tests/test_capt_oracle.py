"""CAPT restatement: internal consistency + the survey's recorded behaviour (CPU only)."""
import numpy as np

from vamp_mvt_amd.workloads import shell_cloud


def test_capt_structure_and_queries(oracle):
    pts = shell_cloud(3000, seed=2)
    e = oracle.env()
    e.add_capt(pts, 0.012, 0.08, 0.0025)
    c = e.capt(0)
    n_leaves = 1 << c["nlog2"]
    assert n_leaves == 4096 and len(c["tests"]) == n_leaves - 1 and len(c["aff_starts"]) == n_leaves + 1
    assert c["aff_starts"][0] == 0 and c["aff_starts"][-1] == c["aff"].shape[1]
    assert np.all(np.diff(c["aff_starts"].astype(np.int64)) >= 0)
    assert np.array_equal(c["aabb_top"][:3], pts.min(0)) and np.array_equal(c["aabb_top"][3:], pts.max(0))
    # every real point is the representative of exactly one leaf
    reps = c["aff"][:, c["aff_starts"][:-1][np.diff(c["aff_starts"]) > 0], 0].T
    assert len(reps) == len(pts)
    assert set(map(tuple, reps.tolist())) == set(map(tuple, pts.tolist()))

    rng = np.random.default_rng(4)
    q = shell_cloud(4000, seed=9) + rng.normal(0, 0.05, (4000, 3)).astype(np.float32)
    r = rng.uniform(0.012, 0.08, 4000).astype(np.float32)
    scalar = np.array([e.capt_collides(q[i], r[i]) for i in range(len(q))])
    simd = np.array([e.capt_collides_simd(q[i:i + 1, 0], q[i:i + 1, 1], q[i:i + 1, 2], r[i:i + 1]) for i in range(len(q))])
    assert np.array_equal(scalar, simd)  # CAPT::collides == CAPT::collides_simd
    # rake result = OR of the 8 single-lane results (lanes are independent inside collides_simd)
    for i in range(0, 4000, 8):
        rake = e.capt_collides_simd(q[i:i + 8, 0], q[i:i + 8, 1], q[i:i + 8, 2], r[i:i + 8])
        assert rake == bool(simd[i:i + 8].any())
    # no false positives against brute force; false negatives exist by construction
    d = np.sqrt(((q[:, None, :] - pts[None, :, :]) ** 2).sum(-1)).min(1)
    brute = d <= (r + 0.0025) * (1 + 1e-5)
    assert not np.any(scalar & ~brute)
    assert scalar.sum() > 100
