"""Generators of the survey's recorded known-answer experiments.

Test infrastructure.  The survey measured these answers with the reference's own headers; the inputs are defined by a
`std::mt19937` stream + `std::uniform_real_distribution<float>(0, 1)` and libm's `cosf`/`sinf`, restated here so the
CPU oracle and the HIP path can be run on exactly the same numbers:

  * 64-primitive environment + 1,000,000 Panda configurations  -> 634,173 valid, FNV hash 5cbe9a5badce93fe
  * 10,000-point shell cloud -> CAPT affordance-vector counts 24,169 / 177,408 / 873,895 (Panda / Fetch / Baxter radii)
  * 200,000 sphere queries vs that cloud -> 4,039 false negatives / 0 false positives against brute force
"""
from __future__ import annotations

import ctypes
import ctypes.util

import numpy as np

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.cosf.restype = _libm.sinf.restype = ctypes.c_float
_libm.cosf.argtypes = _libm.sinf.argtypes = [ctypes.c_float]
F = np.float32


class Mt19937Uniform:
    """std::mt19937(seed) feeding uniform_real_distribution<float>(0, 1) (libstdc++: one 32-bit draw per float,
    float(draw) / 2^32, results that round up to 1 are replaced by nextafter(1, 0))."""

    def __init__(self, seed, count):
        raw = np.random.RandomState(seed).randint(0, 2 ** 32, count, dtype=np.uint64)
        u = (raw.astype(np.float32) / np.float32(4294967296.0)).astype(np.float32)
        u[u >= 1] = np.nextafter(np.float32(1), np.float32(0))
        self.u, self.pos = u, 0

    def take(self, n):
        out = self.u[self.pos:self.pos + n]
        assert out.size == n, "stream too short"
        self.pos += n
        return out

    def one(self):
        return self.take(1)[0]


def _fma(a, b, c):
    # a * b is exact in double (24 x 24 bits); the sum is rounded to double, then to float — equal to the single
    # rounding of a hardware FMA except in double-rounding ties that do not occur in these streams' ranges
    return F(np.float64(F(a)) * np.float64(F(b)) + np.float64(F(c)))


def _ring_point(rng, r0, rs, z0, zs, fma=False):
    a = F(6.2831853) * rng.one()
    if fma:
        rr = _fma(rs, rng.one(), r0)
        z = _fma(zs, rng.one(), z0)
    else:
        rr = F(r0) + F(rs) * rng.one()
        z = F(z0) + F(zs) * rng.one()
    return F(rr * F(_libm.cosf(a))), F(rr * F(_libm.sinf(a))), F(z)


def prim64_problem(n_configs=1_000_000, lo=None, span=None):
    """-> (spec, q): 32 spheres + 32 z-aligned cuboids (canonical 15-float form, identity axes) and n Panda
    configurations s_a + s_m * u, all from ONE mt19937(0) stream (SURVEY.md §A.2 / VERDICT r1 item 1a)."""
    rng = Mt19937Uniform(0, 32 * 4 + 32 * 6 + n_configs * 7)
    spec = []
    for _ in range(32):
        x, y, z = _ring_point(rng, 0.45, 0.5, 0.0, 1.2)
        r = F(0.03) + F(0.05) * rng.one()
        spec.append(("sphere", np.array([x, y, z, r], np.float32)))
    for _ in range(32):
        x, y, z = _ring_point(rng, 0.45, 0.5, 0.0, 1.2)
        h = [F(0.03) + F(0.05) * rng.one() for _ in range(3)]
        spec.append(("cuboid", np.array([x, y, z, 1, 0, 0, 0, 1, 0, 0, 0, 1, *h], np.float32)))
    u = rng.take(n_configs * 7).reshape(n_configs, 7)
    q = (lo + span * u).astype(np.float32)
    return spec, q


def fnv_bytes(valid):
    """h = 1469598103934665603; h = (h ^ b) * 1099511628211 over the 0/1 bytes (the survey's variant of FNV-1a)."""
    h = 1469598103934665603
    mask = (1 << 64) - 1
    for b in np.asarray(valid, np.uint8).tolist():
        h = ((h ^ b) * 1099511628211) & mask
    return f"{h:016x}"


def capt_cloud(seed=0, n=10000, fma=False, rng=None):
    """10,000 points: a = 2pi u, rr = 0.6 + 0.6 u, z = 0.2 + 1.3 u -> (rr cosf(a), rr sinf(a), z).
    fma=True evaluates `0.6f + 0.6f*u` and `0.2f + 1.3f*u` as single fused multiply-adds, which is what the survey's
    driver did (g++ -march=native contracts by default); rng continues an existing stream instead of seeding one."""
    rng = rng or Mt19937Uniform(seed, 3 * n)
    return np.array([_ring_point(rng, 0.6, 0.6, 0.2, 1.3, fma) for _ in range(n)], np.float32)


def capt_queries(rng, n, r_min, r_max):
    """n sphere queries c = (2.6u - 1.3, 2.6u - 1.3, 1.7u), r = r_min + (r_max - r_min) u, continuing `rng`."""
    u = rng.take(4 * n).reshape(n, 4)
    c = np.stack([F(2.6) * u[:, 0] - F(1.3), F(2.6) * u[:, 1] - F(1.3), F(1.7) * u[:, 2]], 1).astype(np.float32)
    r = (F(r_min) + F(F(r_max) - F(r_min)) * u[:, 3]).astype(np.float32)
    return c, r


def brute_force_collides(cloud, c, r, r_point):
    """collides iff some point lies within r + r_point of the centre (fp32, inclusive, the query's own arithmetic:
    sql2_3(point, centre) <= (r + r_point)^2).  A KD-tree only pre-selects the nearest candidates."""
    from scipy.spatial import cKDTree

    cloud = np.ascontiguousarray(cloud, np.float32)
    _, nn = cKDTree(cloud.astype(np.float64)).query(c.astype(np.float64), k=4)
    p = cloud[nn]                                    # [n][4][3]
    d = (p - c[:, None, :]).astype(np.float32)
    d2 = ((d[..., 0] * d[..., 0]).astype(np.float32) + (d[..., 1] * d[..., 1]).astype(np.float32)).astype(np.float32)
    d2 = (d2 + (d[..., 2] * d[..., 2]).astype(np.float32)).astype(np.float32)
    rr = (r + F(r_point)).astype(np.float32)
    return (d2 <= (rr * rr).astype(np.float32)[:, None]).any(1)
