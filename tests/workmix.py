"""Non-degenerate test inputs: configurations and edges of which a real share is valid AND a real share is not.

Test infrastructure.  Uniform samples of a 14-DoF (Baxter) or 8-DoF (Fetch) robot self-collide ~97 % / ~80 % of the
time, so a kernel that answers "invalid" for everything would pass a bit-exact comparison on them.  The generators
here search around configurations the ORACLE finds valid in the given environment (perturbations of a few widths, so
that many samples sit close to the validity boundary) and then pick a subset with a fixed valid share.  Seeds are
explicit integers (zlib.crc32 of the case name): the inputs are the same in every process."""
from __future__ import annotations

import zlib

import numpy as np

VALID_SHARE = 0.4  # of the returned configurations / edges


def case_seed(*parts):
    return zlib.crc32("/".join(str(p) for p in parts).encode())


def _scaled_noise(rng, shape, sigma, span):
    # sigma is in radians for a joint with the usual ~2 pi range; joints with a short range (Fetch's prismatic torso)
    # get proportionally less
    scale = np.minimum(span / np.float32(6.0), np.float32(1.0))
    return (rng.normal(0.0, sigma, shape) * scale).astype(np.float32)


def _pick(rng, valid_mask, n, share):
    vi, ii = np.flatnonzero(valid_mask), np.flatnonzero(~valid_mask)
    nv = min(len(vi), max(int(round(share * n)), n - len(ii)))
    ni = min(len(ii), n - nv)
    assert nv + ni == n, "not enough candidates"
    idx = np.concatenate([rng.choice(vi, nv, replace=False), rng.choice(ii, ni, replace=False)])
    rng.shuffle(idx)
    return idx


def valid_seeds(oracle, rid, oenv, rng, want=64, budget=400000):
    """configurations the oracle finds valid in `oenv`, by uniform search inside the joint bounds"""
    lo, span = oracle.bounds(rid)
    found, tried = [], 0
    while tried < budget and sum(len(f) for f in found) < want:
        q = (lo + span * rng.random((50000, len(lo)), dtype=np.float32)).astype(np.float32)
        v = oracle.validate_batch(rid, oenv, q, threads=8)
        found.append(q[v])
        tried += len(q)
    seeds = np.concatenate(found)
    assert len(seeds) > 0, "no valid configuration found for this (robot, environment): the case would be degenerate"
    return seeds


def mixed_configs(oracle, name, oenv, n, seed, share=VALID_SHARE):
    """-> (rid, q[n][dim], want[n]) with about `share` of the configurations valid (oracle answers)"""
    rid = oracle.robot(name)
    lo, span = oracle.bounds(rid)
    rng = np.random.default_rng(seed)
    seeds = valid_seeds(oracle, rid, oenv, rng)
    parts = [(lo + span * rng.random((3 * n, len(lo)), dtype=np.float32)).astype(np.float32)]
    for sigma in (0.03, 0.1, 0.3):
        base = seeds[rng.integers(len(seeds), size=n)]
        parts.append((base + _scaled_noise(rng, base.shape, sigma, span)).astype(np.float32))
    q = np.concatenate(parts)
    v = oracle.validate_batch(rid, oenv, q, threads=8)
    idx = _pick(rng, v, n, share)
    return rid, np.ascontiguousarray(q[idx]), v[idx]


def mixed_edges(oracle, name, oenv, n, seed, share=VALID_SHARE, zero_every=0):
    """-> (rid, a, b, want): edges from (mostly) valid starts with steps of several lengths; about `share` valid.
    zero_every > 0 makes every zero_every-th edge zero-length (n = 1, block = start)."""
    rid = oracle.robot(name)
    lo, span = oracle.bounds(rid)
    rng = np.random.default_rng(seed)
    seeds = valid_seeds(oracle, rid, oenv, rng)
    m = 3 * n
    a = seeds[rng.integers(len(seeds), size=m)]
    a = (a + _scaled_noise(rng, a.shape, 0.05, span)).astype(np.float32)
    sig = rng.choice(np.array([0.05, 0.15, 0.4, 1.0], np.float32), size=(m, 1))
    b = (a + _scaled_noise(rng, a.shape, 1.0, span) * sig).astype(np.float32)
    if zero_every:
        b[::zero_every] = a[::zero_every]
    v = oracle.validate_motion_batch(rid, oenv, a, b, threads=8)
    idx = _pick(rng, v, n, share)
    return rid, np.ascontiguousarray(a[idx]), np.ascontiguousarray(b[idx]), v[idx]
