#!/usr/bin/env python3
"""Offline study: how many affordance vectors does a CAPT query need if a leaf's points are sorted by their distance to the
leaf's k-d CELL (a lower bound of the distance to any query centre that descends to this leaf) and the query stops at
the first vector whose nearest point is farther than r + r_point?  (oracle = test infrastructure; numpy only)

    python tools/experiments/capt_prefix_study.py [fetch|baxter|panda] [n_configs]"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle  # noqa: E402
from vamp_mvt_amd.workloads import POINT_RADIUS, RADII, shell_cloud  # noqa: E402

robot = sys.argv[1] if len(sys.argv) > 1 else "fetch"
ncfg = int(sys.argv[2]) if len(sys.argv) > 2 else 400
o = Oracle()
env = o.env()
pts = shell_cloud(10000, 3) if robot != "baxter" else shell_cloud(10000, 4, 1.0, 1.8)
env.add_capt(pts, *RADII[robot], POINT_RADIUS)
v = env.capt()
tests, starts, aabbs = v["tests"], v["aff_starts"], v["aabbs"].reshape(-1, 6)
ax, ay, az = v["aff"]
nlog2 = int(v["nlog2"]); n_tests = len(tests); n_leaves = n_tests + 1
print(robot, "vectors", len(ax), "leaves", n_leaves, "nlog2", nlog2)

# cell bounds per leaf from the implicit tree
lo = np.full((2 * n_tests + 1, 3), -np.inf)
hi = np.full((2 * n_tests + 1, 3), np.inf)
for i in range(n_tests):
    depth = int(np.floor(np.log2(i + 1)))
    k = depth % 3
    for child, side in ((2 * i + 1, 0), (2 * i + 2, 1)):
        lo[child], hi[child] = lo[i].copy(), hi[i].copy()
        if side == 0:
            hi[child][k] = tests[i]
        else:
            lo[child][k] = tests[i]
clo, chi = lo[n_tests:], hi[n_tests:]

# per vector: min over its 8 points of dist(point, cell); sorted within the leaf
leaf_of_vec = np.repeat(np.arange(n_leaves), np.diff(starts))
P = np.stack([ax, ay, az], -1).astype(np.float64)  # [vec][8][3]
d = np.maximum(np.maximum(clo[leaf_of_vec][:, None, :] - P, P - chi[leaf_of_vec][:, None, :]), 0.0)
dist_pt = np.sqrt((d * d).sum(-1))  # [vec][8]; padding (inf/nan) -> inf
dist_pt = np.where(np.isfinite(dist_pt), dist_pt, np.inf)
# re-pack: sort the POINTS of a leaf by distance, then chunk into vectors of 8 -> per-vector key = its first (nearest) point
keys_sorted = []
for L in range(n_leaves):
    s, e = starts[L], starts[L + 1]
    dd = np.sort(dist_pt[s:e].reshape(-1))
    keys_sorted.append(dd[::8])
print("cell extents (median):", np.median(np.where(np.isfinite(chi - clo), chi - clo, np.nan), 0) if False else "")

rid = o.robot(robot)
lob, span = o.bounds(rid)
rng = np.random.default_rng(0)
q = (lob + span * rng.random((ncfg, len(lob)), dtype=np.float32)).astype(np.float32)
nf = o.n_spheres(rid)
tot_full = tot_pref = nq = nq_pass = 0
tot_full_b = tot_pref_b = 0
BUCKETS = (16, 32, 64)
t0 = RADII[robot][0] + POINT_RADIUS
tot_bucket = {B: [0, 0] for B in BUCKETS}
hist = np.zeros(12, int)
top = v["aabb_top"]
for c in q:
    S = o.fk_all(rid, c)
    for si, (x, y, z, r) in enumerate(S):
        if not (x + r >= top[0] and x - r <= top[3] and y + r >= top[1] and y - r <= top[4] and z + r >= top[2] and z - r <= top[5]):
            continue
        idx, k = 0, 0
        p = (x, y, z)
        for _ in range(nlog2):
            idx = 2 * idx + 1 + (1 if p[k] >= tests[idx] else 0)
            k = (k + 1) % 3
        zi = idx - n_tests
        bb = aabbs[zi]
        dd = np.array(p) - np.clip(np.array(p), bb[:3], bb[3:])
        rr = r + POINT_RADIUS
        nq += 1
        if (dd * dd).sum() > rr * rr:
            continue
        nq_pass += 1
        full = starts[zi + 1] - starts[zi]
        pref = int(np.searchsorted(keys_sorted[zi], rr + 1e-4, side="right"))
        for B in BUCKETS:
            step = (RADII[robot][1] - RADII[robot][0]) / (B - 1)
            b = min(max(int(np.floor((rr + 1e-4 - t0) / step)) + 1, 0), B - 1)
            T = np.inf if b == B - 1 else t0 + b * step
            c_ = int(np.searchsorted(keys_sorted[zi], T, side="right"))
            tot_bucket[B][si >= nf] += c_
            if B == 32:
                hist[min(c_, 11)] += 1
        if si >= nf:
            tot_full_b += full; tot_pref_b += pref
        else:
            tot_full += full; tot_pref += pref
print(f"queries {nq} pass-leaf {nq_pass}")
print(f"fine spheres: full {tot_full} prefix {tot_pref} ({tot_pref / max(tot_full, 1):.3f})")
for B in BUCKETS:
    print(f"  {B} uniform radius buckets: fine {tot_bucket[B][0]} ({tot_bucket[B][0] / max(tot_full, 1):.3f}) bounding {tot_bucket[B][1]} ({tot_bucket[B][1] / max(tot_full_b, 1):.3f})")
print(f"bounding    : full {tot_full_b} prefix {tot_pref_b} ({tot_pref_b / max(tot_full_b, 1):.3f})")
print("prefix histogram (32 buckets; last bin = 11+):", hist.tolist())
