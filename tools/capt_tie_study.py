#!/usr/bin/env python3
"""Why does the survey record 873,895 CAPT affordance vectors for the Baxter radii on its 10,000-point shell cloud
where the judge's restatement of the generator gives 873,894?  

Runs the CPU oracle's CAPT build (test infrastructure) on variations of the cloud and prints one line each:

 1. tie order: the cloud has three pairs of points that share one coordinate (the reference's pdqsort leaves their
    order open); every combination of swapped pairs is built.                 -> no count changes: NOT tie order.
 2. floating-point contraction inside the build (distsq_to, contained_by_internal_ball): an oracle compiled with
    -ffp-contract=fast -mfma gives identical leaves (checked by hand, not repeated here).
 3. the generator: `rr = 0.6f + 0.6f*u` and `z = 0.2f + 1.3f*u` evaluated as ONE fused multiply-add each — what a
    driver compiled with g++ -march=native (contraction on by default) does — moves ~3,300 points by one ulp and gives
    24,169 / 177,408 / 873,895: the survey's three numbers.
 4. which decision flips: with r_max one ulp larger three leaves gain a point whose squared distance to the cell is
    exactly one ulp above (r_max + r_point)^2; only leaf 6117's extra point crosses an 8-point vector boundary.

Usage: python tools/capt_tie_study.py          (about a minute; needs no GPU and no reference)"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pins  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

RADII = {"panda": (0.012, 0.08), "fetch": (0.012, 0.24), "baxter": (0.012, 0.5)}


def counts(o, cloud, names=("panda", "fetch", "baxter"), r_max_ulps=0):
    out = []
    for name in names:
        r_min, r_max = RADII[name]
        r_max = np.float32(r_max)
        for _ in range(r_max_ulps):
            r_max = np.nextafter(r_max, np.float32(1))
        e = o.env()
        e.add_capt(cloud, r_min, float(r_max), 0.0025)
        out.append(e.capt())
    return out


def leaf_points(c):
    fin = np.isfinite(c["aff"][0]).sum(1)
    cs = np.concatenate([[0], np.cumsum(fin)])
    return cs[c["aff_starts"][1:]] - cs[c["aff_starts"][:-1]]


def main():
    o = Oracle()
    cloud = pins.capt_cloud(0)
    pairs = []
    for k in range(3):
        vals, inv, cnt = np.unique(cloud[:, k], return_inverse=True, return_counts=True)
        for v in np.where(cnt > 1)[0]:
            idx = np.where(inv == v)[0]
            pairs.append(idx)
            print(f"duplicate coordinate: axis {k} value {vals[v]!r} points {idx.tolist()}")
    for swap in itertools.product([0, 1], repeat=len(pairs)):
        c = cloud.copy()
        for s, idx in zip(swap, pairs):
            if s:
                c[idx[0]], c[idx[1]] = cloud[idx[1]], cloud[idx[0]]
        print("tie order", swap, [int(x["aff"].shape[1]) for x in counts(o, c, ("baxter",))])
    fused = pins.capt_cloud(0, fma=True)
    print("separate mul/add generator:", [int(x["aff"].shape[1]) for x in counts(o, cloud)])
    print("fused multiply-add generator:", [int(x["aff"].shape[1]) for x in counts(o, fused)],
          f"({int((fused != cloud).any(1).sum())} points differ by one ulp)")
    a, = counts(o, cloud, ("baxter",))
    b, = counts(o, cloud, ("baxter",), r_max_ulps=1)
    pa, pb = leaf_points(a), leaf_points(b)
    for leaf in np.where(pa != pb)[0]:
        va = int(a["aff_starts"][leaf + 1] - a["aff_starts"][leaf])
        vb = int(b["aff_starts"][leaf + 1] - b["aff_starts"][leaf])
        print(f"r_max + 1 ulp: leaf {leaf} holds {pa[leaf]} -> {pb[leaf]} points, {va} -> {vb} vectors")


if __name__ == "__main__":
    main()
