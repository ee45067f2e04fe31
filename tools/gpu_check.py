#!/usr/bin/env python3
"""Developer check on a GPU box: HIP path vs CPU oracle on several robots/environments + a quick timing.
(Not part of the test-suite; tests/ holds the real parity tests.)"""
from __future__ import annotations

import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import vamp_mvt_amd as vamp  # noqa: E402
from envs import make_env  # noqa: E402
from oracle_lib import Oracle  # noqa: E402


def main():
    names = sys.argv[1:] or ["panda", "ur5", "fetch", "baxter"]
    o = Oracle()
    rng = np.random.default_rng(5)
    ok = True
    for name in names:
        mod = getattr(vamp, name)
        rid = o.robot(name)
        lo, span = o.bounds(rid)
        for kind in ("empty", "cage", "shell64", "mixed", "capt", "mvt", "heightfield", "attach"):
            env, oenv = make_env(kind, o, name)
            n = 4096 if kind not in ("capt", "mvt") else 2048
            q = (lo + span * rng.random((n, len(lo)), dtype=np.float32)).astype(np.float32)
            t = time.time()
            got = mod.validate_batch(q, env)
            tg = time.time() - t
            t = time.time()
            want = o.validate_batch(rid, oenv, q, threads=8)
            tc = time.time() - t
            bad = int((got != want).sum())
            ne = 512 if kind not in ("capt", "mvt") else 256
            a = q[:ne]
            b = (a + rng.normal(0, 0.3, a.shape).astype(np.float32)).astype(np.float32)
            got_e = mod.validate_motion_batch(a, b, env)
            want_e = o.validate_motion_batch(rid, oenv, a, b)
            bad_e = int((got_e != want_e).sum())
            print(f"{name:7s} {kind:8s} configs: {bad} mismatches / {n} (valid {int(want.sum())}); "
                  f"edges: {bad_e} / {ne} (valid {int(want_e.sum())})  gpu {tg:.3f}s cpu {tc:.3f}s", flush=True)
            ok = ok and bad == 0 and bad_e == 0
        fk = mod.fk_batch(q[:16])
        fbad = sum(int((fk[i].view(np.uint32) != o.fk(rid, q[i]).view(np.uint32)).sum()) for i in range(16))
        print(f"{name:7s} fk words differing: {fbad}", flush=True)
        ok = ok and fbad == 0
    print("ALL OK" if ok else "MISMATCHES FOUND")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
