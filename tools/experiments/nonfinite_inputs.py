#!/usr/bin/env python3
"""What happens to non-finite joint values (NaN of either sign, +-inf)?  GPU vs oracle, every robot, the 64-primitive scene
and a CAPT cloud.  (The reference's predicates are sign-bit tests on values that are then NaN; its answer for such input is an
artefact of NaN propagation, so this is a survey, not a parity claim.)"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vamp_mvt_amd as vamp  # noqa: E402
from envs import make_env  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

vamp.set_device(0)
o = Oracle()
specials = [np.float32(np.nan), -np.float32(np.nan), np.float32(np.inf), -np.float32(np.inf)]
for name in ("panda", "ur5", "fetch", "baxter"):
    for kind in ("shell64", "capt", "empty"):
        env, oenv = make_env(kind, o, name)
        rid = o.robot(name)
        lo, span = o.bounds(rid)
        rng = np.random.default_rng(3)
        q = (lo + span * rng.random((len(lo) * 4 * 8, len(lo)), dtype=np.float32)).astype(np.float32)
        r = 0
        for j in range(len(lo)):
            for s in specials:
                for _ in range(8):
                    q[r, j] = s
                    r += 1
        got = getattr(vamp, name).validate_batch(q, env)
        want = o.validate_batch(rid, oenv, q)
        a = q[:-1:2][:64]
        b = q[1::2][:64]
        got_e = getattr(vamp, name).validate_motion_batch(a, b, env)
        want_e = o.validate_motion_batch(rid, oenv, a, b)
        print(f"{name:7s} {kind:8s} configs: {int((got != want).sum())} of {len(q)} differ (gpu valid {int(got.sum())}, oracle valid {int(want.sum())});"
              f" edges: {int((got_e != want_e).sum())} of {len(a)} differ", flush=True)
