#!/usr/bin/env python3
"""Planner-level honesty line: what a PLAN costs through this build's batched GPU path.

BASELINE config 1 — the reference's scripts/sphere_cage_example.py problem (Panda, 14 spheres of radius 0.2, start and
goal restated in vamp_mvt_amd.workloads / tests/oracle_lib.py), 12 perturbed trials per planner, Halton sampler — solved
with this package's host-side planners (vamp_mvt_amd/planning.py: loops that ask their validity questions in batches).

Per planner: wall time per solved plan (median / mean), calls of the device path per plan (validate_batch,
validate_motion_batch: each one is a host -> device round trip), units per call, and next to them the reference's
published figures for its serial C++ RRT-Connect on one CPU core (README.md:22: 35 us median; scripts/README.md:9-33:
76.2 us mean on the 700 MotionBenchMaker Panda problems).  Also the latency of ONE <robot>.validate(q, env) call.

A device path answers one question in tens of microseconds and a million in a quarter of a millisecond: a planner
that asks one question at a time (RRT-Connect) is slower here than on the reference's CPU path, a planner that can ask
thousands at once (roadmaps, FCIT* batches) is where the batch API pays.  This script prints the numbers; it makes no
claim beyond them.

    python tools/bench_planners.py [--trials 12] [--json out.json]
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import random
import statistics
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import vamp_mvt_amd as vamp  # noqa: E402
from vamp_mvt_amd.workloads import SPHERE_CAGE  # noqa: E402

START = [0., -0.785, 0., -2.356, 0., 1.571, 0.785]
GOAL = [2.35, 1., 0., -0.8, 0, 2.5, 0.785]
REFERENCE = {"rrtc_median_us": 35.0, "rrtc_mean_us": 76.2, "source": "reference README.md:22, scripts/README.md:9-33 "
             "(Panda RRT-Connect, 700 MotionBenchMaker problems, AMD Ryzen 9 7950X, one core)"}


class CallCounter:
    """wraps the robot module's batched entry points: calls and units per call"""

    def __init__(self, robot):
        self.robot, self.calls, self.units = robot, {}, {}
        self.saved = {}
        for name in ("validate_batch", "validate_motion_batch"):
            fn = getattr(robot, name)
            self.saved[name] = fn
            setattr(robot, name, self._wrap(name, fn))

    def _wrap(self, name, fn):
        def counted(first, *rest, **kw):
            self.calls[name] = self.calls.get(name, 0) + 1
            self.units[name] = self.units.get(name, 0) + int(getattr(first, "shape", [len(first)])[0])
            return fn(first, *rest, **kw)
        return counted

    def reset(self):
        self.calls, self.units = {}, {}

    def restore(self):
        for name, fn in self.saved.items():
            setattr(self.robot, name, fn)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=12)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    vamp.set_device(0)
    out = {"problem": "sphere cage (BASELINE config 1), Panda, 14 spheres r = 0.2 perturbed by U(-0.01, 0.01), Halton sampler",
           "trials": args.trials, "reference": REFERENCE, "planners": {}}

    # latency of one validate / validate_motion call (what a serial planner pays per question)
    env = vamp.Environment()
    for c in SPHERE_CAGE:
        env.add_sphere(vamp.Sphere(c, 0.2))
    vamp.panda.validate(START, env)
    for name, fn in (("validate", lambda: vamp.panda.validate(START, env)),
                     ("validate_motion", lambda: vamp.panda.validate_motion(START, GOAL, env))):
        t = []
        for _ in range(300):
            t0 = time.perf_counter()
            fn()
            t.append(time.perf_counter() - t0)
        out[f"single_{name}_us"] = {"median": statistics.median(t) * 1e6, "p95": sorted(t)[int(0.95 * len(t))] * 1e6}
    print(f"one validate(q, env) call: {out['single_validate_us']['median']:.1f} us median; one validate_motion: "
          f"{out['single_validate_motion_us']['median']:.1f} us   (reference: ~1.5 us per fkcc rake on one core)", flush=True)

    counter = CallCounter(vamp.panda)
    for planner in ("rrtc", "prm", "fcit"):
        module, planner_func, plan_settings, simp_settings = vamp.configure_robot_and_planner_with_kwargs("panda", planner)
        sampler = module.halton()
        random.seed(0)
        np.random.seed(0)
        spheres = [np.array(s) for s in SPHERE_CAGE]
        rows = []
        for trial in range(args.trials + 1):  # trial 0 warms the path up (environment upload, first launches)
            random.shuffle(spheres)
            e = vamp.Environment()
            for s in copy.deepcopy(spheres):
                s += np.random.uniform(low=-0.01, high=0.01, size=(3,))
                e.add_sphere(vamp.Sphere(s, 0.2))
            if not (vamp.panda.validate(START, e) and vamp.panda.validate(GOAL, e)):
                continue
            counter.reset()
            t0 = time.perf_counter()
            result = planner_func(START, GOAL, e, plan_settings, sampler)
            wall = time.perf_counter() - t0
            if trial == 0:
                continue
            calls = sum(counter.calls.values())
            units = sum(counter.units.values())
            rows.append({"solved": bool(result.solved), "wall_us": wall * 1e6, "iterations": int(result.iterations),
                         "device_calls": calls, "units": units, "path_vertices": len(result.path),
                         "path_valid": bool(result.solved and result.path.validate(e))})
        solved = [r for r in rows if r["solved"]]
        summary = {"trials": len(rows), "solved": len(solved), "all_paths_valid": all(r["path_valid"] for r in solved)}
        if solved:
            w = [r["wall_us"] for r in solved]
            summary.update(wall_us_median=statistics.median(w), wall_us_mean=statistics.fmean(w),
                           device_calls_per_plan=statistics.fmean(r["device_calls"] for r in solved),
                           units_per_call=sum(r["units"] for r in solved) / max(1, sum(r["device_calls"] for r in solved)),
                           iterations_median=statistics.median(r["iterations"] for r in solved))
            summary["x_reference_rrtc_median"] = summary["wall_us_median"] / REFERENCE["rrtc_median_us"]
        out["planners"][planner] = summary
        print(f"{planner:5s} solved {summary['solved']}/{summary['trials']}  wall per plan: median "
              f"{summary.get('wall_us_median', float('nan')):.0f} us, mean {summary.get('wall_us_mean', float('nan')):.0f} us  "
              f"device calls per plan {summary.get('device_calls_per_plan', float('nan')):.1f} "
              f"({summary.get('units_per_call', float('nan')):.0f} units per call)   reference rrtc: "
              f"{REFERENCE['rrtc_median_us']:.0f} us median / {REFERENCE['rrtc_mean_us']:.1f} us mean", flush=True)
    counter.restore()
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
