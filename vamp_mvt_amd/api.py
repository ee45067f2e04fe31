"""The reference's planner-facing Python names on top of the batched GPU path.

`import vamp_mvt_amd as vamp` then gives what scripts/sphere_cage_example.py:45-67 uses, under the reference's names
and signatures (bindings/robot_helper.hh:326-597, bindings/settings.cc:13-117, src/vamp/__init__.py:69-139,188-228):

    vamp.configure_robot_and_planner_with_kwargs(robot, planner, **kwargs) -> (module, planner_func, settings, simp_settings)
    vamp.<robot>.halton() -> sampler with next() / skip(n) / reset()          (random/halton.hh, bit-exact)
    vamp.<robot>.rrtc / prm / fcit(start, goal | goals, environment, settings, rng) -> PlanningResult
    vamp.<robot>.roadmap(start, goal, environment, settings, rng) -> Roadmap
    vamp.<robot>.simplify(path, environment, settings, rng) -> PlanningResult
    vamp.<robot>.Path, PlanningResult(solved, path, nanoseconds, iterations, size), space_measure()
    vamp.RRTCSettings, PRMSettings, PRMNeighborParams, FCITSettings, FCITNeighborParams, SimplifySettings, ...
    vamp.results_to_dict(planning_result, simplification_result)

What is NOT the reference's: the planners themselves.  The reference's are serial C++ templates (rrtc.hh, prm.hh,
fcit.hh, simplify.hh) that ask one validity question at a time; these are host-side loops (vamp_mvt_amd/planning.py)
that ask the same kinds of questions in batches of the GPU path, so iteration counts, tree sizes and path costs differ
from the reference's (no reference test pins those).  `simplify` is a shortcutting pass only (the reference's
simplifiers are out of this build's scope): every shortcut is a `validate_motion` question answered on
the GPU.  `xorshift()` raises, as the reference's does where SIMDxorshift is unavailable (robot_helper.hh:406-409)."""
from __future__ import annotations

import logging
import math
import time
from dataclasses import dataclass, field

import numpy as np

from . import planning

DEFAULT_ITERATIONS = 1000000  # src/vamp/constants.py:1
ROBOT_RRT_RANGES = {"sphere": 1, "ur5": 1.5, "panda": 1.0, "fetch": 1.0, "baxter": 0.5}  # constants.py:3-9
# Robot::space_measure() (robots/panda.hh:111-114, ur5.hh:105-108, fetch.hh:117-120, baxter.hh:153-156): model constants
SPACE_MEASURE = {"panda": 57376.4026747593, "ur5": 61528.90796697732, "fetch": 16384.87636281249,
                 "baxter": 590532810.7756369}


# --------------------------------------------------------------------------------------------- settings (settings.cc)
@dataclass
class RRTCSettings:
    """planning/rrtc_settings.hh (dynamic_domain / radius / alpha / min_radius / start_tree_first are accepted and kept;
    the batch driver uses range, balance, tree_ratio, max_iterations, max_samples)"""
    range: float = 2.0
    dynamic_domain: bool = True
    radius: float = 4.0
    alpha: float = 0.0001
    min_radius: float = 1.0
    balance: bool = True
    tree_ratio: float = 1.0
    max_iterations: int = 100000
    max_samples: int = 100000
    start_tree_first: bool = True


def _unit_ball_measure(dim):
    return math.pow(math.sqrt(math.pi), dim) / math.gamma(dim / 2.0 + 1.0)  # planning/roadmap.hh:18-22


class PRMNeighborParams:
    """vp::PRMStarNeighborParams (planning/roadmap.hh:44-81)"""

    def __init__(self, dim, space_measure):
        self.dim, self.space_measure, self.gamma_scale = int(dim), float(space_measure), 2.0

    def max_neighbors(self, num_states):
        return int(math.ceil((math.e + math.e / self.dim) * math.log(float(max(num_states, 1)))))

    def neighbor_radius(self, num_states):
        inv = 1.0 / self.dim
        ratio = self.space_measure / _unit_ball_measure(self.dim)
        c = 2.0 * math.pow(1.0 + inv, inv) * math.pow(ratio, inv)
        n = max(num_states, 2)
        return float(np.float32(self.gamma_scale * c * math.pow(math.log(n) / n, inv)))


class FCITNeighborParams(PRMNeighborParams):
    """vp::FCITStarNeighborParams (roadmap.hh:83-111): all neighbours, infinite radius"""

    def max_neighbors(self, num_states):
        return 2 ** 63 - 1

    def neighbor_radius(self, num_states):
        return float("inf")


class PRMSettings:
    """vp::RoadmapSettings<PRMStarNeighborParams> (settings.cc:54-61)"""

    def __init__(self, neighbor_params):
        self.neighbor_params = neighbor_params
        self.max_iterations = 100000
        self.max_samples = 100000

    def max_neighbors(self, num_states):
        return self.neighbor_params.max_neighbors(num_states)

    def neighbor_radius(self, num_states):
        return self.neighbor_params.neighbor_radius(num_states)


class FCITSettings(PRMSettings):
    """vp::RoadmapSettings<FCITStarNeighborParams> (settings.cc:72-81)"""

    def __init__(self, neighbor_params):
        super().__init__(neighbor_params)
        self.batch_size = 1000
        self.optimize = False


class SimplifyRoutine:
    BSPLINE, REDUCE, SHORTCUT, PERTURB = "BSPLINE", "REDUCE", "SHORTCUT", "PERTURB"


@dataclass
class BSplineSettings:
    max_steps: int = 5
    min_change: float = 0.05
    midpoint_interpolation: float = 0.5


@dataclass
class ReduceSettings:
    max_steps: int = 25
    max_empty_steps: int = 10
    range_ratio: float = 0.33


@dataclass
class ShortcutSettings:
    pass


@dataclass
class PerturbSettings:
    max_steps: int = 25
    max_empty_steps: int = 10
    perturbation_attempts: int = 5
    range_ratio: float = 0.1


@dataclass
class SimplifySettings:
    """vp::SimplifySettings (settings.cc:109-117); this build runs the SHORTCUT routine only"""
    max_iterations: int = 4
    interpolate: int = 0
    operations: list = field(default_factory=lambda: [SimplifyRoutine.SHORTCUT, SimplifyRoutine.BSPLINE])
    reduce: ReduceSettings = field(default_factory=ReduceSettings)
    shortcut: ShortcutSettings = field(default_factory=ShortcutSettings)
    perturb: PerturbSettings = field(default_factory=PerturbSettings)
    bspline: BSplineSettings = field(default_factory=BSplineSettings)


class AORRTCSettings:
    def __init__(self):
        raise NotImplementedError("aorrtc is not part of this build (SURVEY.md §2: out of scope)")


# --------------------------------------------------------------------------------------------- per-robot objects
def _cfg(robot, q):
    a = np.ascontiguousarray(q, dtype=np.float32)
    if a.shape != (robot.dimension(),):
        raise TypeError(f"expected a configuration of {robot.dimension()} floats")
    return a


def make_path_class(robot):
    class Path(list):
        """vamp.<robot>.Path (planning/plan.hh:10-169): list of configurations"""

        def append(self, c):
            super().append(_cfg(robot, c))

        def insert(self, i, c):
            super().insert(i, _cfg(robot, c))

        def __setitem__(self, i, c):
            if isinstance(i, slice):
                super().__setitem__(i, [_cfg(robot, q) for q in c])
            else:
                super().__setitem__(i, _cfg(robot, c))

        def cost(self):
            if len(self) < 2:
                return float("inf")
            p = self.numpy()
            return float(np.float32(np.sqrt(((p[1:] - p[:-1]) ** 2).sum(1)).sum()))

        def subdivide(self):
            out = []
            for a, b in zip(self[:-1], self[1:]):
                out += [a, (a + (b - a) * np.float32(0.5)).astype(np.float32)]
            out.append(self[-1])
            self[:] = out

        def interpolate_to_resolution(self, resolution):
            if len(self) < 2:
                return
            out = []
            for a, b in zip(list(self[:-1]), list(self[1:])):
                seg = float(np.linalg.norm(b - a))
                n = int(seg * float(resolution))
                out.append(a)
                if seg < 1.0 / float(resolution):
                    continue
                out += [(a + (b - a) * np.float32(i / n)).astype(np.float32) for i in range(1, n)]
            out.append(self[-1])
            self[:] = out

        def interpolate_to_n_states(self, n):
            if len(self) < 2 or n < len(self):
                return
            p = self.numpy()
            seg = np.sqrt(((p[1:] - p[:-1]) ** 2).sum(1))
            if seg.sum() < np.finfo(np.float32).eps:
                return
            t = np.concatenate([[0.0], np.cumsum(seg)]) / seg.sum()
            s = np.linspace(0.0, 1.0, n)
            out = np.stack([np.interp(s, t, p[:, j]) for j in range(p.shape[1])], 1).astype(np.float32)
            self[:] = list(out)

        def validate(self, environment):
            """every consecutive pair is a valid motion (plan.hh:155-168): one batch on the GPU"""
            return planning.validate_path(robot, list(self), environment)

        def numpy(self):
            return np.stack(self).astype(np.float32) if len(self) else np.zeros((0, robot.dimension()), np.float32)

    Path.__qualname__ = f"{robot._name}.Path"
    return Path


class PlanningResult:
    """vamp.<robot>.PlanningResult (robot_helper.hh:486-498, planning/plan.hh:171-179)"""

    def __init__(self, path=None, nanoseconds=0, iterations=0, size=None, cost=0.0):
        self.path = path if path is not None else []
        self.nanoseconds, self.iterations, self.size, self.cost = int(nanoseconds), int(iterations), list(size or []), cost

    @property
    def solved(self):
        return len(self.path) >= 2


class Roadmap:
    """vamp.<robot>.Roadmap (robot_helper.hh:500-517)"""

    def __init__(self, vertices, edges, nanoseconds, iterations):
        self.vertices, self.edges, self.nanoseconds, self.iterations = vertices, edges, nanoseconds, iterations

    def __len__(self):
        return len(self.vertices)

    def __getitem__(self, i):
        return self.vertices[i]


def _goals(robot, goal):
    g = np.asarray(goal, np.float32)
    return [_cfg(robot, g)] if g.ndim == 1 else [_cfg(robot, x) for x in g]


def install(robot):
    """adds the reference's per-robot planner names to a vamp_mvt_amd robot module"""
    Path = make_path_class(robot)
    robot.Path, robot.PlanningResult, robot.Roadmap = Path, PlanningResult, Roadmap
    robot.space_measure = lambda: float(np.float32(SPACE_MEASURE[robot._name]))
    robot.halton = lambda: planning.Halton(robot)

    def xorshift():
        raise RuntimeError("XORShift is not supported by this build (the reference needs SIMDxorshift for it)")

    robot.xorshift = xorshift

    def as_result(res, t0):
        path = Path()
        for q in res.path:
            path.append(q)
        size = res.size if isinstance(res.size, (list, tuple)) else [int(res.size)]
        return PlanningResult(path, time.perf_counter_ns() - t0, res.iterations, size,
                              path.cost() if len(path) >= 2 else 0.0)

    def rrtc(start, goal, environment, settings, rng):
        t0 = time.perf_counter_ns()
        s = planning.RRTCSettings(range=float(settings.range), balance=bool(settings.balance),
                                  tree_ratio=float(settings.tree_ratio), max_iterations=int(settings.max_iterations),
                                  max_samples=int(settings.max_samples))
        return as_result(planning.rrtc(robot, _cfg(robot, start), _goals(robot, goal), environment, s, rng), t0)

    def fcit(start, goal, environment, settings, rng):
        t0 = time.perf_counter_ns()
        s = planning.FCITSettings(batch_size=int(settings.batch_size), max_samples=int(settings.max_samples),
                                  max_iterations=min(int(settings.max_iterations), 4096), optimize=bool(settings.optimize))
        best = None
        for g in _goals(robot, goal):  # the reference's multi-goal form searches towards every goal
            r = planning.fcit(robot, _cfg(robot, start), g, environment, s, rng)
            if r.solved and (best is None or r.cost < best.cost):
                best = r
        return as_result(best if best is not None else planning.PlanningResult(), t0)

    def roadmap(start, goal, environment, settings, rng, n_samples=2048):
        t0 = time.perf_counter_ns()
        goals = _goals(robot, goal)
        n = int(min(settings.max_samples, n_samples))
        k = max(1, min(settings.max_neighbors(n), 16))
        rm = planning.build_roadmap(robot, environment, n_samples=n, k=k, sampler=rng,
                                    extra_vertices=[_cfg(robot, start)] + goals)
        return Roadmap(list(rm.vertices), [tuple(map(int, e)) for e in rm.edges], time.perf_counter_ns() - t0, 1), rm

    def prm(start, goal, environment, settings, rng):
        t0 = time.perf_counter_ns()
        start_c, goals = _cfg(robot, start), _goals(robot, goal)
        ok = robot.validate_batch(np.stack([start_c] + goals), environment)
        res = planning.PlanningResult()
        if ok[0] and ok[1:].any():
            n, it = 512, 0
            while n <= min(int(settings.max_samples), 65536) and not res.solved:
                it += 1  # each attempt draws the sampler's NEXT n samples (the caller's sampler is never rewound)
                _, rm = roadmap(start_c, goals, environment, settings, rng, n_samples=n)
                res.iterations, res.size = it, [len(rm.vertices)]
                # the extra vertices survive validation in order: start first, then the valid goals
                n_goal = int(ok[1:].sum())
                best = None
                for gi in range(1, 1 + n_goal):
                    idx = rm.shortest_path(0, gi)
                    if idx is not None:
                        cost = sum(float(np.linalg.norm(rm.vertices[a] - rm.vertices[b])) for a, b in zip(idx[:-1], idx[1:]))
                        if best is None or cost < best[0]:
                            best = (cost, idx)
                if best is not None:
                    res.path = [rm.vertices[i] for i in best[1]]
                n *= 2
        return as_result(res, t0)

    def simplify(path, environment, settings, rng):
        """greedy shortcutting: for every waypoint the farthest later waypoint it can reach by a valid motion; the
        candidate motions of one pass are validated together (`validate_motion_batch`)"""
        t0 = time.perf_counter_ns()
        pts = [np.asarray(p, np.float32) for p in path]
        for _ in range(max(1, int(settings.max_iterations))):
            if len(pts) < 3:
                break
            n = len(pts)
            pairs = [(i, j) for i in range(n - 2) for j in range(i + 2, n)]
            a = np.stack([pts[i] for i, _ in pairs])
            b = np.stack([pts[j] for _, j in pairs])
            ok = robot.validate_motion_batch(a, b, environment)
            reach = {i: i + 1 for i in range(n - 1)}
            for (i, j), v in zip(pairs, ok):
                if v and j > reach[i]:
                    reach[i] = j
            out, i = [pts[0]], 0
            while i < n - 1:
                i = reach[i]
                out.append(pts[i])
            if len(out) == len(pts):
                break
            pts = out
        res = Path()
        for q in pts:
            res.append(q)
        if int(getattr(settings, "interpolate", 0)) > 0:
            res.interpolate_to_n_states(int(settings.interpolate))
        return PlanningResult(res, time.perf_counter_ns() - t0, 0, [], res.cost() if len(res) >= 2 else 0.0)

    robot.rrtc, robot.fcit, robot.prm, robot.simplify = rrtc, fcit, prm, simplify
    robot.roadmap = lambda start, goal, environment, settings, rng: roadmap(start, goal, environment, settings, rng)[0]


# ------------------------------------------------------------------- module-level helpers (contract: src/vamp/__init__.py)
_log = logging.getLogger("vamp_mvt_amd")


def _planner_settings(robot_module, robot_name, planner_name):
    """default settings object of a planner (the reference's per-planner defaults, src/vamp/__init__.py:76-95)"""
    makers = {
        "rrtc": lambda: RRTCSettings(range=ROBOT_RRT_RANGES.get(robot_name, RRTCSettings.range)),
        "prm": lambda: PRMSettings(PRMNeighborParams(robot_module.dimension(), robot_module.space_measure())),
        "fcit": lambda: FCITSettings(FCITNeighborParams(robot_module.dimension(), robot_module.space_measure())),
    }
    if planner_name not in makers:
        raise NotImplementedError(f"no automatic settings for planner '{planner_name}' in this build")
    settings = makers[planner_name]()
    settings.max_iterations = settings.max_samples = DEFAULT_ITERATIONS
    return settings


def _route_kwarg(key, plan_settings, simp_settings):
    """Where a keyword of configure_robot_and_planner_with_kwargs lands: yields (owner, attribute, label).  The rules
    are the reference's (src/vamp/__init__.py:100-134): a planner attribute by its own name; a simplifier attribute
    behind the marker `simplification_`; an attribute of one simplifier routine behind `<routine>_` (markers are
    matched anywhere in the key and removed from it)."""
    yield plan_settings, key, "planner"
    if "simplification_" in key:
        yield simp_settings, key.replace("simplification_", ""), "simplifier"
    for routine in ("reduce", "shortcut", "bspline", "perturb"):
        if routine in key:
            yield getattr(simp_settings, routine), key.replace(routine + "_", ""), f"simplifier.{routine}"


def configure_robot_and_planner_with_kwargs(robots, robot_name: str, planner_name: str, **kwargs):
    """-> (robot module, planner function, planner settings, simplifier settings); same defaults, keyword routing and
    errors as the reference's function of this name (src/vamp/__init__.py:69-139)."""
    robot_module = robots[robot_name]
    if not hasattr(robot_module, planner_name):
        raise ValueError(f"Robot {robot_name} does not support planner {planner_name}!")
    plan_settings = _planner_settings(robot_module, robot_name, planner_name)
    simp_settings = SimplifySettings()
    for key, value in kwargs.items():
        for owner, attribute, label in _route_kwarg(key, plan_settings, simp_settings):
            if not hasattr(owner, attribute):
                continue
            if owner is simp_settings and attribute == "operations":  # routine names -> SimplifyRoutine members
                value = [getattr(SimplifyRoutine, name) for name in value]
            _log.info("%s.%s = %r", label, attribute, value)
            setattr(owner, attribute, value)
    return robot_module, getattr(robot_module, planner_name), plan_settings, simp_settings


def results_to_dict(planning_result, simplification_result=None):
    """One flat record per solved problem, with the column names of the reference's function of this name
    (src/vamp/__init__.py:188-228) so that its evaluation scripts' DataFrames keep their columns."""
    try:
        from pandas import Timedelta
    except ImportError as exc:
        raise RuntimeError("pandas is not installed!") from exc
    plan_path = planning_result.path
    final = simplification_result if simplification_result else None
    final_path = final.path if final is not None else plan_path
    t_plan = Timedelta(nanoseconds=planning_result.nanoseconds)
    t_simp = Timedelta(nanoseconds=final.nanoseconds if final is not None else 0)
    return {
        "planning_time": t_plan,
        "planning_iterations": planning_result.iterations,
        "solved": len(plan_path) > 0,
        "planning_graph_size": sum(planning_result.size),
        "initial_path_vertices": len(plan_path),
        "initial_path_cost": plan_path.cost(),
        "simplification_time": t_simp,
        "simplified_path_vertices": len(final_path),
        "simplified_path_cost": final_path.cost(),
        "total_time": t_plan + t_simp,
    }
