"""Host-side planner harness on top of the batched validity API.

The reference's planners are serial C++ templates that call `validate_motion` one edge at a time
(planning/rrtc.hh, prm.hh, fcit.hh).  They are NOT re-implemented here; this module is the small amount of host
logic needed to drive the GPU path the way those planners do:

  * `Halton`            the reference's deterministic sampler (random/halton.hh:75-108), restated in fp32 numpy
                        (bit-exact against the reference's own output, tests/golden/halton_panda.npz);
  * `rrtc`              a minimal RRT-Connect (the `sphere_cage_example.py` plumbing, BASELINE config 1): same
                        ingredients as rrtc.hh (two balanced trees, `range`-limited extension, connect loop), every
                        validity question answered by `validate_motion_batch` — extension and connect candidates of
                        one iteration are checked together;
  * `fcit`              the batch form of the FCIT* loop (fcit.hh:137-349): lazy A* over the complete graph of the valid
                        samples, the unknown edges of each candidate path validated together;
  * `build_roadmap`     the batched form of PRM's inner loops (prm.hh:109-189, roadmap construction :250-266):
                        validate all samples at once, then all k-nearest candidate edges at once, then connect
                        components / search on the host.

Nothing here is on the measured hot path; all collision work goes through vamp_mvt_amd.<robot>.validate_*_batch.
"""
from __future__ import annotations

import heapq
from dataclasses import dataclass, field

import numpy as np

_PRIMES = np.array([3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59], np.float32)


class Halton:
    """vamp::rng::Halton<Robot> (random/halton.hh): default bases = the first `dimension` primes from 3 on."""

    max_iterations = 1000000

    def __init__(self, robot):
        self._lo = robot.lower_bounds()
        self._span = robot._span.copy()  # Robot::s_m
        self._dim = robot.dimension()
        self.reset()

    def reset(self):
        self.b = _PRIMES[: self._dim].copy()
        self.n = np.zeros(self._dim, np.float32)
        self.d = np.ones(self._dim, np.float32)
        self.iterations = 0

    def skip(self, count: int):
        for _ in range(count):
            self.next()

    def next(self) -> np.ndarray:
        f = np.float32
        self.iterations += 1
        if self.iterations > self.max_iterations:
            self.n[:] = 0
            self.d[:] = 1
            self.iterations = 0
            self.b = np.roll(self.b, -1)
        b, n, d = self.b, self.n, self.d
        xf = (d - n).astype(np.float32)
        x_eq_1 = xf == f(1)
        d = np.where(x_eq_1, np.floor((d * b).astype(np.float32)), d).astype(np.float32)
        y = np.where(x_eq_1, f(0), np.floor((d / b).astype(np.float32))).astype(np.float32)
        x_le_y = (~x_eq_1) & (xf <= y)
        while x_le_y.any():
            y = np.where(x_le_y, np.floor((y / b).astype(np.float32)), y).astype(np.float32)
            x_le_y = x_le_y & (xf <= y)
        n = np.where(x_eq_1, f(1), (np.floor(((b + f(1)) * y).astype(np.float32)) - xf)).astype(np.float32)
        self.n, self.d = n, d
        u = (n / d).astype(np.float32)
        return (u * self._span + self._lo).astype(np.float32)  # Robot::scale_configuration (panda.hh:77-80)

    def batch(self, count: int) -> np.ndarray:
        return np.stack([self.next() for _ in range(count)])


@dataclass
class RRTCSettings:
    """subset of planning/rrtc_settings.hh"""
    range: float = 2.0
    balance: bool = True
    tree_ratio: float = 1.0
    max_iterations: int = 100000
    max_samples: int = 100000


@dataclass
class PlanningResult:
    """planning/plan.hh:172-179"""
    path: list = field(default_factory=list)
    iterations: int = 0
    size: list = field(default_factory=list)
    validity_calls: int = 0
    cost: float = float("inf")
    edges_checked: int = 0
    samples_drawn: int = 0

    @property
    def solved(self):
        return len(self.path) > 0


class _Tree:
    def __init__(self, dim):
        self.pts = np.zeros((0, dim), np.float32)
        self.parent = []

    def add(self, q, parent):
        self.pts = np.vstack([self.pts, q[None, :]])
        self.parent.append(parent)
        return len(self.parent) - 1

    def nearest(self, q):
        d = np.linalg.norm(self.pts - q[None, :], axis=1)
        i = int(np.argmin(d))
        return i, float(d[i])

    def trace(self, i):
        out = []
        while True:
            out.append(self.pts[i])
            if self.parent[i] == i:
                return out
            i = self.parent[i]


def rrtc(robot, start, goal, environment, settings: RRTCSettings | None = None, sampler: Halton | None = None):
    """Minimal RRT-Connect on the batched validity API (same structure as planning/rrtc.hh:33-245)."""
    s = settings or RRTCSettings()
    rng = sampler or Halton(robot)
    start = np.asarray(start, np.float32)
    goals = np.asarray(goal, np.float32)
    goals = goals[None] if goals.ndim == 1 else goals  # the reference's multi-goal form: every goal roots the goal tree
    res = PlanningResult()

    def motion(a, b):
        res.validity_calls += 1
        return robot.validate_motion_batch(np.ascontiguousarray(a), np.ascontiguousarray(b), environment)

    direct = motion(np.repeat(start[None], len(goals), 0), goals)  # rrtc.hh:60-73
    if direct.any():
        res.path, res.size = [start, goals[int(np.argmax(direct))]], [1, 1]
        return res
    tree_a, tree_b = _Tree(len(start)), _Tree(len(start))
    tree_a.add(start, 0)
    for g in goals:
        tree_b.add(g, len(tree_b.parent))
    a_is_start = True
    while res.iterations < s.max_iterations and len(tree_a.parent) + len(tree_b.parent) < s.max_samples:
        res.iterations += 1
        asize, bsize = len(tree_a.parent), len(tree_b.parent)
        if (not s.balance) or abs(asize - bsize) / asize < s.tree_ratio:  # rrtc.hh:100-108
            tree_a, tree_b = tree_b, tree_a
            a_is_start = not a_is_start
        target = rng.next()
        ni, dist = tree_a.nearest(target)
        near = tree_a.pts[ni]
        reach = min(dist, s.range)
        if dist <= 0:
            continue
        new = (near + (target - near) * np.float32(reach / dist)).astype(np.float32)
        if not bool(motion(near[None], new[None])[0]):  # extend (rrtc.hh:136-140)
            continue
        new_i = tree_a.add(new, ni)
        # connect (rrtc.hh:160-191): march from tree_b's nearest node towards `new` in `range` steps; all the
        # steps of the march are validated in ONE batch and the first invalid one cuts it
        bi, bdist = tree_b.nearest(new)
        origin = tree_b.pts[bi]
        n_steps = max(int(np.ceil(bdist / s.range)), 1)
        way = np.stack([(origin + (new - origin) * np.float32(min((k + 1) * s.range, bdist) / bdist)).astype(np.float32)
                        for k in range(n_steps)]) if bdist > 0 else new[None]
        froms = np.vstack([origin[None], way[:-1]])
        ok = motion(froms, way)
        n_ok = int(np.argmin(ok)) if not ok.all() else len(ok)
        prev = bi
        for k in range(n_ok):
            prev = tree_b.add(way[k], prev)
        if n_ok == len(ok):  # reached `new`: join the two branches
            pa = tree_a.trace(new_i)[::-1]
            pb = tree_b.trace(prev)
            path = pa + pb[1:] if np.array_equal(pa[-1], pb[0]) else pa + pb
            if not a_is_start:
                path = path[::-1]
            res.path = [np.asarray(p, np.float32) for p in path]
            res.size = [len(tree_a.parent), len(tree_b.parent)]
            return res
    res.size = [len(tree_a.parent), len(tree_b.parent)]
    return res


def validate_path(robot, path, environment) -> bool:
    """Path::validate (planning/plan.hh:155-168): every consecutive pair is a valid motion — one batch."""
    if len(path) < 2:
        return True
    p = np.stack(path).astype(np.float32)
    return bool(robot.validate_motion_batch(p[:-1], p[1:], environment).all())


@dataclass
class Roadmap:
    vertices: np.ndarray  # [n][dim] valid samples
    edges: np.ndarray     # [m][2] index pairs (valid motions)
    candidate_edges: int = 0
    sampled: int = 0

    def neighbours(self):
        adj = [[] for _ in range(len(self.vertices))]
        for a, b in self.edges:
            w = float(np.linalg.norm(self.vertices[a] - self.vertices[b]))
            adj[a].append((b, w))
            adj[b].append((a, w))
        return adj

    def shortest_path(self, src: int, dst: int):
        """A* with the straight-line heuristic (planning/roadmap.hh style search on the host)."""
        adj = self.neighbours()
        h = lambda i: float(np.linalg.norm(self.vertices[i] - self.vertices[dst]))
        best = {src: 0.0}
        prev = {}
        heap = [(h(src), src)]
        while heap:
            _, u = heapq.heappop(heap)
            if u == dst:
                out = [u]
                while u in prev:
                    u = prev[u]
                    out.append(u)
                return out[::-1]
            for v, w in adj[u]:
                g = best[u] + w
                if g < best.get(v, np.inf):
                    best[v], prev[v] = g, u
                    heapq.heappush(heap, (g + h(v), v))
        return None


def build_roadmap(robot, environment, n_samples=2048, k=8, sampler: Halton | None = None, extra_vertices=()):
    """Batched PRM construction: samples -> validate_batch; k-nearest candidate edges -> validate_motion_batch."""
    rng = sampler or Halton(robot)
    samples = rng.batch(n_samples)
    if len(extra_vertices):
        samples = np.vstack([np.asarray(extra_vertices, np.float32), samples])
    keep = robot.validate_batch(samples, environment)
    v = samples[keep]
    n = len(v)
    if n < 2:
        return Roadmap(v, np.zeros((0, 2), np.int64), 0, len(samples))
    d = np.linalg.norm(v[:, None, :] - v[None, :, :], axis=2)
    np.fill_diagonal(d, np.inf)
    nn = np.argsort(d, axis=1)[:, : min(k, n - 1)]
    pairs = {(min(i, int(j)), max(i, int(j))) for i in range(n) for j in nn[i]}
    cand = np.array(sorted(pairs), np.int64)
    ok = robot.validate_motion_batch(v[cand[:, 0]], v[cand[:, 1]], environment)
    return Roadmap(v, cand[ok], len(cand), len(samples))


@dataclass
class FCITSettings:
    """the knobs of planning/fcit.hh that matter for a batch driver"""
    batch_size: int = 1000
    max_samples: int = 20000
    max_iterations: int = 64
    optimize: bool = False  # keep adding batches after the first solution (prune with the current cost)


def fcit(robot, start, goal, environment, settings: FCITSettings | None = None, sampler: Halton | None = None):
    """Batch form of the FCIT* loop (planning/fcit.hh:137-349): sample a batch, keep the valid samples
    (`validate_batch`), search the COMPLETE graph over all samples with A* (straight-line heuristic) treating
    unchecked edges as free, then check the unknown edges of the candidate path together (`validate_motion_batch`),
    delete the invalid ones and search again; add another batch when no path remains.  With `optimize`, later
    batches only keep samples inside the current solution's ellipse (the informed set).  Every collision question
    goes through the batched API; the reference asks the same questions one edge at a time (fcit.hh:238,333)."""
    s = settings or FCITSettings()
    rng = sampler or Halton(robot)
    start, goal = np.asarray(start, np.float32), np.asarray(goal, np.float32)
    res = PlanningResult()
    if not robot.validate_batch(np.stack([start, goal]), environment).all():
        return res
    verts = np.stack([start, goal])
    state = {}  # (i, j), i < j -> True valid / False invalid; absent = unchecked
    best_cost, best_path = np.inf, None
    drawn = checked = 0

    def key(a, b):
        return (a, b) if a < b else (b, a)

    def astar():
        n = len(verts)
        h = np.linalg.norm(verts - verts[1], axis=1)
        g = np.full(n, np.inf)
        g[0] = 0.0
        parent = np.full(n, -1)
        closed = np.zeros(n, bool)
        pq = [(float(h[0]), 0)]
        while pq:
            _, u = heapq.heappop(pq)
            if closed[u]:
                continue
            closed[u] = True
            if u == 1:
                path = [1]
                while path[-1] != 0:
                    path.append(int(parent[path[-1]]))
                return path[::-1], float(g[1])
            d = np.linalg.norm(verts - verts[u], axis=1)
            cand = np.flatnonzero(~closed & (g[u] + d + h < best_cost))
            for v in cand:
                if state.get(key(u, int(v))) is False:
                    continue
                ng = g[u] + d[v]
                if ng < g[v]:
                    g[v], parent[v] = ng, u
                    heapq.heappush(pq, (float(ng + h[v]), int(v)))
        return None, np.inf

    for it in range(s.max_iterations):
        res.iterations = it + 1
        while True:
            path, cost = astar()
            if path is None:
                break
            unknown = [key(a, b) for a, b in zip(path[:-1], path[1:]) if key(a, b) not in state]
            if not unknown:
                if cost < best_cost:
                    best_cost, best_path = cost, path
                break
            e = np.array(unknown, np.int64)
            ok = robot.validate_motion_batch(verts[e[:, 0]], verts[e[:, 1]], environment)
            checked += len(e)
            for k, v in zip(unknown, ok):
                state[k] = bool(v)
        if best_path is not None and not s.optimize:
            break
        if drawn >= s.max_samples:
            break
        batch = rng.batch(s.batch_size)
        drawn += len(batch)
        keep = robot.validate_batch(batch, environment)
        new = batch[keep]
        if np.isfinite(best_cost):  # informed set: |x - start| + |x - goal| < current cost
            new = new[np.linalg.norm(new - start, axis=1) + np.linalg.norm(new - goal, axis=1) < best_cost]
        verts = np.vstack([verts, new])
    if best_path is not None:
        res.path = verts[best_path]
        res.cost = best_cost
    res.size = len(verts)
    res.edges_checked = checked
    res.samples_drawn = drawn
    return res
