#!/usr/bin/env python3
"""URDF/SRDF -> straight-line sphere-FK program (the build's own "robot tracer").

The reference ships pre-generated per-robot headers (robots/*.hh, produced by
the external `cricket` tracer, reference README.md:189-194).  This tool owns the
same job for this build: it reads a *spherized* URDF (kinematic tree + collision
spheres) and the SRDF (disabled collision pairs), walks the kinematic chain with
a symbolic scalar, and emits a robot model (JSON) holding

  * the joint list / bounds / resolution,
  * the fine collision spheres and one bounding sphere per link,
  * a straight-line fp32 op tape that computes every sphere centre,
  * the environment check groups and the self-collision pair groups.

tools/gen_code.py turns that JSON into HIP device code and into the C oracle.

Arithmetic contract.  Bit-exact collision booleans need the same floating-point
expression DAG as the reference's generated code, so the symbolic scalar here
follows the documented behaviour of the tool chain the reference was generated
with (Pinocchio forward kinematics recorded on a CppAD tape and printed by
CppADCodeGen), restated from their published algorithms:

  * URDF rpy -> quaternion (urdfdom `Rotation::setFromRPY`) -> 3x3 matrix
    (Eigen `Quaternion::toRotationMatrix`), all in double;
  * fixed joints are folded into constant placements (double SE3 products);
  * joint placement * joint motion uses the per-axis column forms
    (`cos*col_a + sin*col_b`, `-sin*col_a + cos*col_b`);
  * world placement: R = Rp * Rl (sum over k = 0,1,2), t = tp + Rp * tl;
    sphere centre: R * p + t;
  * tape simplifications: c∘c folds in double, 0*x = 0, 1*x = x, x+0 = x,
    parameters are recorded first in commutative ops (c + x, c * x);
  * printing: sub-expressions used once are inlined *without parentheses for
    nested sums/products*, so C++ evaluates the flattened chain left to right;
    nodes used more than once are temporaries; constants print with 15
    significant digits and are narrowed to fp32 where they meet a vector.

tools/check_trace.py compares the result with the reference (in-container only).
"""
from __future__ import annotations

import json
import math
import os
import sys
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

# ----------------------------------------------------------------------------
# constant placements (double)
# ----------------------------------------------------------------------------


def rpy_to_matrix(roll: float, pitch: float, yaw: float) -> np.ndarray:
    """urdfdom setFromRPY -> normalised quaternion -> Eigen toRotationMatrix."""
    phi, the, psi = roll / 2.0, pitch / 2.0, yaw / 2.0
    x = math.sin(phi) * math.cos(the) * math.cos(psi) - math.cos(phi) * math.sin(the) * math.sin(psi)
    y = math.cos(phi) * math.sin(the) * math.cos(psi) + math.sin(phi) * math.cos(the) * math.sin(psi)
    z = math.cos(phi) * math.cos(the) * math.sin(psi) - math.sin(phi) * math.sin(the) * math.cos(psi)
    w = math.cos(phi) * math.cos(the) * math.cos(psi) + math.sin(phi) * math.sin(the) * math.sin(psi)
    s = math.sqrt(x * x + y * y + z * z + w * w)
    if s == 0.0:
        x, y, z, w = 0.0, 0.0, 0.0, 1.0
    else:
        x, y, z, w = x / s, y / s, z / s, w / s
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([
        [1.0 - (tyy + tzz), txy - twz, txz + twy],
        [txy + twz, 1.0 - (txx + tzz), tyz - twx],
        [txz - twy, tyz + twx, 1.0 - (txx + tyy)],
    ], dtype=np.float64)


@dataclass
class SE3:
    R: np.ndarray
    t: np.ndarray

    @staticmethod
    def identity():
        return SE3(np.eye(3), np.zeros(3))

    def __mul__(self, o: "SE3") -> "SE3":
        return SE3(_matmul3(self.R, o.R), self.t + _matvec3(self.R, o.t))

    def act(self, p: np.ndarray) -> np.ndarray:
        return _matvec3(self.R, p) + self.t


def _matmul3(a, b):
    out = np.zeros((3, 3))
    for i in range(3):
        for j in range(3):
            out[i, j] = (a[i, 0] * b[0, j] + a[i, 1] * b[1, j]) + a[i, 2] * b[2, j]
    return out


def _matvec3(a, v):
    return np.array([(a[i, 0] * v[0] + a[i, 1] * v[1]) + a[i, 2] * v[2] for i in range(3)])


# ----------------------------------------------------------------------------
# symbolic scalar (tape)
# ----------------------------------------------------------------------------


class Tape:
    def __init__(self):
        self.nodes = []  # tuples: (op, a, b); operands are ints (node ids) or ('c', float)

    def new(self, op, a=None, b=None):
        self.nodes.append((op, a, b))
        return len(self.nodes) - 1


@dataclass(frozen=True)
class S:
    """Symbolic scalar: a parameter (double) or a tape variable."""
    tape: Tape = field(compare=False, repr=False)
    c: float | None = None
    n: int | None = None

    @property
    def is_param(self):
        return self.n is None

    def _arg(self):
        return ("c", self.c) if self.is_param else self.n

    def __mul__(self, o):
        o = self._lift(o)
        if self.is_param and o.is_param:
            return S(self.tape, c=self.c * o.c)
        if self.is_param or o.is_param:
            p, v = (self, o) if self.is_param else (o, self)
            if p.c == 0.0:
                return S(self.tape, c=0.0)
            if p.c == 1.0:
                return v
            return S(self.tape, n=self.tape.new("mul", p._arg(), v._arg()))
        return S(self.tape, n=self.tape.new("mul", self._arg(), o._arg()))

    __rmul__ = lambda self, o: self._lift(o) * self

    def __add__(self, o):
        o = self._lift(o)
        if self.is_param and o.is_param:
            return S(self.tape, c=self.c + o.c)
        if self.is_param or o.is_param:
            p, v = (self, o) if self.is_param else (o, self)
            if p.c == 0.0:
                return v
            return S(self.tape, n=self.tape.new("add", p._arg(), v._arg()))
        return S(self.tape, n=self.tape.new("add", self._arg(), o._arg()))

    __radd__ = lambda self, o: self._lift(o) + self

    def __sub__(self, o):
        o = self._lift(o)
        if self.is_param and o.is_param:
            return S(self.tape, c=self.c - o.c)
        if o.is_param:
            if o.c == 0.0:
                return self
            return S(self.tape, n=self.tape.new("sub", self._arg(), o._arg()))
        if self.is_param and self.c == 0.0:
            return -o
        return S(self.tape, n=self.tape.new("sub", self._arg(), o._arg()))

    def __neg__(self):
        if self.is_param:
            return S(self.tape, c=-self.c)
        return S(self.tape, n=self.tape.new("neg", self._arg()))

    def _lift(self, o):
        if isinstance(o, S):
            return o
        return S(self.tape, c=float(o))


# ----------------------------------------------------------------------------
# URDF model
# ----------------------------------------------------------------------------


def _floats(s, n=3, default=0.0):
    if s is None:
        return [default] * n
    return [float(t) for t in s.split()]


@dataclass
class Joint:
    name: str
    type: str
    parent: str
    child: str
    origin: SE3
    axis: list
    lower: float
    upper: float


@dataclass
class Link:
    name: str
    spheres: list  # [(xyz double[3], radius double)]


def parse_urdf(path):
    root = ET.parse(path).getroot()
    links, joints = {}, []
    order = []
    for l in root.findall("link"):
        spheres = []
        for c in l.findall("collision"):
            g = c.find("geometry")
            sp = g.find("sphere") if g is not None else None
            if sp is None:
                continue
            o = c.find("origin")
            xyz = _floats(o.get("xyz") if o is not None else None)
            rpy = _floats(o.get("rpy") if o is not None else None)
            spheres.append((SE3(rpy_to_matrix(*rpy), np.array(xyz)), float(sp.get("radius"))))
        links[l.get("name")] = Link(l.get("name"), spheres)
        order.append(l.get("name"))
    for j in root.findall("joint"):
        o = j.find("origin")
        xyz = _floats(o.get("xyz") if o is not None else None)
        rpy = _floats(o.get("rpy") if o is not None else None)
        ax = j.find("axis")
        axis = _floats(ax.get("xyz")) if ax is not None else [1.0, 0.0, 0.0]
        lim = j.find("limit")
        lower = float(lim.get("lower", 0.0)) if lim is not None else 0.0
        upper = float(lim.get("upper", 0.0)) if lim is not None else 0.0
        joints.append(Joint(j.get("name"), j.get("type"), j.find("parent").get("link"), j.find("child").get("link"),
                            SE3(rpy_to_matrix(*rpy), np.array(xyz)), axis, lower, upper))
    return links, order, joints


def parse_srdf_disabled(path):
    root = ET.parse(path).getroot()
    dis = set()
    for d in root.findall("disable_collisions"):
        dis.add(frozenset((d.get("link1"), d.get("link2"))))
    return dis


# ----------------------------------------------------------------------------
# bounding spheres: smallest ball enclosing a set of balls (exact, small n)
# ----------------------------------------------------------------------------


def _ball_contains(c, r, balls, eps=1e-9):
    return all(np.linalg.norm(c - p) + q <= r + eps for p, q in balls)


def _ball_from_support(sup):
    """Smallest ball tangent-enclosing all balls in `sup` (1..4 balls), or None."""
    if len(sup) == 1:
        return sup[0][0].copy(), sup[0][1]
    if len(sup) == 2:
        (p0, r0), (p1, r1) = sup
        d = np.linalg.norm(p1 - p0)
        if d + r1 <= r0:
            return p0.copy(), r0
        if d + r0 <= r1:
            return p1.copy(), r1
        R = (d + r0 + r1) / 2.0
        c = p0 + (p1 - p0) * ((R - r0) / d)
        return c, R
    # |c - p_i| = R - r_i  for all i.  Subtract first equation -> linear in (c, R).
    p0, r0 = sup[0]
    A, b = [], []
    for p, r in sup[1:]:
        A.append(np.concatenate([2.0 * (p - p0), [2.0 * (r0 - r)]]))
        b.append(p @ p - p0 @ p0 - r * r + r0 * r0)
    A, b = np.array(A), np.array(b)
    # solution set: x = x_p + s * null (if len(sup)==3 in 3D: restrict centre to the plane of the centres)
    if len(sup) == 3:
        n = np.cross(sup[1][0] - p0, sup[2][0] - p0)
        if np.linalg.norm(n) < 1e-14:
            return None
        A = np.vstack([A, np.concatenate([n, [0.0]])])
        b = np.concatenate([b, [n @ p0]])
    # now A is (3x4): one free parameter -> quadratic in R
    try:
        # express c = u + v R
        M = A[:, :3]
        u = np.linalg.solve(M, b)
        v = np.linalg.solve(M, -A[:, 3])
    except np.linalg.LinAlgError:
        return None
    # |u + v R - p0|^2 = (R - r0)^2
    w = u - p0
    qa = v @ v - 1.0
    qb = 2.0 * (w @ v) + 2.0 * r0
    qc = w @ w - r0 * r0
    roots = []
    if abs(qa) < 1e-14:
        if abs(qb) > 1e-14:
            roots = [-qc / qb]
    else:
        disc = qb * qb - 4 * qa * qc
        if disc < 0:
            return None
        sq = math.sqrt(disc)
        roots = [(-qb - sq) / (2 * qa), (-qb + sq) / (2 * qa)]
    best = None
    for R in roots:
        if R < max(r for _, r in sup) - 1e-12:
            continue
        c = u + v * R
        if best is None or R < best[1]:
            best = (c, R)
    return best


def min_enclosing_ball(balls):
    """Exact smallest enclosing ball of balls by support-set enumeration (n <= ~30)."""
    from itertools import combinations
    best = None
    n = len(balls)
    for k in (1, 2, 3, 4):
        for idx in combinations(range(n), k):
            res = _ball_from_support([balls[i] for i in idx])
            if res is None:
                continue
            c, R = res
            if best is not None and R >= best[1]:
                continue
            if _ball_contains(c, R, balls):
                best = (c, R)
    return best


# ----------------------------------------------------------------------------
# tracing
# ----------------------------------------------------------------------------

_AXES = {(1.0, 0.0, 0.0): 0, (0.0, 1.0, 0.0): 1, (0.0, 0.0, 1.0): 2}


class Traced:
    pass


def trace(urdf, srdf, joint_names, end_effector=None, bounding=None, resolution=32, name=None,
          bounding_residue=None, extra_disabled=()):
    links, link_order, joints = parse_urdf(urdf)
    disabled = parse_srdf_disabled(srdf) if srdf else set()
    extra_disabled = {frozenset(p) for p in extra_disabled}
    child_joints = {}
    parent_joint = {}
    for j in joints:
        child_joints.setdefault(j.parent, []).append(j)
        parent_joint[j.child] = j
    roots = [l for l in link_order if l not in parent_joint]
    assert len(roots) == 1, roots
    root = roots[0]

    tape = Tape()
    P = lambda c: S(tape, c=float(c))
    q_index = {n: i for i, n in enumerate(joint_names)}

    # joint frames: id 0 = universe
    oM = {0: None}  # joint id -> (R 3x3 of S, t 3 of S) ; None = identity/universe
    link_frame = {}  # link -> (joint id, SE3 placement in that joint's frame)
    joint_ids = {}
    joint_parent = {}
    lower, upper = [0.0] * len(joint_names), [0.0] * len(joint_names)

    def const_R(R):
        return [[P(R[i, j]) for j in range(3)] for i in range(3)]

    dfs_order = []

    # "full" joint numbering: every non-fixed URDF joint counts, listed or not (0 = universe); only used to decide
    # which link pairs can never be tested against each other
    full_joint = {}  # link -> full joint id
    full_parent = {0: None}

    def visit(link, jid, placement, fj=0):
        link_frame[link] = (jid, placement)
        full_joint[link] = fj
        dfs_order.append(link)
        # urdfdom keeps joints in a name-sorted map; the parser's depth-first walk follows that order
        for j in sorted(child_joints.get(link, []), key=lambda jj: jj.name):
            jp = placement * j.origin  # placement of the joint frame in parent joint frame (double)
            if j.type == "fixed" or j.name not in q_index:
                if j.type != "fixed":
                    # unlisted movable joint: frozen at q = 0 (acts like a fixed joint)
                    pass
                cfj = fj
                if j.type != "fixed":
                    cfj = len(full_parent)
                    full_parent[cfj] = fj
                visit(j.child, jid, jp, cfj)
                continue
            new_id = len(oM)
            joint_ids[j.name] = new_id
            joint_parent[new_id] = jid
            qi = q_index[j.name]
            lower[qi], upper[qi] = j.lower, j.upper
            x = S(tape, n=tape.new("in", qi))
            Rp = const_R(jp.R)
            tp = [P(v) for v in jp.t]
            ax = _AXES.get(tuple(float(a) for a in j.axis))
            if j.type in ("revolute", "continuous"):
                if ax is None:
                    raise NotImplementedError(f"unaligned revolute axis {j.axis} on {j.name}")
                sn = S(tape, n=tape.new("sin", x.n))
                cs = S(tape, n=tape.new("cos", x.n))
                col = lambda k: [Rp[i][k] for i in range(3)]
                lin = lambda a, va, b, vb: [a * va[i] + b * vb[i] for i in range(3)]
                if ax == 2:
                    c0 = lin(cs, col(0), sn, col(1))
                    c1 = lin(-sn, col(0), cs, col(1))
                    c2 = col(2)
                elif ax == 0:
                    c0 = col(0)
                    c1 = lin(cs, col(1), sn, col(2))
                    c2 = lin(-sn, col(1), cs, col(2))
                else:
                    c0 = lin(cs, col(0), -sn, col(2))
                    c1 = col(1)
                    c2 = lin(sn, col(0), cs, col(2))
                Rl = [[c0[i], c1[i], c2[i]] for i in range(3)]
                tl = tp
            elif j.type == "prismatic":
                if ax is None:
                    raise NotImplementedError(f"unaligned prismatic axis {j.axis} on {j.name}")
                Rl = Rp
                tl = [tp[i] + Rp[i][ax] * x for i in range(3)]
            else:
                raise NotImplementedError(j.type)
            if oM[jid] is None:
                Rw, tw = Rl, tl
            else:
                Rpar, tpar = oM[jid]
                Rw = [[(Rpar[i][0] * Rl[0][k] + Rpar[i][1] * Rl[1][k]) + Rpar[i][2] * Rl[2][k] for k in range(3)]
                      for i in range(3)]
                tw = [tpar[i] + ((Rpar[i][0] * tl[0] + Rpar[i][1] * tl[1]) + Rpar[i][2] * tl[2]) for i in range(3)]
            oM[new_id] = (Rw, tw)
            cfj = len(full_parent)
            full_parent[cfj] = fj
            visit(j.child, new_id, SE3.identity(), cfj)

    visit(root, 0, SE3.identity())

    def world(jid, p):
        pc = [P(v) for v in p]
        if oM[jid] is None:
            return pc
        R, t = oM[jid]
        return [((R[i][0] * pc[0] + R[i][1] * pc[1]) + R[i][2] * pc[2]) + t[i] for i in range(3)]

    # fine spheres, URDF link order
    spheres = []  # dict(link, r(float32), expr[3], local)
    link_spheres = {}
    link_order = dfs_order
    for ln in link_order:
        if not links[ln].spheres:
            continue
        jid, fr = link_frame[ln]
        for (org, rad) in links[ln].spheres:
            p_link = org.t
            p_joint = fr.act(p_link)
            spheres.append(dict(link=ln, r=float(np.float32(rad)), expr=world(jid, p_joint), local=p_link, rad=rad))
            link_spheres.setdefault(ln, []).append(len(spheres) - 1)
    n_fine = len(spheres)
    sphere_links = [ln for ln in link_order if ln in link_spheres]

    # bounding spheres, one per link with spheres, in link order
    bound_index = {}
    for ln in sphere_links:
        jid, fr = link_frame[ln]
        if bounding and ln in bounding:
            cx, cy, cz, br = bounding[ln]
            c = np.array([cx, cy, cz], np.float64)
            r32 = np.float32(br)
        else:
            balls = [(np.array(spheres[i]["local"], np.float64), float(np.float32(spheres[i]["rad"])))
                     for i in link_spheres[ln]]
            c, r = min_enclosing_ball(balls)
            r32 = np.float32(r)
        # every fine sphere lies inside the bounding sphere (the kernels' candidate pruning relies on it with a 1e-4 m
        # margin, vmv_device.h kCandidateMargin): enforce it here, in the link frame, to 2e-6 m
        for i_f in link_spheres[ln]:
            gap = np.linalg.norm(np.array(spheres[i_f]["local"], np.float64) - c) + float(np.float32(spheres[i_f]["rad"])) \
                - float(r32)
            assert gap <= 2e-6, (ln, i_f, gap)
        # the bounding centre is kept as fp32 in the frame of the link's movable joint
        p_joint = fr.act(c).astype(np.float32).astype(np.float64)
        p_joint[np.abs(p_joint) < 1e-12] = 0.0  # round-off of this solver / of the fixed-frame product is not geometry
        for axis, val in (bounding_residue or {}).get(ln, {}).items():
            p_joint[int(axis)] = val
        spheres.append(dict(link=ln, r=float(r32), expr=world(jid, p_joint), local=c, bounding=True,
                            joint_frame=[float(v) for v in p_joint], static=(oM[jid] is None)))
        bound_index[ln] = len(spheres) - 1

    # groups
    env_groups = [dict(link=ln, bound=bound_index[ln], fine=link_spheres[ln]) for ln in reversed(sphere_links)]
    self_groups = []
    adjacent = set()
    for i, a in enumerate(sphere_links):
        for b in sphere_links[i + 1:]:
            if frozenset((a, b)) in disabled:
                continue
            ja, jb = full_joint[a], full_joint[b]
            if ja == jb:
                continue  # rigidly attached to each other
            if frozenset((a, b)) in extra_disabled:
                continue
            self_groups.append(dict(a=a, b=b, bound_a=bound_index[a], bound_b=bound_index[b],
                                    pairs=[[s, t] for s in link_spheres[a] for t in link_spheres[b]]))

    # end-effector frame (Robot::fkcc_attach / eefk: y[n_total*4 ..] = translation, then the rotation column-major):
    # world frame of the movable joint composed with the frame's constant placement, as joint frames compose
    ee_expr = None
    if end_effector is not None:
        jid, fr = link_frame[end_effector]
        Rl, tl = const_R(fr.R), [P(v) for v in fr.t]
        if oM[jid] is None:
            Rw, tw = Rl, tl
        else:
            Rpar, tpar = oM[jid]
            Rw = [[(Rpar[i][0] * Rl[0][k] + Rpar[i][1] * Rl[1][k]) + Rpar[i][2] * Rl[2][k] for k in range(3)]
                  for i in range(3)]
            tw = [tpar[i] + ((Rpar[i][0] * tl[0] + Rpar[i][1] * tl[1]) + Rpar[i][2] * tl[2]) for i in range(3)]
        ee_expr = [tw[0], tw[1], tw[2]] + [Rw[i][k] for k in range(3) for i in range(3)]

    out = Traced()
    out.ee_expr = ee_expr
    out.tape, out.spheres, out.n_fine = tape, spheres, n_fine
    out.env_groups, out.self_groups = env_groups, self_groups
    out.lower, out.upper = lower, upper
    out.link_order = sphere_links
    out.joint_names = joint_names
    out.name = name
    out.resolution = resolution
    out.end_effector = end_effector
    return out


# ----------------------------------------------------------------------------
# lowering: flatten single-use nested sums/products, fold constants, emit SSA
# ----------------------------------------------------------------------------


def _c15(c: float) -> float:
    """Constant as the generated source carries it: 15 significant digits."""
    return float("%.15g" % c)


def lower(tr: Traced, ee_only: bool = False):
    """Return (ops, outputs): ops = SSA list over fp32 values; outputs[s] = [x,y,z] each ('op',id)|('const',f32).
    ee_only: the program of the end-effector frame alone (12 outputs in one list), lowered with the use counts of the
    tape that holds the spheres AND the frame (the reference's fkcc_attach), emitting only what the frame needs."""
    nodes = tr.tape.nodes
    # reachability + use counts over the pruned graph
    use = [0] * len(nodes)
    seen = [False] * len(nodes)
    roots = []
    for sp in tr.spheres:
        for e in sp["expr"]:
            if not e.is_param:
                roots.append(e.n)
    if ee_only:
        roots += [e.n for e in tr.ee_expr if not e.is_param]
    stack = []
    for r in roots:
        use[r] += 1
        if not seen[r]:
            seen[r] = True
            stack.append(r)
    while stack:
        n = stack.pop()
        op, a, b = nodes[n]
        for arg in (a, b):
            if isinstance(arg, int) and op != "in":
                use[arg] += 1
                if not seen[arg]:
                    seen[arg] = True
                    stack.append(arg)
    is_root = set(roots)

    ops = []  # (op, a, b) with a/b = int id or float const (already fp32-narrowed python float)
    memo = {}

    def f32(c):
        return float(np.float32(_c15(c)))

    def emit(op, a=None, b=None):
        ops.append((op, a, b))
        return len(ops) - 1

    def inline_ok(arg):
        return isinstance(arg, int) and use[arg] == 1 and arg not in is_root

    def flat(n, kind):
        """Flatten node n of op `kind` ('add'|'mul') into its printed operand list."""
        op, a, b = nodes[n]
        out = []
        for arg in (a, b):
            if inline_ok(arg) and nodes[arg][0] == kind:
                out.extend(flat(arg, kind))
            else:
                out.append(arg)
        return out

    def value(arg):
        """-> ('c', double) | ('v', ssa id)"""
        if not isinstance(arg, int):
            return ("c", _c15(arg[1]))
        if arg in memo:
            return ("v", memo[arg])
        op, a, b = nodes[arg]
        if op == "in":
            r = emit("in", a)
        elif op in ("sin", "cos"):
            r = emit(op, value(a)[1])
        elif op == "neg":
            r = emit("neg", value(a)[1])
        elif op == "sub":
            va, vb = value(a), value(b)
            ea = va[1] if va[0] == "v" else emit("const", f32(va[1]))
            eb = vb[1] if vb[0] == "v" else emit("const", f32(vb[1]))
            r = emit("sub", ea, eb)
        else:
            terms = flat(arg, op)
            acc = None  # ('c', double) | ('v', id)
            for t in terms:
                tv = value(t)
                if acc is None:
                    acc = tv
                elif acc[0] == "c" and tv[0] == "c":
                    acc = ("c", acc[1] + tv[1] if op == "add" else acc[1] * tv[1])
                elif acc[0] == "c":
                    acc = ("v", emit("c" + op, f32(acc[1]), tv[1]))
                elif tv[0] == "c":
                    acc = ("v", emit("c" + op, f32(tv[1]), acc[1]))
                else:
                    acc = ("v", emit(op, acc[1], tv[1]))
            assert acc[0] == "v"
            r = acc[1]
        memo[arg] = r
        return ("v", r)

    outputs = []
    if ee_only:
        return ops, [("const", f32(e.c)) if e.is_param else ("op", value(e.n)[1]) for e in tr.ee_expr]
    for sp in tr.spheres:
        o = []
        for e in sp["expr"]:
            if e.is_param:
                o.append(("const", f32(e.c)))
            else:
                o.append(("op", value(e.n)[1]))
        outputs.append(o)
    return ops, outputs


def to_json(tr: Traced):
    ops, outputs = lower(tr)
    radii = [sp["r"] for sp in tr.spheres]
    fine_r = radii[:tr.n_fine]
    model = dict(
        name=tr.name,
        dimension=len(tr.joint_names),
        joint_names=tr.joint_names,
        end_effector=tr.end_effector,
        resolution=tr.resolution,
        n_spheres=tr.n_fine,
        n_bounding=len(tr.spheres) - tr.n_fine,
        min_radius=min(fine_r),
        max_radius=max(fine_r),
        lower=[float(np.float32(v)) for v in tr.lower],
        upper=[float(np.float32(v)) for v in tr.upper],
        span=[float(np.float32(u - l)) for u, l in zip(tr.upper, tr.lower)],
        descale=[float(np.float32(1.0 / (u - l))) for u, l in zip(tr.upper, tr.lower)],
        links=tr.link_order,
        sphere_link=[tr.link_order.index(sp["link"]) for sp in tr.spheres],
        radii=radii,
        ops=[[op, a, b] for (op, a, b) in ops],
        outputs=[[list(c) for c in o] for o in outputs],
        env_groups=tr.env_groups,
        self_groups=tr.self_groups,
        bounding_joint_frame={sp["link"]: sp["joint_frame"] for sp in tr.spheres if sp.get("bounding")},
        bounding_static={sp["link"]: sp["static"] for sp in tr.spheres if sp.get("bounding")},
    )
    if tr.ee_expr is not None:
        # Robot::fkcc_attach / eefk: translation xyz, then the rotation column-major (vector/math.hh to_isometry)
        ee_ops, ee_out = lower(tr, ee_only=True)
        model["ee_ops"] = [[op, a, b] for (op, a, b) in ee_ops]
        model["ee_outputs"] = [list(c) for c in ee_out]
    return model


ROBOTS = {
    "panda": dict(urdf="panda/panda_spherized.urdf", srdf="panda/panda.srdf", resolution=32,
                  end_effector="panda_grasptarget", attach_anchor="panda_hand",
                  # ~1e-18 of solver round-off the reference's generated code carries in the hand's bounding centre
                  # (robots/panda.hh fkcc, y[268..270]); recorded as model data so special configurations
                  # (e.g. q = 0, where everything else cancels) stay bit-exact.
                  bounding_residue={"panda_hand": {0: -5.20417042793042e-18, 1: -2.16840434497101e-18}},
                  joints=["panda_joint1", "panda_joint2", "panda_joint3", "panda_joint4", "panda_joint5",
                          "panda_joint6", "panda_joint7"]),
    "ur5": dict(urdf="ur5/ur5_spherized.urdf", srdf="ur5/ur5.srdf", resolution=32,
                end_effector="robotiq_85_base_link", attach_anchor="fts_robotside",
                # one pair the shipped ur5.srdf leaves enabled but the reference's generated checker never tests
                # (robots/ur5.hh has no "wrist_2_link vs. fts_robotside" group); recorded as model data.
                extra_disabled=[("wrist_2_link", "fts_robotside")],
                joints=["shoulder_pan_joint", "shoulder_lift_joint", "elbow_joint", "wrist_1_joint",
                        "wrist_2_joint", "wrist_3_joint"]),
    "fetch": dict(urdf="fetch/fetch_spherized.urdf", srdf="fetch/fetch.srdf", resolution=32,
                  end_effector="gripper_link", attach_anchor="gripper_link",
                  # The smallest enclosing ball of a mirror-symmetric sphere set has y = 0 exactly; the generator
                  # the reference used left ~1e-17 of solver round-off there (robots/fetch.hh fkcc, bounding
                  # spheres of the three static links).  Recorded as model data so the boolean stays bit-exact.
                  bounding_residue={"base_link": {1: -1.875 * 2.0 ** -55}, "torso_fixed_link": {1: 2.0 ** -55},
                                    "head_pan_link": {1: -1.75 * 2.0 ** -56},
                                    "shoulder_lift_link": {2: 1.73472347597681e-18},
                                    "upperarm_roll_link": {2: -1.73472347597681e-18},
                                    "elbow_flex_link": {2: 1.73472347597681e-18},
                                    "forearm_roll_link": {2: -1.73472347597681e-18},
                                    "wrist_flex_link": {2: 4.33680868994202e-18}},
                  joints=["torso_lift_joint", "shoulder_pan_joint", "shoulder_lift_joint", "upperarm_roll_joint",
                          "elbow_flex_joint", "forearm_roll_joint", "wrist_flex_joint", "wrist_roll_joint"]),
    "baxter": dict(urdf="baxter/baxter_spherized.urdf", srdf="baxter/baxter.srdf", resolution=64,
                   end_effector="right_gripper", attach_anchor="r_gripper_r_finger_tip",
                   joints=["left_s0", "left_s1", "left_e0", "left_e1", "left_w0", "left_w1", "left_w2",
                           "right_s0", "right_s1", "right_e0", "right_e1", "right_w0", "right_w1", "right_w2"]),
}


def main():
    res = os.environ.get("VAMP_RESOURCES", "/root/reference/resources")
    names = sys.argv[1:] or list(ROBOTS)
    outdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vamp_mvt_amd", "robots")
    for n in names:
        cfg = ROBOTS[n]
        tr = trace(os.path.join(res, cfg["urdf"]), os.path.join(res, cfg["srdf"]), cfg["joints"],
                   end_effector=cfg["end_effector"], resolution=cfg["resolution"], name=n,
                   bounding=cfg.get("bounding"), bounding_residue=cfg.get("bounding_residue"),
                   extra_disabled=cfg.get("extra_disabled", ()))
        model = to_json(tr)
        # Robot::fkcc_attach tests the attachment against the links the anchor link is tested against ("Attachment
        # vs. <link>" blocks of robots/<robot>.hh), in link order; the anchor is recorded per robot above
        anchor = cfg["attach_anchor"]
        partners = {g["b"] if g["a"] == anchor else g["a"] for g in model["self_groups"] if anchor in (g["a"], g["b"])}
        model["attach_links"] = [ln for ln in model["links"] if ln in partners]
        with open(os.path.join(outdir, n + ".json"), "w") as f:
            json.dump(model, f, separators=(",", ":"))
        print(n, "ops", len(model["ops"]), "spheres", model["n_spheres"], "+", model["n_bounding"],
              "env groups", len(model["env_groups"]), "self groups", len(model["self_groups"]),
              "fine pairs", sum(len(g["pairs"]) for g in model["self_groups"]))


if __name__ == "__main__":
    main()
