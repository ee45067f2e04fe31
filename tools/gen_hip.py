#!/usr/bin/env python3
"""Robot model JSON -> HIP device code for gfx950 (vamp_mvt_amd/csrc/gen/robots_dev.inc).

Kernel-side shape of one robot:

  template <int G> bool fkcc_env(E, q[dim], slab, skip)   per lane: "an environment group of this rake collides"
  template <int G> bool fkcc_self(q[dim], skip)           per lane: "a self-collision group of this rake collides"

  The reference's fkcc is the OR of the two; they are separate device functions (and separate kernels) because
  their resource profiles differ: the environment half needs an LDS slab and few registers (4+ waves per SIMD hide
  the LDS latency of the primitive loops), the self-collision half needs no LDS and many registers.

  * the FK op tape is emitted link by link along the kinematic chain, each link's ops just before its checks;
  * fkcc_env: the current link's spheres (bounding first, then chunks of CHUNK fine spheres) are written to the
    wave's LDS slab (lane-contiguous); vmv::env_gate / env_fine read them with wave-uniform or re-dealt indices;
  * fkcc_self: fully unrolled on registers; (r_a + r_b)^2 are literals computed here with the same two fp32
    roundings as the reference (`rs = ar + br; rs * rs`).
"""
from __future__ import annotations

import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..")


def flit(v: float) -> str:
    f = float(np.float32(v))
    if f == 0.0:
        return "-0.0f" if str(f).startswith("-") else "0.0f"
    return float.hex(f) + "f"


def op_expr(op, a, b, sin="vmv::vsin", cos="vmv::vcos", qname="q", prefix="t"):
    t = prefix
    if op == "in":
        return f"{qname}[{a}]"
    if op == "sin":
        return f"{sin}({t}{a})"
    if op == "cos":
        return f"{cos}({t}{a})"
    if op == "neg":
        return f"-{t}{a}"
    if op == "const":
        return flit(a)
    if op == "mul":
        return f"{t}{a} * {t}{b}"
    if op == "add":
        return f"{t}{a} + {t}{b}"
    if op == "sub":
        return f"{t}{a} - {t}{b}"
    if op == "cmul":
        return f"{flit(a)} * {t}{b}"
    if op == "cadd":
        return f"{flit(a)} + {t}{b}"
    raise ValueError(op)


def deps(op, a, b):
    if op in ("sin", "cos", "neg"):
        return [a]
    if op in ("mul", "add", "sub"):
        return [a, b]
    if op in ("cmul", "cadd"):
        return [b]
    return []


class Emitter:
    def __init__(self, m, qname="q", prefix="t", indent="        "):
        self.m = m
        self.ops = m["ops"]
        self.done = [False] * len(self.ops)
        self.lines = []
        self.qname, self.prefix, self.indent = qname, prefix, indent

    def need(self, spheres):
        """Emit (in tape order) every not-yet-emitted op the given spheres depend on."""
        want = set()
        stack = [v for s in spheres for (kind, v) in self.m["outputs"][s] if kind == "op"]
        while stack:
            i = stack.pop()
            if i in want or self.done[i]:
                continue
            want.add(i)
            stack += deps(*self.ops[i])
        for i in sorted(want):
            op, a, b = self.ops[i]
            self.lines.append(f"{self.indent}const float {self.prefix}{i} = "
                              f"{op_expr(op, a, b, qname=self.qname, prefix=self.prefix)};")
            self.done[i] = True

    def closure(self, spheres):
        """every op (emitted or not) the given spheres depend on"""
        want = set()
        stack = [v for s in spheres for (kind, v) in self.m["outputs"][s] if kind == "op"]
        while stack:
            i = stack.pop()
            if i in want:
                continue
            want.add(i)
            stack += deps(*self.ops[i])
        return want

    def emit_ops(self, op_set, indent=None):
        """emit (in tape order) the not-yet-emitted ops of op_set"""
        ind = self.indent if indent is None else indent
        for i in sorted(op_set):
            if self.done[i]:
                continue
            op, a, b = self.ops[i]
            self.lines.append(f"{ind}const float {self.prefix}{i} = "
                              f"{op_expr(op, a, b, qname=self.qname, prefix=self.prefix)};")
            self.done[i] = True

    def coord(self, s, k):
        kind, v = self.m["outputs"][s][k]
        return f"{self.prefix}{v}" if kind == "op" else flit(v)


def f32(x):
    return np.float32(x)


def eval_tape(m, q, batch=4096):
    """float64 evaluation of the model's op tape for configurations q[N][dim] -> sphere centres [N][n_total][3]
    (code-shape statistics and the clearance tables; the device code evaluates the tape itself)."""
    if q.shape[0] > batch:  # keep the intermediates in cache
        return np.concatenate([eval_tape(m, q[i:i + batch], batch) for i in range(0, q.shape[0], batch)])
    vals = [None] * len(m["ops"])
    for i, (op, a, b) in enumerate(m["ops"]):
        if op == "in":
            v = q[:, a]
        elif op == "sin":
            v = np.sin(vals[a])
        elif op == "cos":
            v = np.cos(vals[a])
        elif op == "neg":
            v = -vals[a]
        elif op == "mul":
            v = vals[a] * vals[b]
        elif op == "add":
            v = vals[a] + vals[b]
        elif op == "sub":
            v = vals[a] - vals[b]
        elif op == "cmul":
            v = a * vals[b]
        elif op == "cadd":
            v = a + vals[b]
        else:
            raise ValueError(op)
        vals[i] = v
    out = np.zeros((q.shape[0], len(m["outputs"]), 3))
    for s, o in enumerate(m["outputs"]):
        for k, (kind, v) in enumerate(o):
            out[:, s, k] = vals[v] if kind == "op" else v
    return out


LINK_SAMPLES_N = 48  # samples per joint of the per-link reach certificates


def link_samples(m, N=LINK_SAMPLES_N):
    """Reach certificates for the first links of the chain.  A link whose bounding-sphere centre depends on at most two
    REVOLUTE joints (found numerically; periodic, so the joint values need no bounds) gets a grid of N x N sample centres
    over [0, 2 pi)^2 and a slack: every centre the link can ever have lies within `slack` of a sample (lever arms
    rho = |dc/dq|, exact chords of a +-h rotation, maximum over the samples + 5 %, second-order term as in self_tables).
    At vmv_env_finalize a link is skipped for an environment when every primitive is farther than
    bounding radius + slack + 1e-3 m from every sample (vmv_api.hip).  -> {env group index: (samples[n][3], slack)}"""
    dim = m["dimension"]
    lo, span = np.array(m["lower"], float), np.array(m["span"], float)
    rng = np.random.default_rng(11)
    q0 = lo + span * rng.random((6, dim))
    c0 = eval_tape(m, q0)
    out = {}
    static = set(static_links(m))
    h = 1e-3
    for gi, g in enumerate(m["env_groups"]):
        if g["link"] in static:
            continue
        b = g["bound"]
        dep = []
        for j in range(dim):
            q1 = q0.copy()
            q1[:, j] += 0.37 * span[j] * np.where(q1[:, j] - lo[j] < 0.5 * span[j], 1.0, -1.0)
            if np.abs(eval_tape(m, q1)[:, b] - c0[:, b]).max() > 1e-9:
                dep.append(j)
        if not 1 <= len(dep) <= 2:
            continue
        # revolute = periodic with period 2 pi in every dependent joint
        periodic = True
        for j in dep:
            q1 = q0.copy()
            q1[:, j] += 2 * np.pi
            periodic = periodic and np.abs(eval_tape(m, q1)[:, b] - c0[:, b]).max() < 1e-9
        if not periodic:
            continue
        if len(dep) == 1:
            dep = dep + [dep[0]]
        i, j = dep
        t = 2 * np.pi * (np.arange(N) + 0.5) / N
        Q = np.tile(lo + 0.5 * span, (N * N, 1))
        if i == j:
            Q = Q[:N]
            Q[:, i] = t
        else:
            Q[:, i] = np.repeat(t, N)
            Q[:, j] = np.tile(t, N)

        def centres(axis, dq):
            Q2 = Q.copy()
            Q2[:, axis] += dq
            return eval_tape(m, Q2)[:, b]
        C = eval_tape(m, Q)[:, b]
        rho_i = np.linalg.norm(centres(i, h) - centres(i, -h), axis=1).max() / (2 * np.sin(h))
        rho_j = 0.0 if i == j else np.linalg.norm(centres(j, h) - centres(j, -h), axis=1).max() / (2 * np.sin(h))
        d = np.pi / N * 1.01  # half a cell
        slack = 1.05 * ((rho_i + rho_j * d) * d + (rho_j + rho_i * d) * d) + 1e-6
        out[gi] = (C, float(slack))
    return out


def gate_rates(m, n=4096, seed=0, tables=()):
    """Share of uniformly random configurations whose bounding-pair gate fires (and whose clearance-table bit, if the
    group has one, is set), per self-collision group."""
    rng = np.random.default_rng(seed)
    q = np.array(m["lower"]) + np.array(m["span"]) * rng.random((n, m["dimension"]))
    c = eval_tape(m, q)
    r = np.array(m["radii"])
    fire = [np.linalg.norm(c[:, g["bound_a"]] - c[:, g["bound_b"]], axis=1) < r[g["bound_a"]] + r[g["bound_b"]]
            for g in m["self_groups"]]
    for t in tables:
        N = t["table"].shape[0]
        (i, j), (loi, loj), (ii, ij) = t["joints"], t["lo"], t["inv"]
        cell = t["table"][np.clip(((q[:, i] - loi) * ii).astype(int), 0, N - 1), np.clip(((q[:, j] - loj) * ij).astype(int), 0, N - 1)]
        for bit, gi in enumerate(t["groups"]):
            fire[gi] = fire[gi] & (((cell >> bit) & 1) != 0)
    return [float(f.mean()) for f in fire]


SELF_TABLE_N = int(os.environ.get("VMV_SELF_TABLE_N", 256))  # cells per joint of the two-joint clearance tables (0: none)
SELF_TABLE_MARGIN = 1e-4  # metres, on top of the Lipschitz slack of a cell (fp32 effects at metre scale are ~1e-6)


def self_tables(m, N=SELF_TABLE_N):
    """Two-joint clearance tables for the self-collision half.

    The distance of a fine pair (a on link A, b on link B) depends only on the joints between A and B.  For the groups
    where these are at most two joints (i, j), a table over (q_i, q_j) says for every cell whether the group is
    CERTAINLY free there: every pair's clearance at the cell centre exceeds what it can lose inside the cell plus
    SELF_TABLE_MARGIN.  What a pair can lose: |d/dq_i |pb - pa|| <= rho_i = |d(pb - pa)/dq_i| (the lever arm about joint i,
    measured as the exact chord of a +-h rotation), likewise rho_j; rho of the outer joint is constant, rho of the inner
    one changes by at most rho_outer * dq_outer inside the cell (both bounds are applied symmetrically, so the order of the
    joints does not matter); + 10 % and cells taken 1 % larger than they are (fp32 cell index at the borders).  A group
    whose bit is 0 for a configuration's cell has no colliding fine pair there, so skipping it cannot change the answer;
    outside the joint bounds (and for NaN) every bit is 1.  -> [dict(joints=(i, j), groups=[index into self_groups...],
    table=uint8[N][N] (bit g = group g of this table must be tested), lo=(..), inv=(..))]"""
    if N <= 0:
        return []
    dim = m["dimension"]
    lo, span = np.array(m["lower"], float), np.array(m["span"], float)
    r = np.array(m["radii"], float)
    rng = np.random.default_rng(7)
    q0 = lo + span * rng.random((6, dim))
    c0 = eval_tape(m, q0)
    by_joints = {}
    for gi, g in enumerate(m["self_groups"]):
        pr = np.array(g["pairs"])
        d0 = np.linalg.norm(c0[:, pr[:, 0]] - c0[:, pr[:, 1]], axis=2)
        dep = []
        for j in range(dim):
            q1 = q0.copy()
            q1[:, j] += 0.37 * span[j] * np.where(q1[:, j] - lo[j] < 0.5 * span[j], 1.0, -1.0)
            c1 = eval_tape(m, q1)
            if np.abs(np.linalg.norm(c1[:, pr[:, 0]] - c1[:, pr[:, 1]], axis=2) - d0).max() > 1e-9:
                dep.append(j)
        if 1 <= len(dep) <= 2:
            if len(dep) == 1:
                dep = dep + [dep[0] + 1 if dep[0] + 1 < dim else dep[0] - 1]  # any second joint: constant along it
            by_joints.setdefault(tuple(sorted(dep)), []).append(gi)
    out = []
    h = 1e-3
    for (i, j), gis in sorted(by_joints.items()):
        for c in range(0, len(gis), 8):
            chunk = gis[c:c + 8]
            qi = lo[i] + span[i] * (np.arange(N) + 0.5) / N
            qj = lo[j] + span[j] * (np.arange(N) + 0.5) / N
            Q = np.tile(lo + 0.5 * span, (N * N, 1))
            Q[:, i] = np.repeat(qi, N)
            Q[:, j] = np.tile(qj, N)
            hi_, hj_ = 0.5 * span[i] / N * 1.01, 0.5 * span[j] / N * 1.01

            def shifted(axis, dq):
                Q2 = Q.copy()
                Q2[:, axis] += dq
                return eval_tape(m, Q2)
            C = eval_tape(m, Q)
            Cip, Cim, Cjp, Cjm = shifted(i, h), shifted(i, -h), shifted(j, h), shifted(j, -h)
            table = np.zeros(N * N, np.uint8)
            shares = []
            for bit, gi in enumerate(chunk):
                pr = np.array(m["self_groups"][gi]["pairs"])
                a, b = pr[:, 0], pr[:, 1]
                v = C[:, b] - C[:, a]
                clear = np.linalg.norm(v, axis=2) - r[a] - r[b]
                rho_i = np.linalg.norm((Cip[:, b] - Cip[:, a]) - (Cim[:, b] - Cim[:, a]), axis=2) / (2 * np.sin(h))
                rho_j = np.linalg.norm((Cjp[:, b] - Cjp[:, a]) - (Cjm[:, b] - Cjm[:, a]), axis=2) / (2 * np.sin(h))
                slack = 1.1 * ((rho_i + rho_j * hj_) * hi_ + (rho_j + rho_i * hi_) * hj_) + SELF_TABLE_MARGIN
                must = (clear <= slack).any(axis=1)
                table |= (must.astype(np.uint8) << bit)
                shares.append(1.0 - float(must.mean()))
            out.append(dict(joints=(i, j), groups=chunk, table=table.reshape(N, N), lo=(float(lo[i]), float(lo[j])),
                            inv=(N / float(span[i]), N / float(span[j])), free_share=shares))
    return out


SELF_DEAL_MAX_A = 40  # A-side spheres kept in registers for the re-dealt self-collision form
ENV_CHUNK = {"panda": 4, "ur5": 6, "baxter": 5}   # fine spheres per slab chunk in the environment kernels (default CHUNK); a smaller slab
ENV_BLOCKS = {"panda": 6, "ur5": 5, "baxter": 5}  # (baxter: 0.331 -> 0.307 ms per 1M configs at 5)  # ... lets five or six workgroups (20 - 24 waves) share a CU's LDS where the registers allow it
# Panda, round 3: with the merged gates and the FK of a group emitted chunk by chunk the three-list kernel needs 81 VGPRs;
# at six workgroups per CU (80 VGPRs, no spill) and chunks of 4 (23.9 KB of LDS per workgroup) it runs 0.1375 -> 0.1288 ms
# per 1M configurations.  The same setting costs UR5 1.5 % (and 2.5 % on its edges): it stays at 5 x 6.
# workgroups per CU the self kernel is compiled for (512 / blocks VGPRs per lane) and fine spheres per slab chunk of the
# self-collision kernels.  Panda, measured (blocks x chunk, self kernel ms per 1M configs): 3x8 0.145, 4x8 0.133,
# 4x7 0.127, 4x6 0.126, 4x5 0.132, 4x4 0.140, 5x5 0.210 - with chunks of 8 the fourth workgroup did not fit the LDS.
# (baxter: 0.289 -> 0.227 ms at 3.  Fetch was faster at 2 than at 3 while its 41.3 KB of LDS allowed three workgroups; without
# the attachment radii (40.2 KB) four fit: 128 VGPRs, 11 - 13 spilled: config 3 0.581 -> 0.569 ms, its edges 1.617 -> 1.538)
SELF_BLOCKS = {"panda": 4, "ur5": 4, "fetch": 4, "baxter": 3}
SELF_CHUNK = {"panda": 6, "ur5": 5}
# the (edge, rake) task kernel of the self-collision half: the configuration kernel's bound unless listed.  UR5 with chunks of
# 5 fits five workgroups per CU (96 VGPRs, 13 spilled, 30.4 KB of LDS): BASELINE config 4 5.46 -> 5.20 ms; its configuration
# kernel is 1.5 % slower that way and stays at four
MOTION_SELF_BLOCKS = {"ur5": 5}
# fused one-FK kernels (the task kernel of planner-sized edge batches, n < 16,384): at 3 workgroups per CU (168 VGPRs) Panda's
# body spills 28 VGPRs instead of 77 at 4, and such batches never fill more than 2 - 3 waves per SIMD: 256 edges 0.104 ->
# 0.096 ms, 2,048 0.133 -> 0.126, 8,192 0.148 -> 0.135 (at 2: 0.094 / 0.123 / 0.148); UR5 (no spill at 3) unchanged
FUSED_BLOCKS = {r: int(os.environ.get("VMV_FUSED_BLOCKS", 3)) for r in ("panda", "ur5")}
for _r in ("panda", "ur5", "fetch", "baxter"):  # tuning knobs: VMV_{SELF,ENV}_BLOCKS_<ROBOT>, VMV_{SELF,ENV}_CHUNK_<ROBOT>
    if f"VMV_ENV_BLOCKS_{_r.upper()}" in os.environ:
        ENV_BLOCKS[_r] = int(os.environ[f"VMV_ENV_BLOCKS_{_r.upper()}"])
    if f"VMV_ENV_CHUNK_{_r.upper()}" in os.environ:
        ENV_CHUNK[_r] = int(os.environ[f"VMV_ENV_CHUNK_{_r.upper()}"])
    if f"VMV_SELF_BLOCKS_{_r.upper()}" in os.environ:
        SELF_BLOCKS[_r] = int(os.environ[f"VMV_SELF_BLOCKS_{_r.upper()}"])
    if f"VMV_SELF_CHUNK_{_r.upper()}" in os.environ:
        SELF_CHUNK[_r] = int(os.environ[f"VMV_SELF_CHUNK_{_r.upper()}"])
    if f"VMV_MOTION_SELF_BLOCKS_{_r.upper()}" in os.environ:
        MOTION_SELF_BLOCKS[_r] = int(os.environ[f"VMV_MOTION_SELF_BLOCKS_{_r.upper()}"])
SELF_DENSE_RATE = float(os.environ.get('VMV_SELF_DENSE_RATE', 0.5))   # groups whose bounding-pair gate fires for at least this share of uniform configurations ...
SELF_DENSE_MIN_A = 3    # ... and whose A side is at least this large use the pre-test + compaction form
SPARSE_BATCH = 8        # sparse groups merged per item list (the list holds SPARSE_BATCH * 64 entries = CHUNK * 64)
SELF_MARGIN = 1e-4      # metres; enclosure of fine spheres by bounding spheres is asserted to 2e-6 by tools/robot_trace.py
LAZY_FINE_FK = os.environ.get("VMV_LAZY_FINE_FK", "1") == "1"  # see emit_env_link
DEFAULT_CHUNK = 8  # fine spheres staged in the LDS slab at a time
assert SPARSE_BATCH <= 8 and DEFAULT_CHUNK <= 8  # vmv::kSelfScratchWords holds 8 * 64 list entries


GRID_CLASSES = 4  # vmv::kGridClasses


MERGE_RIGID_LINKS = os.environ.get("VMV_NO_MERGED_GROUPS") is None  # (A/B knob at generation time)
# distances inside a rigid cluster still vary by a few 1e-9 m over the configurations (the tape's fixed-joint rotations
# are 15-digit constants, not exactly orthonormal); anything a joint moves varies by millimetres and more
RIGID_TOL = 1e-7
MERGED_GATE_MAX = 0.13  # metres: links join a common gate while it stays below this (or 1.3 x the largest member's)
_MERGED_CACHE = {}


def merged_groups(m):
    """Environment groups of the PRIMITIVE-ONLY kernel variants: consecutive links whose fine spheres keep a constant
    distance to one bounding centre (links hanging on the same moving joint: a gripper's fingers and pads, a wrist's
    flange) share ONE gate.  For environments made of primitives the reference's answer is the OR over the fine spheres
    (a flat evaluation reproduces fkcc on 1,000,000 / 1,000,000 configurations) and the bounding-sphere
    gate only decides which of them are tested, so any sphere that encloses a group's fine spheres is a valid gate;
    the variant that serves point clouds keeps the reference's groups (its answers depend on them).
    -> list, in chain order, of dict(link=name, members=[links], bound=sphere id whose CENTRE is the gate's centre,
    fine=[sphere ids], radius=gate radius)."""
    if m["name"] in _MERGED_CACHE:
        return _MERGED_CACHE[m["name"]]
    env_by_link = {g["link"]: g for g in m["env_groups"]}
    static = set(static_links(m))
    rng = np.random.default_rng(12345)
    lo, sp = np.array(m["lower"]), np.array(m["span"])
    q = lo + sp * rng.random((96, m["dimension"]))
    q[:32] *= 2.5  # far outside the bounds too: the distances must not depend on the configuration at all
    C = eval_tape(m, q)  # float64 centres [96][n_total][3]
    R = np.array(m["radii"], np.float64)

    def reach(anchor, spheres):
        """(largest distance + radius of `spheres` from the centre of sphere `anchor`, its spread over the configurations)"""
        d = np.linalg.norm(C[:, spheres, :] - C[:, [anchor], :], axis=2) + R[spheres][None, :]
        return float(d.max()), float((d.max(axis=0) - d.min(axis=0)).max())

    out, cur = [], None
    for ln in m["links"]:
        g = env_by_link[ln]
        if ln in static or not MERGE_RIGID_LINKS:
            cur = None
            out.append(dict(link=ln, members=[ln], bound=g["bound"], fine=list(g["fine"]), radius=m["radii"][g["bound"]]))
            continue
        if cur is not None and reach(cur["bound"], g["fine"])[1] < RIGID_TOL:
            # joining is worth it while the common gate stays small: a gate the size of a torso fires for every
            # configuration and hands every primitive near it to every fine sphere of the group
            members, fine = cur["members"] + [ln], cur["fine"] + g["fine"]
            cands = [reach(env_by_link[x]["bound"], fine) for x in members]
            far = min(f for f, spread in cands if spread < RIGID_TOL)
            biggest = max(m["radii"][env_by_link[x]["bound"]] for x in members)
            if far <= max(MERGED_GATE_MAX, 1.3 * biggest):
                cur["members"], cur["fine"] = members, fine
                continue
        cur = dict(link=ln, members=[ln], bound=g["bound"], fine=list(g["fine"]), radius=m["radii"][g["bound"]])
        out.append(cur)
    for grp in out:
        if len(grp["members"]) == 1:
            continue
        # the member whose bounding centre gives the smallest gate (every candidate re-checked for constancy)
        best = None
        for ln in grp["members"]:
            b = env_by_link[ln]["bound"]
            far, spread = reach(b, grp["fine"])
            if spread < RIGID_TOL and (best is None or far < best[0]):
                best = (far, b)
        far, b = best
        # + 5e-6 m: fp32 FK moves centres by ~1e-7 m; the candidate margin of the gate (1e-4 m) is counted from here
        grp["bound"], grp["radius"] = b, float(np.nextafter(np.float32(far + 5e-6), np.float32(np.inf)))
        grp["link"] = "+".join(grp["members"])
    _MERGED_CACHE[m["name"]] = out
    return out


def grid_classes(m):
    """Split the links into GRID_CLASSES classes by bounding radius (contiguous in sorted order), minimising the summed
    candidate volume (R_class + typical primitive size + cell)^3 over the links.  -> (class radii, {link: class}).
    The gates of the merged groups (merged_groups) take part under their own names, with their own radii."""
    links = [(m["radii"][g["bound"]], g["link"]) for g in m["env_groups"]]
    links += [(grp["radius"], grp["link"]) for grp in merged_groups(m) if len(grp["members"]) > 1]
    links.sort()
    r = [x[0] for x in links]
    n = len(r)
    cost = lambda i, j: (j - i) * (r[j - 1] + 0.1) ** 3  # links i..j-1 share the radius r[j-1]
    INF = float("inf")
    best = [[INF] * (n + 1) for _ in range(GRID_CLASSES + 1)]
    cut = [[0] * (n + 1) for _ in range(GRID_CLASSES + 1)]
    best[0][0] = 0.0
    for k in range(1, GRID_CLASSES + 1):
        for j in range(1, n + 1):
            for i in range(k - 1, j):
                c = best[k - 1][i] + cost(i, j)
                if c < best[k][j]:
                    best[k][j], cut[k][j] = c, i
    k = min(GRID_CLASSES, n)
    bounds, j = [], n
    while k > 0:
        i = cut[k][j]
        bounds.append((i, j))
        j, k = i, k - 1
    bounds.reverse()
    radii, cls = [], {}
    for ci, (i, j) in enumerate(bounds):
        radii.append(r[j - 1])
        for t in range(i, j):
            cls[links[t][1]] = ci
    while len(radii) < GRID_CLASSES:
        radii.append(radii[-1])
    return radii, cls


def static_links(m):
    """links whose spheres do not depend on the configuration"""
    out = []
    for g in m["env_groups"]:
        if all(kind != "op" for s in [g["bound"]] + g["fine"] for kind, v in m["outputs"][s]):
            out.append(g["link"])
    return out


def emit_robot(m):
    n = m["name"]
    CHUNK = SELF_CHUNK.get(n, DEFAULT_CHUNK)
    assert 1 <= CHUNK <= 8
    L = []
    dim = m["dimension"]
    links = m["links"]
    radii = m["radii"]
    env_by_link = {g["link"]: g for g in m["env_groups"]}
    self_by_b = {}
    for g in m["self_groups"]:
        self_by_b.setdefault(g["b"], []).append(g)
    self_links = {g["a"] for g in m["self_groups"]} | {g["b"] for g in m["self_groups"]}

    # constant table: radii per link group, [bounding, fine...]
    radii_tab, radii_off = [], {}
    for ln in links:
        g = env_by_link[ln]
        radii_off[ln] = len(radii_tab)
        radii_tab += [radii[g["bound"]]] + [radii[s] for s in g["fine"]]
    # the merged gates of the primitive-only variants (merged_groups): their own radius, then their fine spheres' radii
    prim_order = []
    merged_radius = {}
    for grp in merged_groups(m):
        prim_order.append(grp["link"])
        if len(grp["members"]) > 1:
            env_by_link[grp["link"]] = dict(link=grp["link"], bound=grp["bound"], fine=grp["fine"])
            merged_radius[grp["link"]] = grp["radius"]
            radii_off[grp["link"]] = len(radii_tab)
            radii_tab += [grp["radius"]] + [radii[s] for s in grp["fine"]]
    max_group = max(len(g["fine"]) for g in m["env_groups"])  # largest link (the self-collision kernels stage links)
    max_env_group = max([max_group] + [len(grp["fine"]) for grp in merged_groups(m)])  # largest gate of any variant
    env_chunk = ENV_CHUNK.get(n, DEFAULT_CHUNK)
    slab_spheres = min(env_chunk, max_env_group)  # rows of the environment slab: one chunk of fine spheres (the bounding sphere travels in registers)
    self_slab_spheres = 1 + min(CHUNK, max_group)

    L.append(f"namespace {n}")
    L.append("{")
    L.append(f"    constexpr int kDim = {dim};")
    L.append(f"    constexpr int kNSpheres = {m['n_spheres']};")
    L.append(f"    constexpr int kResolution = {m['resolution']};")
    L.append(f"    constexpr int kSlabSpheres = {slab_spheres};  // environment kernels: one chunk of fine spheres")
    L.append(f"    constexpr int kSelfSlabSpheres = {self_slab_spheres};  // self-collision kernels: one chunk of B spheres")
    L.append(f"    constexpr int kNRadii = {len(radii_tab)};")
    L.append(f"    __constant__ float kRadii[{len(radii_tab)}] = {{" + ", ".join(flit(v) for v in radii_tab) + "};")
    L.append("    struct Tab")
    L.append("    {")
    L.append("        static __device__ __forceinline__ float radius(int i) { return kRadii[i]; }")
    L.append("    };")
    L.append("")

    # ---- environment half of fkcc ------------------------------------------------------------------------
    class_radii, link_class = grid_classes(m)
    static = set(static_links(m))
    reach = link_samples(m)

    def emit_env_link(em, ln, lazy=LAZY_FINE_FK, no_skip=False, pair_with=None, pre=None, order=None):
        """one link of the environment half: FK ops, slab staging, gate, fine chunks (appends to em.lines).
        lazy: the FK ops only this link's fine spheres need, and the staging of its first chunk, are emitted inside
        `if (wave_any(gate))` - links whose bounding sphere never reaches an obstacle (the base links in a shell-shaped
        scene) then cost their chain ops and one cell lookup, nothing else."""
        order = links if order is None else order  # the groups of this walk, in chain order
        g = env_by_link[ln]
        fine = g["fine"]
        chunks = [fine[i:i + env_chunk] for i in range(0, len(fine), env_chunk)]
        assert all(len(ch) <= slab_spheres for ch in chunks), (ln, slab_spheres)  # a chunk never outgrows the slab
        if ln in static:
            em.lines.append(f"        // ---- {ln}: static, evaluated once per environment (static_env_hit)")
            return
        if ln in os.environ.get("VMV_ABLATE_SKIP_LINKS", "").split(","):  # measurement aid (wrong answers)
            return
        em.lines.append(f"        // ---- {ln}: {len(fine)} spheres")
        gi_env = m["env_groups"].index(g) if g in m["env_groups"] else -1  # (merged gates carry no reach certificate)
        guarded = gi_env in reach and not no_skip
        if guarded:
            # reach certificate (link_samples): for environments no primitive of which this link can ever touch, the
            # launcher sets the link's bit and the wave skips its bounding-sphere FK, cell lookup and gate
            later_all = [s for other in order[order.index(ln) + 1:] if other not in static
                         for s in [env_by_link[other]["bound"]] + env_by_link[other]["fine"]]
            em.emit_ops(em.closure([g["bound"]] + fine) & em.closure(later_all))  # what later links need stays outside
            em.lines.append(f"        if (((skip_links >> {gi_env}) & 1ull) == 0ull)")
            em.lines.append("        {")
        private = set()
        if lazy:
            later = [s for other in order[order.index(ln) + 1:] if other not in static
                     for s in [env_by_link[other]["bound"]] + env_by_link[other]["fine"]]
            em.need([g["bound"]])
            fine_ops = em.closure(fine)
            private = fine_ops - em.closure(later)
            em.emit_ops(fine_ops - private)
        else:
            em.need([g["bound"]] + fine)

        def stage(slot, s, indent):
            for k in range(3):
                em.lines.append(f"{indent}slab[{3 * slot + k} * vmv::kRow] = {em.coord(s, k)};")

        # (the bounding sphere goes to the gate in registers: no slab row, no LDS round trip between caller and gate)
        bc = ", ".join(em.coord(g["bound"], k) for k in range(3))
        if not lazy:
            for si, s in enumerate(chunks[0]):
                stage(si, s, "        ")
        if pair_with is not None:
            # paired variant: the point-cloud queries of this link's bounding sphere and of the next link's, together
            # (vmv::capt_gate_pair); bit 0 = this link, bit 1 = the next one (consumed by its own gate below)
            gb = env_by_link[pair_with]
            em.need([gb["bound"]])
            ca = ", ".join(em.coord(g["bound"], k) for k in range(3))
            cb = ", ".join(em.coord(gb["bound"], k) for k in range(3))
            em.lines.append(f"        const unsigned {pre[0]} = vmv::capt_gate_pair<G, Tab>(E, {ca}, {radii_off[ln]}, {cb}, "
                            f"{radii_off[pair_with]}, !bad);  // {ln} + {pair_with}")
        em.lines.append("        {")
        if pre is not None:
            # (V = kEnvClouds: nothing is left for the gate to test, the lanes whose query hit are listed without a call)
            em.lines.append("            bool gate;")
            em.lines.append(f"            if constexpr (V == vmv::kEnvClouds) gate = vmv::env_list_active<G>(scratch, ({pre[0]} & {pre[1]}u) != 0u && !bad);")
            em.lines.append(f"            else gate = vmv::env_gate<G, Tab, V, true>(E, {bc}, scratch, {radii_off[ln]}, "
                            f"{link_class[ln]}, !bad, ({pre[0]} & {pre[1]}u) != 0u);")
        else:
            em.lines.append(f"            const bool gate = vmv::env_gate<G, Tab, V>(E, {bc}, scratch, {radii_off[ln]}, {link_class[ln]}, !bad);")
        em.lines.append("            const int n_gate = __popcll(__ballot(gate));  // passing lanes (whole rakes), wave-uniform")
        em.lines.append("            if (VMV_ABLATE_ENV >= 1) bad |= gate;  // measurement aid: no fine phase (wrong answers)")
        em.lines.append("            else if (n_gate != 0)")
        em.lines.append("            {")
        if lazy:
            # (chunk by chunk: the FK of a later chunk's spheres is emitted right before that chunk is staged, so the
            # coordinates of a 27-sphere group are never all live across the fine calls)
            em.emit_ops(private & em.closure(chunks[0]), indent="                ")
            for si, s in enumerate(chunks[0]):
                stage(si, s, "                ")
        done = 0
        for ci, ch in enumerate(chunks):
            if ci > 0:
                if lazy:
                    em.emit_ops(private & em.closure(ch), indent="                ")
                for si, s in enumerate(ch):
                    stage(si, s, "                ")
            em.lines.append(f"                vmv::env_fine<G, Tab, V>(E, slab, scratch, {len(ch)}, {radii_off[ln] + 1 + done}, 0, n_gate);")
            done += len(ch)
        em.lines.append("                bad |= gate && vmv::group_any<G>(vmv::env_flag(scratch));")
        em.lines.append("            }")
        em.lines.append("        }")
        if guarded:
            em.lines.append("        }")

    def env_function(name, template, paired):
        L.append(f"    template <{template}>")
        L.append("    __device__ __forceinline__ bool")
        L.append(f"    {name}(const vmv::EnvView &E, const float (&q)[kDim], vmv::lds_ptr slab, const bool skip)")
        L.append("    {")
        if not paired:
            L.append("#ifndef VMV_NO_CAPT_PAIR  // (A/B knob, tools/build_variant.py: every gate queries the clouds on its own)")
            L.append("        if constexpr (V == vmv::kEnvFull || V == vmv::kEnvClouds) return fkcc_env_paired<G, V>(E, q, slab, skip);")
            L.append("#endif")
        L.append("        bool bad = skip || (E.dev->static_hit != 0u);")
        L.append("        // per-wave scratch words live right behind the sphere slab")
        L.append("        const vmv::lds_ptr scratch = slab - __lane_id() + kSlabSpheres * 3 * vmv::kRow;")
        if not paired:
            L.append("        const unsigned long long skip_links = E.dev->link_skip;  // reach certificates (vmv_api.hip), wave-uniform")
        em = Emitter(m)
        skipped = os.environ.get("VMV_ABLATE_SKIP_LINKS", "").split(",")
        movable = [ln for ln in links if ln not in static and ln not in skipped]
        if not paired:
            # primitive-only variants: rigidly connected links share one gate (merged_groups)
            for ln in prim_order:
                emit_env_link(em, ln, order=prim_order)
        for ln in (links if paired else []):
            if ln not in movable:
                emit_env_link(em, ln)
                continue
            # links (0, 1), (2, 3), ... of the chain share one pair of point-cloud queries; reach certificates never
            # apply to the environments this variant serves (they hold a heightfield or a point cloud), so no guards
            i = movable.index(ln)
            var = f"pre{i // 2}"
            if i % 2 == 0 and i + 1 == len(movable):
                emit_env_link(em, ln, no_skip=True)  # odd one out: queries on its own
            elif i % 2 == 0:
                emit_env_link(em, ln, no_skip=True, pair_with=movable[i + 1], pre=(var, 1))
            else:
                emit_env_link(em, ln, no_skip=True, pre=(var, 2))
        L.extend(em.lines)
        L.append("        return bad;")
        L.append("    }")
        L.append("")

    L.append("    // Environment half of Robot::fkcc<rake> (reference robots/%s.hh `fkcc`, \"environment vs. robot" % n)
    L.append("    // collisions\"): true = some link group of this rake reports a collision.")
    L.append("    // `skip` (rake-uniform): this rake's answer is not needed; it only keeps the lanes converged.")
    L.append("    // fkcc_env_paired: the same walk for the variant that serves environments with point clouds / heightfields")
    L.append("    // (V = kEnvFull, kEnvClouds): the CAPT queries of the bounding spheres of two consecutive links are issued together")
    L.append("    // (vmv::capt_gate_pair, two dependent-fetch chains in flight per wave) ahead of the first link's gate; each gate")
    L.append("    // then takes its answer instead of querying.  Same predicates on the same spheres: the OR is unchanged.")
    env_function("fkcc_env_paired", "int G, int V", True)
    env_function("fkcc_env", "int G, int V", False)

    # ---- static links ------------------------------------------------------------------------------------------
    L.append("    // The environment groups of the links that never move, for every lane alike: run once per (environment, robot)")
    L.append("    // by static_links_kernel; fkcc_env starts from its answer (EnvDev::static_hit) and skips these links.")
    L.append("    __device__ __forceinline__ bool static_env_hit(const vmv::EnvView &E)")
    L.append("    {")
    L.append("        bool hit = false;")
    for ln in links:
        if ln not in static:
            continue
        g = env_by_link[ln]
        b = g["bound"]
        bc = [flit(m["outputs"][b][k][1]) for k in range(3)]
        L.append(f"        if (vmv::env_hit<1, 0>(E, {bc[0]}, {bc[1]}, {bc[2]}, {flit(radii[b])}, true, nullptr))  // {ln}")
        L.append("        {")
        for s_ in g["fine"]:
            c = [flit(m["outputs"][s_][k][1]) for k in range(3)]
            L.append(f"            hit |= vmv::env_hit<1, 0>(E, {c[0]}, {c[1]}, {c[2]}, {flit(radii[s_])}, true, nullptr);")
        L.append("        }")
    L.append("        return hit;")
    L.append("    }")
    L.append(f"    constexpr int kNStaticLinks = {len(static)};")
    L.append("")

    # ---- self-collision half of fkcc -----------------------------------------------------------------------
    # Groups (A, B) are handled in passes: each pass keeps the spheres of a batch of A links in registers (at most
    # SELF_DEAL_MAX_A spheres) and walks the chain once, so robots whose A side does not fit run several passes
    # (FK is recomputed per pass; it is cheap next to spilling).
    a_links = [ln for ln in links if any(sg["a"] == ln for sg in m["self_groups"])]
    link_size = {ln: 1 + len(env_by_link[ln]["fine"]) for ln in links}
    batches, cur, cur_n = [], [], 0
    for ln in a_links:
        if cur and cur_n + link_size[ln] > SELF_DEAL_MAX_A:
            batches.append(cur)
            cur, cur_n = [], 0
        cur.append(ln)
        cur_n += link_size[ln]
    if cur:
        batches.append(cur)
    tables = self_tables(m)
    table_bit = {}  # index into self_groups -> (table number, bit)
    for ti, t in enumerate(tables):
        for bit, gi in enumerate(t["groups"]):
            table_bit[gi] = (ti, bit)
    group_index = {id(sg): gi for gi, sg in enumerate(m["self_groups"])}
    rates = gate_rates(m, tables=tables)
    dense_ids = {id(sg) for sg, r in zip(m["self_groups"], rates)
                 if r >= SELF_DENSE_RATE and len({p[0] for p in sg["pairs"]}) >= SELF_DENSE_MIN_A}
    if tables:
        L.append("    // Two-joint clearance tables (tools/gen_hip.py: self_tables): bit g of a cell = group g of the table may have a")
        L.append("    // colliding fine pair somewhere in the cell; 0 = certainly free (every pair's clearance at the cell centre")
        L.append(f"    // exceeds the Lipschitz slack of the cell + {SELF_TABLE_MARGIN} m).  Outside the joint bounds every bit is 1.")
        for ti, t in enumerate(tables):
            N = t["table"].shape[0]
            names = ", ".join(f"bit {b}: {m['self_groups'][gi]['a']} vs. {m['self_groups'][gi]['b']} ({fs * 100:.1f} % of the cells free)"
                              for b, (gi, fs) in enumerate(zip(t["groups"], t["free_share"])))
            L.append(f"    // table {ti}: joints {t['joints'][0]}, {t['joints'][1]}; {names}")
            L.append(f"    __device__ const unsigned char kSelfTable{ti}[{N} * {N}] = {{")
            flat = t["table"].reshape(-1)
            for k in range(0, len(flat), 64):
                L.append("        " + ",".join(str(int(v)) for v in flat[k:k + 64]) + ",")
            L.append("    };")
            L.append(f"    __device__ __forceinline__ unsigned self_table{ti}(const float (&q)[kDim])")
            L.append("    {")
            L.append(f"        const float fi = (q[{t['joints'][0]}] - {flit(t['lo'][0])}) * {flit(t['inv'][0])};")
            L.append(f"        const float fj = (q[{t['joints'][1]}] - {flit(t['lo'][1])}) * {flit(t['inv'][1])};")
            L.append(f"        const bool in = fi >= 0.0f && fj >= 0.0f && fi < {N}.0f && fj < {N}.0f;  // (false for NaN)")
            L.append(f"        typedef const unsigned char __attribute__((address_space(1))) *gb_cptr;")
            L.append(f"        return in ? (unsigned) ((gb_cptr) kSelfTable{ti})[(unsigned) fi * {N}u + (unsigned) fj] : 0xffu;")
            L.append("    }")
    L.append("    // Self-collision half of Robot::fkcc<rake> (\"robot self-collisions\").")
    L.append("    // Groups (A, B) run when B is the current link; gates (bounding pair) are per lane, exact.")
    L.append("    //  * sparse groups (gate rarely fires): the (passing lane, group) pairs of ALL sparse groups of this B are")
    L.append("    //    listed together and their fine pairs re-dealt over the 64 lanes as (entry, B sphere) items - one LDS")
    L.append("    //    round trip per chunk instead of one per group; B's sphere comes from the LDS slab column of the owning")
    L.append("    //    lane, A's spheres from that lane's registers through ds_bpermute (__shfl);")
    L.append("    //  * dense groups (gate fires for most configurations; chosen here from sampled gate rates): every lane")
    L.append("    //    pre-tests its own B spheres against A's bounding sphere and its A spheres against B's (a fine pair can")
    L.append("    //    only overlap if both reach; margin kSelfMargin >> fp32 effects, enclosure asserted by the tracer), the")
    L.append("    //    surviving (lane, B sphere) items are compacted in LDS and only those run the exact pair tests, on the")
    L.append("    //    A spheres that are candidates for some item of the batch.  Pruning only; answers are unchanged.")
    L.append("    // Hits return through LDS flags.  Every __shfl runs with all lanes enabled (ds_bpermute reads 0 from a")
    L.append("    // disabled source lane).")
    L.append(f"    // {len(batches)} pass(es) over the chain; each keeps one batch of A links in registers.")
    L.append("    template <int G>")
    L.append("    __device__ __forceinline__ bool")
    L.append("    fkcc_self(const float (&q)[kDim], vmv::lds_ptr slab, const vmv::lds_cptr radii_, const bool skip)")
    L.append("    {")
    L.append("        bool bad = skip;")
    L.append("        const unsigned lane = __lane_id();")
    L.append("        const vmv::lds_cptr wave_slab = vmv::uniform((vmv::lds_cptr) (slab - lane));")
    L.append("        const vmv::lds_cptr radii = vmv::uniform(radii_);")
    L.append("        vmv::lds_u32 *const list = (vmv::lds_u32 *) (slab - lane + kSelfSlabSpheres * 3 * vmv::kRow);")
    L.append("        vmv::lds_u32 *const flags = list + vmv::kWave;")
    L.append("        vmv::lds_u32 *const cand = list + 2 * vmv::kWave + 4;  // A-side candidate word per owner lane")
    L.append(f"        vmv::lds_u32 *const list2 = cand + vmv::kWave;        // item lists: (owner lane | tag << 6), <= {max(CHUNK, SPARSE_BATCH)} * 64 entries")
    L.append(f"        static_assert(vmv::kSelfScratchWords >= 3 * vmv::kWave + 4 + {CHUNK} * vmv::kWave, \"self-collision scratch\");")
    for ti in range(len(tables)):
        L.append(f"        const unsigned tb{ti} = (VMV_ABLATE_SELF == 16) ? 0xffu : self_table{ti}(q);  // issued first: one load, hidden behind FK")
    def emit_self_link(em, ln, bi, batch_set, I):
        """one B link of the self-collision half (appends to em.lines); groups whose A link is in batch_set"""
        groups = [sg for sg in self_by_b.get(ln, []) if sg["a"] in batch_set]
        if not groups:
            return
        g_env = env_by_link[ln]
        fine = g_env["fine"]
        bb = g_env["bound"]
        chunks = [fine[i:i + CHUNK] for i in range(0, len(fine), CHUNK)]
        em.lines.append(f"{I}// ---- B = {ln}: {len(fine)} spheres, {len(groups)} group(s)")
        for sg in groups:
            em.need(sorted({p[0] for p in sg["pairs"]}) + [sg["bound_a"]])
        em.need([bb] + fine)
        gate_names = []
        for gi, sg in enumerate(groups):
            ba = sg["bound_a"]
            rs = f32(f32(radii[ba]) + f32(radii[bb]))
            gn = f"gate_{bi}_{links.index(ln)}_{gi}"
            gate_names.append(gn)
            tb = table_bit.get(group_index[id(sg)])
            tbit = f" && ((tb{tb[0]} >> {tb[1]}) & 1u) != 0u" if tb else ""
            em.lines.append(
                f"{I}const bool {gn} = vmv::group_any<G>(vmv::neg(vmv::sql2_3({em.coord(ba, 0)}, {em.coord(ba, 1)}, "
                f"{em.coord(ba, 2)}, {em.coord(bb, 0)}, {em.coord(bb, 1)}, {em.coord(bb, 2)}) - {flit(float(f32(rs * rs)))}){tbit})"
                f" && !bad{' && VMV_ABLATE_SELF != 8' if id(sg) in dense_ids else ''};  // {sg['a']} vs. {ln}")
        sparse = [gi for gi, sg in enumerate(groups) if id(sg) not in dense_ids]
        dense = [gi for gi, sg in enumerate(groups) if id(sg) in dense_ids]
        em.lines.append(f"{I}if (VMV_ABLATE_SELF != 2 && vmv::wave_any(" + " || ".join(gate_names) + "))")
        em.lines.append(f"{I}{{")
        em.lines.append(f"{I}    flags[lane] = 0u;")
        J = I + "    "

        def a_coord(s):
            cs = []
            for k in range(3):
                kind, v = m["outputs"][s][k]
                cs.append(f"__shfl({em.prefix}{v}, (int) src)" if kind == "op" else flit(v))
            return cs

        def b_fetch(K, off):
            em.lines.append(f"{K}const vmv::lds_cptr p = wave_slab + 3 * t * vmv::kRow + src;")
            em.lines.append(f"{K}const float bx = p[0], by = p[vmv::kRow], bz = p[2 * vmv::kRow];")
            em.lines.append(f"{K}const float rb = radii[{off} + t];")

        def pair_tests(K, a_sph, acc, guard=None):
            for ai, s in enumerate(a_sph):
                cs = a_coord(s)
                if guard:
                    em.lines.append(f"{K}if (vmv::wave_any((ma & {1 << ai}u) != 0u))")
                em.lines.append(f"{K}{{")
                em.lines.append(f"{K}    const float rs = {flit(radii[s])} + rb;")
                em.lines.append(f"{K}    {acc} |= vmv::neg(vmv::sql2_3({cs[0]}, {cs[1]}, {cs[2]}, bx, by, bz) - rs * rs);")
                em.lines.append(f"{K}}}")

        # dense groups: the owners' A-side candidate words (A spheres that reach B's bounding sphere)
        for gi in dense:
            sg = groups[gi]
            a_sph = sorted({p[0] for p in sg["pairs"]})
            em.lines.append(f"{J}unsigned cand_{gate_names[gi]} = 0u;")
            em.lines.append(f"{J}if (vmv::wave_any({gate_names[gi]}))")
            em.lines.append(f"{J}{{")
            for ai, s in enumerate(a_sph):
                rs = float(f32(radii[s])) + float(f32(radii[bb])) + SELF_MARGIN
                em.lines.append(
                    f"{J}    cand_{gate_names[gi]} |= vmv::neg(vmv::sql2_3({em.coord(s, 0)}, {em.coord(s, 1)}, {em.coord(s, 2)}, "
                    f"{em.coord(bb, 0)}, {em.coord(bb, 1)}, {em.coord(bb, 2)}) - {flit(float(f32(rs * rs)))}) ? {1 << ai}u : 0u;")
            em.lines.append(f"{J}}}")
        done = 0
        for ci, ch in enumerate(chunks):
            off = radii_off[ln] + 1 + done
            for si, s in enumerate(ch):
                for k in range(3):
                    em.lines.append(f"{J}slab[{3 * si + k} * vmv::kRow] = {em.coord(s, k)};")
            em.lines.append(f"{J}vmv::wave_lds_sync();")
            # the merged entry list holds at most SPARSE_BATCH * 64 (lane, group) entries
            for sb in range(0, len(sparse), SPARSE_BATCH):
                batch_groups = sparse[sb:sb + SPARSE_BATCH]
                em.lines.append(f"{J}if (VMV_ABLATE_SELF != 4 && vmv::wave_any(" + " || ".join(gate_names[gi] for gi in batch_groups) + f"))  // sparse groups, chunk {ci}")
                em.lines.append(f"{J}{{")
                K = J + "    "
                # entries are appended group by group, so entry counts are also item boundaries: the items are
                # dealt GROUP-MAJOR (all items of group 0, then group 1, ...) and a round of 64 items touches one or
                # two groups instead of all of them - only those run their A loops
                em.lines.append(f"{K}int k = 0;")
                for li, gi in enumerate(batch_groups):
                    em.lines.append(f"{K}const int kk{li} = k;")
                    em.lines.append(f"{K}k = vmv::deal_append(list2, k, {gate_names[gi]}, {li}u << 6);  // {groups[gi]['a']}")
                em.lines.append(f"{K}const int kk{len(batch_groups)} = k;")
                em.lines.append(f"{K}vmv::wave_lds_sync();")
                em.lines.append(f"{K}const int items = k * {len(ch)};")
                for li in range(len(batch_groups)):
                    em.lines.append(f"{K}const int n{li} = kk{li + 1} - kk{li};")
                    em.lines.append(f"{K}const float inv{li} = 1.0f / (float) (n{li} > 0 ? n{li} : 1);")
                em.lines.append(f"{K}for (int base = 0; base < items; base += vmv::kWave)")
                em.lines.append(f"{K}{{")
                K2 = K + "    "
                em.lines.append(f"{K2}const int i = base + (int) lane;")
                em.lines.append(f"{K2}const bool act = i < items;")
                em.lines.append(f"{K2}int first = 0, n = n0;")
                em.lines.append(f"{K2}float inv = inv0;")
                em.lines.append(f"{K2}unsigned grp = 0u;")
                for li in range(1, len(batch_groups)):
                    em.lines.append(f"{K2}if (i >= kk{li} * {len(ch)}) first = kk{li}, n = n{li}, inv = inv{li}, grp = {li}u;")
                em.lines.append(f"{K2}const int local = i - first * {len(ch)};")
                em.lines.append(f"{K2}const int t = act ? (int) (((float) local + 0.5f) * inv) : 0;")
                em.lines.append(f"{K2}const int j = act ? first + (local - t * n) : 0;")
                em.lines.append(f"{K2}const unsigned e = list2[j];")
                em.lines.append(f"{K2}const unsigned src = e & 63u;")
                em.lines.append(f"{K2}grp = act ? grp : ~0u;")
                b_fetch(K2, off)
                em.lines.append(f"{K2}bool h = false;")
                for li, gi in enumerate(batch_groups):
                    sg = groups[gi]
                    a_sph = sorted({p[0] for p in sg["pairs"]})
                    b_sph = sorted({p[1] for p in sg["pairs"]})
                    assert b_sph == fine and sg["pairs"] == [[s, t] for s in a_sph for t in b_sph]
                    em.lines.append(f"{K2}if (vmv::wave_any(grp == {li}u))  // {sg['a']} vs. {ln}")
                    em.lines.append(f"{K2}{{")
                    em.lines.append(f"{K2}    bool hg = false;")
                    pair_tests(K2 + "    ", a_sph, "hg")
                    em.lines.append(f"{K2}    h |= hg && (grp == {li}u);")
                    em.lines.append(f"{K2}}}")
                em.lines.append(f"{K2}if (h) flags[src] = 1u;")
                em.lines.append(f"{K}}}")
                em.lines.append(f"{K}vmv::wave_lds_sync();")
                em.lines.append(f"{J}}}")
            for gi in dense:
                sg = groups[gi]
                gn = gate_names[gi]
                ba = sg["bound_a"]
                a_sph = sorted({p[0] for p in sg["pairs"]})
                b_sph = sorted({p[1] for p in sg["pairs"]})
                assert b_sph == fine and sg["pairs"] == [[s, t] for s in a_sph for t in b_sph]
                em.lines.append(f"{J}if (vmv::wave_any({gn}))  // dense: {sg['a']} vs. {ln}, chunk {ci}")
                em.lines.append(f"{J}{{")
                K = J + "    "
                em.lines.append(f"{K}cand[lane] = cand_{gn};")
                em.lines.append(f"{K}int n2 = 0;")
                for si, s in enumerate(ch):
                    rs = float(f32(radii[ba])) + SELF_MARGIN + float(f32(radii[s]))
                    em.lines.append(
                        f"{K}n2 = vmv::deal_append(list2, n2, {gn} && vmv::neg(vmv::sql2_3({em.coord(ba, 0)}, {em.coord(ba, 1)}, "
                        f"{em.coord(ba, 2)}, {em.coord(s, 0)}, {em.coord(s, 1)}, {em.coord(s, 2)}) - {flit(float(f32(rs * rs)))}), {si}u << 6);")
                em.lines.append(f"{K}vmv::wave_lds_sync();")
                em.lines.append(f"{K}for (int base = 0; base < n2 && VMV_ABLATE_SELF != 1; base += vmv::kWave)")
                em.lines.append(f"{K}{{")
                K2 = K + "    "
                em.lines.append(f"{K2}const int i = base + (int) lane;")
                em.lines.append(f"{K2}const bool act = i < n2;")
                em.lines.append(f"{K2}const unsigned e = list2[act ? i : 0];")
                em.lines.append(f"{K2}const unsigned src = e & 63u;")
                em.lines.append(f"{K2}const int t = (int) (e >> 6);")
                b_fetch(K2, off)
                em.lines.append(f"{K2}const unsigned ma = act ? cand[src] : 0u;")
                em.lines.append(f"{K2}bool h = false;")
                pair_tests(K2, a_sph, "h", guard=True)
                em.lines.append(f"{K2}if (h && act) flags[src] = 1u;")
                em.lines.append(f"{K}}}")
                em.lines.append(f"{K}vmv::wave_lds_sync();")
                em.lines.append(f"{J}}}")
            done += len(ch)
        em.lines.append(f"{I}    bad |= vmv::group_any<G>(flags[lane] != 0u);")
        em.lines.append(f"{I}}}")

    for bi, batch in enumerate(batches):
        batch_set = set(batch)
        L.append(f"        {{  // pass {bi}: A in {{{', '.join(batch)}}}")
        qn = "q"
        if len(batches) > 1:
            # an opaque copy of the configuration keeps the compiler from merging the passes' FK back together
            L.append("            float qp[kDim];")
            L.append("#pragma unroll")
            L.append("            for (int j = 0; j < kDim; ++j)")
            L.append("            {")
            L.append("                qp[j] = q[j];")
            L.append('                asm volatile("" : "+v"(qp[j]));')
            L.append("            }")
            qn = "qp"
        em = Emitter(m, qname=qn, prefix=f"p{bi}_", indent="            ")
        for ln in links:
            emit_self_link(em, ln, bi, batch_set, "            ")
        L += em.lines
        L.append("        }")
    L.append("        return bad;")
    L.append("    }")
    L.append(f"    constexpr int kSelfPasses = {len(batches)};")
    L.append("")

    # ---- both halves in one walk of the chain (one FK per configuration) -------------------------------------------
    # Only for robots whose self-collision half needs a single pass (all A-side spheres fit the registers).
    fused = len(batches) == 1
    L.append(f"    constexpr bool kHasFused = {'true' if fused else 'false'};")
    if fused:
        L.append("    // Robot::fkcc<rake>, both halves along ONE walk of the chain: a link's FK ops, its environment group")
        L.append("    // (gate + fine phase through the slab), then the self-collision groups whose B side it is.  The two")
        L.append("    // halves use the per-wave LDS region one after the other (the environment half's lists are folded")
        L.append("    // into `bad` before the self-collision half stages its chunk).")
        L.append("    template <int G, int V>")
        L.append("    __device__ __forceinline__ bool")
        L.append("    fkcc_fused(const vmv::EnvView &E, const float (&q)[kDim], vmv::lds_ptr slab, const bool skip)")
        L.append("    {")
        L.append("        bool bad = skip || (E.dev->static_hit != 0u);")
        L.append("        const unsigned lane = __lane_id();")
        L.append("        const vmv::lds_ptr scratch = slab - lane + kSlabSpheres * 3 * vmv::kRow;")
        L.append("        const vmv::lds_cptr wave_slab = vmv::uniform((vmv::lds_cptr) (slab - lane));")
        L.append("        const vmv::lds_cptr radii = vmv::uniform(E.radii);")
        L.append("        vmv::lds_u32 *const list = (vmv::lds_u32 *) (slab - lane + kSelfSlabSpheres * 3 * vmv::kRow);")
        L.append("        vmv::lds_u32 *const flags = list + vmv::kWave;")
        L.append("        vmv::lds_u32 *const cand = list + 2 * vmv::kWave + 4;")
        L.append("        vmv::lds_u32 *const list2 = cand + vmv::kWave;")
        for ti in range(len(tables)):
            L.append(f"        const unsigned tb{ti} = (VMV_ABLATE_SELF == 16) ? 0xffu : self_table{ti}(q);")
        em = Emitter(m, prefix="t", indent="        ")
        bset = set(batches[0])
        for ln in links:
            emit_env_link(em, ln, lazy=False, no_skip=True)
            emit_self_link(em, ln, 0, bset, "        ")
        L += em.lines
        L.append("        (void) list;")
        L.append("        return bad;")
        L.append("    }")
        L.append("")

    # ---- end-effector frame + attachments -----------------------------------------------------------------------
    ee_m = dict(ops=m["ee_ops"], outputs=[m["ee_outputs"][3 * i:3 * i + 3] for i in range(4)])
    L.append("    // Robot::eefk / the frame Robot::fkcc_attach poses attachments at: f[0..2] translation, f[3..11] rotation,")
    L.append("    // column-major (vector/math.hh:40-51).  Own op tape (model key ee_ops), bit-exact vs the reference's outputs.")
    L.append("    __device__ __forceinline__ void ee_frame(const float (&q)[kDim], float (&f)[12])")
    L.append("    {")
    em = Emitter(ee_m, prefix="e")
    em.need([0, 1, 2, 3])
    L += em.lines
    for i in range(4):
        for k in range(3):
            L.append(f"        f[{3 * i + k}] = {em.coord(i, k)};")
    L.append("    }")
    L.append("")
    L.append("    // The attachment part of Robot::fkcc_attach (robots/%s.hh; collision/attachments.hh, validity.hh:258-303):" % n)
    L.append("    // pose the attached spheres at the end-effector frame, test them against the environment and against the")
    L.append("    // links of the reference's \"Attachment vs. <link>\" blocks.  true = this rake collides.  The two frame")
    L.append("    // products are Eigen expressions in the reference (parity unpinned); restated as ((a0 b0 + a1 b1) + a2 b2) [+ t].")
    L.append("    template <int G>")
    L.append("    __device__ __forceinline__ bool")
    L.append("    fkcc_attach(const vmv::EnvView &E, const float (&q)[kDim], vmv::lds_ptr slab, const bool skip)")
    L.append("    {")
    L.append("        bool bad = skip;")
    L.append("        const vmv::lds_ptr scratch = slab - __lane_id() + kSlabSpheres * 3 * vmv::kRow;")
    L.append("        const vmv::env_cptr Dp = E.dev;")
    L.append("        const unsigned na = Dp->n_attach;")
    L.append("        float f[12];")
    L.append("        ee_frame(q, f);")
    L.append("        float Rn[3][3], tn[3];")
    L.append("#pragma unroll")
    L.append("        for (int i = 0; i < 3; ++i)")
    L.append("        {")
    L.append("            const float r0 = f[3 + i], r1 = f[6 + i], r2 = f[9 + i];")
    L.append("#pragma unroll")
    L.append("            for (int j = 0; j < 3; ++j)")
    L.append("                Rn[i][j] = ((r0 * Dp->attach_tf[j]) + (r1 * Dp->attach_tf[4 + j])) + (r2 * Dp->attach_tf[8 + j]);")
    L.append("            tn[i] = (((r0 * Dp->attach_tf[3]) + (r1 * Dp->attach_tf[7])) + (r2 * Dp->attach_tf[11])) + f[i];")
    L.append("        }")
    L.append("        const vmv::cf_cptr sph = (vmv::cf_cptr) Dp->attach_spheres;  // wave-uniform indices: scalar loads")
    L.append("        auto posed = [&](const unsigned s, float &x, float &y, float &z, float &r)")
    L.append("        {")
    L.append("            const float sx = sph[4 * s], sy = sph[4 * s + 1], sz = sph[4 * s + 2];")
    L.append("            r = sph[4 * s + 3];")
    L.append("            x = (((Rn[0][0] * sx) + (Rn[0][1] * sy)) + (Rn[0][2] * sz)) + tn[0];")
    L.append("            y = (((Rn[1][0] * sx) + (Rn[1][1] * sy)) + (Rn[1][2] * sz)) + tn[1];")
    L.append("            z = (((Rn[2][0] * sx) + (Rn[2][1] * sy)) + (Rn[2][2] * sz)) + tn[2];")
    L.append("        };")
    L.append("        // attachment vs. environment: chunks of posed spheres through the slab, re-dealt like a link's fine spheres")
    att_chunk = slab_spheres  # the slab holds min(CHUNK, largest link) spheres
    L.append(f"        for (unsigned base = 0; base < na; base += {att_chunk}u)")
    L.append("        {")
    L.append(f"            const unsigned cnt = (na - base < {att_chunk}u) ? na - base : {att_chunk}u;")
    L.append("            for (unsigned s = 0; s < cnt; ++s)")
    L.append("            {")
    L.append("                float x, y, z, r;")
    L.append("                posed(base + s, x, y, z, r);")
    L.append("                slab[(3 * s + 0) * vmv::kRow] = x;")
    L.append("                slab[(3 * s + 1) * vmv::kRow] = y;")
    L.append("                slab[(3 * s + 2) * vmv::kRow] = z;")
    L.append("            }")
    L.append("            const bool act = vmv::env_list_active<G>(scratch, !bad);")
    L.append("            if (vmv::wave_any(act))")
    L.append("            {")
    L.append("                vmv::env_fine<G, Tab>(E, slab, scratch, (int) cnt, kNRadii + (int) base, 1);")
    L.append("                bad |= act && vmv::group_any<G>(vmv::env_flag(scratch));")
    L.append("            }")
    L.append("        }")
    L.append("        // attachment vs. robot")
    em = Emitter(m)
    for ln in m["attach_links"]:
        g = env_by_link[ln]
        b = g["bound"]
        em.lines.append(f"        // ---- Attachment vs. {ln}")
        em.need([b] + g["fine"])
        em.lines.append("        {")
        em.lines.append("            bool gate = false;")
        em.lines.append("            for (unsigned s = 0; s < na; ++s)")
        em.lines.append("            {")
        em.lines.append("                float x, y, z, r;")
        em.lines.append("                posed(s, x, y, z, r);")
        em.lines.append(f"                const float rs = {flit(radii[b])} + r;")
        em.lines.append(f"                gate |= vmv::neg(vmv::sql2_3({em.coord(b, 0)}, {em.coord(b, 1)}, {em.coord(b, 2)}, x, y, z) - rs * rs);")
        em.lines.append("            }")
        em.lines.append("            gate = vmv::group_any<G>(gate) && !bad;")
        em.lines.append("            if (vmv::wave_any(gate))")
        em.lines.append("            {")
        em.lines.append("                bool h = false;")
        em.lines.append("                for (unsigned s = 0; s < na; ++s)")
        em.lines.append("                {")
        em.lines.append("                    float x, y, z, r;")
        em.lines.append("                    posed(s, x, y, z, r);")
        for fs in g["fine"]:
            em.lines.append("                    {")
            em.lines.append(f"                        const float rs = {flit(radii[fs])} + r;")
            em.lines.append(f"                        h |= vmv::neg(vmv::sql2_3({em.coord(fs, 0)}, {em.coord(fs, 1)}, {em.coord(fs, 2)}, x, y, z) - rs * rs);")
            em.lines.append("                    }")
        em.lines.append("                }")
        em.lines.append("                bad |= gate && vmv::group_any<G>(h);")
        em.lines.append("            }")
        em.lines.append("        }")
    L += em.lines
    L.append("        return bad;")
    L.append("    }")
    L.append("")

    # ---- fine pair table (contact report) ---------------------------------------------------------------------
    pairs = [p for sg in m["self_groups"] for p in sg["pairs"]]
    L.append(f"    constexpr int kNSelfPairs = {len(pairs)};  // Robot::fkcc_debug tests every fine pair, without the gates")
    L.append(f"    __constant__ unsigned short kSelfPairs[{max(len(pairs), 1)}][2] = {{" +
             ", ".join("{%d, %d}" % (a, b) for a, b in pairs) + "};")
    L.append("")

    # ---- sphere_fk ----------------------------------------------------------------------------------------
    L.append("    // Robot::sphere_fk (reference robots/%s.hh): out[s] = (x, y, z, r) of the fine spheres." % n)
    L.append("    __device__ __forceinline__ void sphere_fk(const float (&q)[kDim], float4 *out)")
    L.append("    {")
    em = Emitter(m)
    em.need(list(range(m["n_spheres"])))
    L += em.lines
    for s in range(m["n_spheres"]):
        L.append(f"        out[{s}] = make_float4({em.coord(s, 0)}, {em.coord(s, 1)}, {em.coord(s, 2)}, {flit(radii[s])});")
    L.append("    }")
    L.append("}  // namespace " + n)
    L.append("")
    L.append(f"struct {n}_traits")
    L.append("{")
    L.append(f"    static constexpr int kDim = {n}::kDim;")
    L.append(f"    static constexpr int kNSpheres = {n}::kNSpheres;")
    L.append(f"    static constexpr int kResolution = {n}::kResolution;")
    L.append(f"    static constexpr int kSlabSpheres = {n}::kSlabSpheres;")
    L.append(f"    static constexpr int kSelfSlabSpheres = {n}::kSelfSlabSpheres;")
    L.append("#ifdef VMV_ENV_BLOCKS  // tuning knob (tools/build_variant.py)")
    L.append("    static constexpr int kEnvBlocks = VMV_ENV_BLOCKS;")
    L.append("#else")
    L.append(f"    static constexpr int kEnvBlocks = {ENV_BLOCKS.get(n, 4)};  // workgroups per CU the environment kernel is compiled for")
    L.append("#endif")
    L.append(f"    static constexpr int kNRadii = {n}::kNRadii;")
    L.append(f"    static constexpr int kNStaticLinks = {n}::kNStaticLinks;")
    L.append(f"    static constexpr int kNSelfPairs = {n}::kNSelfPairs;")
    L.append("    static __device__ __forceinline__ bool static_env_hit(const vmv::EnvView &E)")
    L.append("    {")
    L.append(f"        return {n}::static_env_hit(E);")
    L.append("    }")
    L.append(f"    static constexpr int kSelfBlocks = {SELF_BLOCKS.get(n, 2)};  // workgroups per CU the self-collision kernel is compiled for")
    L.append(f"    static constexpr int kMotionSelfBlocks = {MOTION_SELF_BLOCKS.get(n, SELF_BLOCKS.get(n, 2))};  // ... and its (edge, rake) task kernel")
    L.append("    template <int G, int V>")
    L.append("    static __device__ __forceinline__ bool")
    L.append("    fkcc_env(const vmv::EnvView &E, const float (&q)[kDim], vmv::lds_ptr slab, const bool skip)")
    L.append("    {")
    L.append(f"        return {n}::fkcc_env<G, V>(E, q, slab, skip);")
    L.append("    }")
    L.append(f"    static constexpr bool kHasFused = {n}::kHasFused;")
    L.append(f"    static constexpr int kFusedBlocks = {FUSED_BLOCKS.get(n, 4)};  // workgroups per CU the fused kernel is compiled for")
    L.append("    template <int G, int V>")
    L.append("    static __device__ __forceinline__ bool")
    L.append("    fkcc_fused(const vmv::EnvView &E, const float (&q)[kDim], vmv::lds_ptr slab, const bool skip)")
    L.append("    {")
    if len(batches) == 1:
        L.append(f"        return {n}::fkcc_fused<G, V>(E, q, slab, skip);")
    else:
        L.append("        return skip;  // not built for this robot (several self-collision passes): never launched")
    L.append("    }")
    L.append("    template <int G>")
    L.append("    static __device__ __forceinline__ bool")
    L.append("    fkcc_self(const float (&q)[kDim], vmv::lds_ptr slab, const vmv::lds_cptr radii, const bool skip)")
    L.append("    {")
    L.append(f"        return {n}::fkcc_self<G>(q, slab, radii, skip);")
    L.append("    }")
    L.append("    static __device__ __forceinline__ void sphere_fk(const float (&q)[kDim], float4 *out)")
    L.append("    {")
    L.append(f"        {n}::sphere_fk(q, out);")
    L.append("    }")
    L.append("    static __device__ __forceinline__ void ee_frame(const float (&q)[kDim], float (&f)[12])")
    L.append("    {")
    L.append(f"        {n}::ee_frame(q, f);")
    L.append("    }")
    L.append("    template <int G>")
    L.append("    static __device__ __forceinline__ bool")
    L.append("    fkcc_attach(const vmv::EnvView &E, const float (&q)[kDim], vmv::lds_ptr slab, const bool skip)")
    L.append("    {")
    L.append(f"        return {n}::fkcc_attach<G>(E, q, slab, skip);")
    L.append("    }")
    L.append("};")
    L.append("")
    return "\n".join(L)


def main(models):
    d = os.path.join(ROOT, "vamp_mvt_amd", "csrc", "gen")
    os.makedirs(d, exist_ok=True)
    for m in models:
        n = m["name"]
        with open(os.path.join(d, f"{n}_dev.inc"), "w") as f:
            f.write("\n".join(["// GENERATED by tools/gen_hip.py from vamp_mvt_amd/robots/%s.json - do not edit." % n,
                               "#pragma once", "", "namespace vmv", "{", emit_robot(m), "}  // namespace vmv"]) + "\n")
        with open(os.path.join(d, f"tu_{n}.hip"), "w") as f:
            f.write("\n".join([
                "// GENERATED by tools/gen_hip.py - do not edit.  One translation unit per robot.",
                f"#define VMV_ROBOT_NS {n}",
                f"#define VMV_ROBOT_LAUNCH k{n.capitalize()}Launchers",
                '#include "../vmv_common.h"',
                f'#include "{n}_dev.inc"',
                '#include "../vmv_robot_tu.inc"', ""]))
    # host-side robot table
    host = ["// GENERATED by tools/gen_hip.py - do not edit.", "#pragma once", ""]
    for m in models:
        pairs = [p for sg in m["self_groups"] for p in sg["pairs"]]
        host.append(f"static const uint16_t kSelfPairs_{m['name']}[{max(len(pairs), 1)}][2] = {{" +
                    ", ".join("{%d, %d}" % (a, b) for a, b in pairs) + "};")
    # reach certificates of the first links (link_samples): sample centres + slack per link
    for m in models:
        reach = link_samples(m)
        for gi, (C, slack) in reach.items():
            host.append(f"static const float kReachSamples_{m['name']}_{gi}[{len(C)}][3] = {{" +
                        ", ".join("{%s, %s, %s}" % (flit(c[0]), flit(c[1]), flit(c[2])) for c in C) + "};")
        host.append(f"static const vmv_link_reach kReach_{m['name']}[{max(len(reach), 1)}] = {{" +
                    (", ".join(f"{{{gi}, {len(C)}, {flit(slack)}, {flit(m['radii'][m['env_groups'][gi]['bound']])}, kReachSamples_{m['name']}_{gi}}}"
                               for gi, (C, slack) in reach.items()) or "{0, 0, 0.0f, 0.0f, nullptr}") + "};")
    host.append("static const vmv_robot_info kRobots[] = {")
    for m in models:
        lo = ", ".join(flit(v) for v in m["lower"] + [0.0] * (16 - m["dimension"]))
        sp = ", ".join(flit(v) for v in m["span"] + [0.0] * (16 - m["dimension"]))
        ds = ", ".join(flit(v) for v in m["descale"] + [0.0] * (16 - m["dimension"]))
        jn = ", ".join('"%s"' % j for j in m["joint_names"])
        max_bound = max(m["radii"][m["n_spheres"]:] + [grp["radius"] for grp in merged_groups(m)])
        gr = ", ".join(flit(v) for v in grid_classes(m)[0])
        host.append(f'    {{"{m["name"]}", {m["dimension"]}, {m["n_spheres"]}, {m["resolution"]}, '
                    f'{flit(m["min_radius"])}, {flit(m["max_radius"])}, {flit(max_bound)}, {{{gr}}}, {{{lo}}}, {{{sp}}}, {{{ds}}}, '
                    f'"{m["end_effector"]}", {{{jn}}}, {sum(len(sg["pairs"]) for sg in m["self_groups"])}, '
                    f'kSelfPairs_{m["name"]}, {len(link_samples(m))}, kReach_{m["name"]}}},')
    host.append("};")
    host.append(f"static const int kNumRobots = {len(models)};")
    with open(os.path.join(d, "robots_host.inc"), "w") as f:
        f.write("\n".join(host) + "\n")
    print("wrote vamp_mvt_amd/csrc/gen/{<robot>_dev.inc, tu_<robot>.hip, robots_host.inc}")


if __name__ == "__main__":
    import gen_code
    main([gen_code.load(n) for n in gen_code.ROBOTS])
