#!/usr/bin/env python3
"""In-container checker: evaluate the reference's generated FK statements as data.

TEST/VALIDATION TOOLING ONLY.  Runs only where /root/reference exists (never on
the GPU box, never from the product path).  The reference's robot headers
(`src/impl/vamp/robots/{panda,ur5,fetch,baxter}.hh`) cannot be compiled in this
image (they pull Eigen) so, to obtain golden FK vectors from the
reference itself, this script *reads the straight-line statements of
`Robot::fkcc` / `Robot::sphere_fk` as text* (`v[i] = ...;` / `y[i] = ...;`) and
evaluates them in fp32 with exactly the C++ semantics of the reference's
operator overloads (vector/interface.hh:993-1017: a `double` literal next to a
vector is narrowed to float once; double∘double sub-expressions fold in double
first; everything else is IEEE fp32, left-to-right as C++ parses it) and with
the reference's own sin/cos (oracle/_ref/libref_vector.so, built from
vector/avx.hh:455-548 and vector/interface.hh:447-458 where they lie).

Nothing of the reference text is written anywhere: outputs are numeric vectors
(inputs + expected sphere centres) for tests/golden/, and structural facts
(sphere radii, group lists) used to cross-check tools/robot_trace.py.
"""
from __future__ import annotations

import ctypes
import os
import re
import sys
from dataclasses import dataclass

import numpy as np

REF = os.environ.get("VAMP_REFERENCE", "/root/reference")
ROBOT_HH = REF + "/src/impl/vamp/robots/{name}.hh"
_HERE = os.path.dirname(os.path.abspath(__file__))
_REFLIB = os.path.join(_HERE, "..", "oracle", "_ref", "libref_vector.so")

_fp = ctypes.POINTER(ctypes.c_float)


class RefVector:
    """ctypes view of oracle/_ref/libref_vector.so (reference vector.hh)."""

    def __init__(self, path: str = _REFLIB):
        self.lib = ctypes.CDLL(path)
        for fn in ("ref_sin", "ref_cos", "ref_sqrt_approx"):
            getattr(self.lib, fn).argtypes = [_fp, _fp, ctypes.c_size_t]
            getattr(self.lib, fn).restype = None
        self.lib.ref_l2_norm.argtypes = [_fp, ctypes.c_size_t]
        self.lib.ref_l2_norm.restype = ctypes.c_float
        self.lib.ref_halton.argtypes = [ctypes.c_size_t, _fp, _fp, ctypes.c_size_t, _fp]
        self.lib.ref_halton.restype = ctypes.c_int

    def _map(self, fn, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        getattr(self.lib, fn)(x.ctypes.data_as(_fp), out.ctypes.data_as(_fp), x.size)
        return out

    def sin(self, x):
        return self._map("ref_sin", x)

    def cos(self, x):
        return self._map("ref_cos", x)

    def sqrt_approx(self, x):
        return self._map("ref_sqrt_approx", x)

    def l2_norm(self, q):
        q = np.ascontiguousarray(q, np.float32)
        return float(self.lib.ref_l2_norm(q.ctypes.data_as(_fp), q.size))

    def halton(self, s_m, s_a, count):
        s_m = np.ascontiguousarray(s_m, np.float32)
        s_a = np.ascontiguousarray(s_a, np.float32)
        out = np.empty((count, s_m.size), np.float32)
        rc = self.lib.ref_halton(s_m.size, s_m.ctypes.data_as(_fp), s_a.ctypes.data_as(_fp), count,
                                 out.ctypes.data_as(_fp))
        if rc != 0:
            raise ValueError("unsupported dimension")
        return out


# ----------------------------------------------------------------------------
# expression parsing (C++ precedence: unary -, then * , then + -; left assoc)
# ----------------------------------------------------------------------------
_TOK = re.compile(r"\s*(?:(\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)|([A-Za-z_]\w*)|(.))")


def _tokenize(s):
    out = []
    pos = 0
    while pos < len(s):
        m = _TOK.match(s, pos)
        if not m:
            break
        pos = m.end()
        if m.group(1) is not None:
            out.append(("num", m.group(1)))
        elif m.group(2) is not None:
            out.append(("id", m.group(2)))
        elif m.group(3) is not None and not m.group(3).isspace():
            out.append(("op", m.group(3)))
    return out


class _Parser:
    def __init__(self, toks):
        self.t = toks
        self.i = 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else (None, None)

    def eat(self, kind=None, val=None):
        k, v = self.peek()
        if (kind and k != kind) or (val and v != val):
            raise SyntaxError(f"expected {kind} {val}, got {k} {v}")
        self.i += 1
        return v

    def expr(self):
        node = self.term()
        while self.peek() in (("op", "+"), ("op", "-")):
            op = self.eat()
            rhs = self.term()
            node = (op, node, rhs)
        return node

    def term(self):
        node = self.unary()
        while self.peek() == ("op", "*"):
            self.eat()
            rhs = self.unary()
            node = ("*", node, rhs)
        return node

    def unary(self):
        if self.peek() == ("op", "-"):
            self.eat()
            return ("neg", self.unary())
        return self.atom()

    def atom(self):
        k, v = self.peek()
        if k == "num":
            self.eat()
            return ("const", float(v))
        if k == "op" and v == "(":
            self.eat()
            n = self.expr()
            self.eat("op", ")")
            return n
        if k == "id":
            self.eat()
            if v in ("sin", "cos"):
                self.eat("op", "(")
                a = self.expr()
                self.eat("op", ")")
                return (v, a)
            if v in ("v", "y", "x"):
                self.eat("op", "[")
                idx = int(self.eat("num"))
                self.eat("op", "]")
                return ("ref", v, idx)
        raise SyntaxError(f"unexpected token {k} {v}")


@dataclass
class RefProgram:
    name: str
    function: str
    statements: list  # (kind 'v'|'y', index, ast)
    n_v: int
    n_y: int
    env_groups: list  # [(link, bound_y_index, [fine y indices])]
    self_groups: list  # [(link_a, link_b, (ya, yb), [(ya, yb), ...])]


def load_program(name: str, function: str = "fkcc") -> RefProgram:
    """Parse `function` ('fkcc', 'fkcc_attach' or 'sphere_fk') of robots/<name>.hh."""
    path = ROBOT_HH.format(name=name)
    with open(path) as f:
        lines = f.read().split("\n")
    if function == "fkcc":
        start = next(i for i, l in enumerate(lines) if "inline static bool fkcc(" in l)
    elif function == "sphere_fk":
        start = next(i for i, l in enumerate(lines) if "inline static void sphere_fk(" in l)
    elif function == "fkcc_attach":
        start = next(i for i, l in enumerate(lines) if "inline static bool fkcc_attach(" in l)
    else:
        raise ValueError(function)
    # body runs until the next top-level template/function declaration
    end = next(i for i in range(start + 1, len(lines)) if re.match(r"\s+template <std::size_t rake>", lines[i])
               or re.match(r"\s+(inline )?static .*\(", lines[i]) and i > start + 3)
    body = lines[start:end]
    n_v = n_y = 0
    for l in body[:12]:
        m = re.search(r"std::array<FloatVector<rake, 1>, (\d+)> v;", l)
        if m:
            n_v = int(m.group(1))
        m = re.search(r"std::array<FloatVector<rake, 1>, (\d+)> y;", l)
        if m:
            n_y = int(m.group(1))
    text = "\n".join(body)
    # --- FK statements: everything shaped `v[i] = ...;` or `y[i] = ...;`
    statements = []
    for m in re.finditer(r"(?m)^\s+([vy])\[(\d+)\] =\s*([^;]*);", text):
        kind, idx, rhs = m.group(1), int(m.group(2)), m.group(3)
        ast = _Parser(_tokenize(rhs)).expr()
        statements.append((kind, idx, ast))
    # --- check structure
    env_groups, self_groups = [], []
    if function == "fkcc":
        cur = None
        for l in body:
            m = re.match(r"\s+// (\S+) vs\. (\S+)\s*$", l)
            if m:
                cur = ("self", m.group(1), m.group(2), [])
                self_groups.append(cur)
                continue
            m = re.match(r"\s+// (\S+)\s*$", l)
            if m and m.group(1) not in ("environment", "robot", "dependent"):
                cur = ("env", m.group(1), [])
                env_groups.append(cur)
                continue
        # collect y-index tuples in order of appearance per group
        joined = re.sub(r"\s+", " ", text)
        # split by group comment markers
        parts = re.split(r"// (\S+ vs\. \S+|\S+) (?=if )", joined)
        # parts: [pre, marker, chunk, marker, chunk ...]
        env_out, self_out = [], []
        for k in range(1, len(parts) - 1, 2):
            marker, chunk = parts[k], parts[k + 1]
            if " vs. " in marker:
                a, b = marker.split(" vs. ")
                pairs = re.findall(
                    r"sphere_sphere_self_collision<decltype\(x\[0\]\)>\( y\[(\d+)\], y\[\d+\], y\[\d+\], y\[\d+\], y\[(\d+)\]",
                    chunk)
                pairs = [(int(p), int(q)) for p, q in pairs]
                self_out.append((a, b, pairs[0], pairs[1:]))
            else:
                idx = [int(p) for p in re.findall(r"sphere_environment_in_collision\(environment, y\[(\d+)\]", chunk)]
                if idx:
                    env_out.append((marker, idx[0], idx[1:]))
        env_groups, self_groups = env_out, self_out
    return RefProgram(name, function, statements, n_v, n_y, env_groups, self_groups)


class Evaluator:
    """fp32 evaluation of a RefProgram over N configurations at once."""

    def __init__(self, prog: RefProgram, refvec: RefVector | None = None):
        self.prog = prog
        self.rv = refvec or RefVector()

    def _ev(self, ast, env):
        k = ast[0]
        if k == "const":
            return ast[1]  # python float == C++ double
        if k == "ref":
            return env[ast[1]][ast[2]]
        if k == "neg":
            a = self._ev(ast[1], env)
            if isinstance(a, float):
                return -a
            return -a  # sign-bit flip, exact
        if k in ("sin", "cos"):
            a = self._ev(ast[1], env)
            return getattr(self.rv, k)(a)
        a = self._ev(ast[1], env)
        b = self._ev(ast[2], env)
        if isinstance(a, float) and isinstance(b, float):
            return {"+": a + b, "-": a - b, "*": a * b}[k]  # double arithmetic
        if isinstance(a, float):
            a = np.float32(a)
        if isinstance(b, float):
            b = np.float32(b)
        if k == "+":
            return a + b
        if k == "-":
            return a - b
        return a * b

    def run(self, q: np.ndarray):
        """q: [N, dim] float32 -> y: [n_y, N] float32."""
        q = np.ascontiguousarray(q, np.float32)
        n = q.shape[0]
        env = {"x": [np.ascontiguousarray(q[:, j]) for j in range(q.shape[1])],
               "v": [None] * max(self.prog.n_v, 1), "y": [None] * max(self.prog.n_y, 1)}
        for kind, idx, ast in self.prog.statements:
            val = self._ev(ast, env)
            if isinstance(val, float):
                val = np.full(n, np.float32(val), np.float32)
            env[kind][idx] = val.astype(np.float32, copy=False)
        ys = env["y"]
        out = np.zeros((len(ys), n), np.float32)
        for i, yv in enumerate(ys):
            if yv is not None:
                out[i] = yv
        return out


def robot_constants(name: str) -> dict:
    """dimension / n_spheres / resolution / radii / s_m / s_a of robots/<name>.hh (numeric data)."""
    with open(ROBOT_HH.format(name=name)) as f:
        text = f.read()
    out = {}
    for key in ("dimension", "n_spheres", "resolution"):
        out[key] = int(re.search(rf"static constexpr std::size_t {key} = (\d+);", text).group(1))
    for key in ("min_radius", "max_radius"):
        out[key] = float(re.search(rf"static constexpr float {key} = ([-\d.eE]+);", text).group(1))
    for key in ("s_m", "s_a", "d_m"):
        m = re.search(rf"std::array<float, dimension> {key}\{{([^}}]*)\}}", text)
        out[key] = [float(t) for t in m.group(1).replace("\n", " ").split(",")]
    m = re.search(r"joint_names = \{([^}]*)\}", text)
    out["joint_names"] = re.findall(r'"([^"]+)"', m.group(1))
    out["end_effector"] = re.search(r'end_effector = "([^"]+)"', text).group(1)
    out["space_measure"] = float(re.search(r"space_measure\(\)[^{]*\{\s*return ([-\d.eE]+);", text).group(1))
    return out


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "panda"
    p = load_program(name)
    print(name, "statements", len(p.statements), "n_v", p.n_v, "n_y", p.n_y, "env groups", len(p.env_groups),
          "self groups", len(p.self_groups), "fine pairs", sum(len(g[3]) for g in p.self_groups))
    print(robot_constants(name))
