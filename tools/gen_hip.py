#!/usr/bin/env python3
"""Robot model JSON -> HIP device code for gfx950 (vamp_mvt_amd/csrc/gen/robots_dev.inc).

Kernel-side shape of one robot:

  template <int G> bool fkcc(E, q[dim], slab)   per lane: "this rake is in collision"

  * the FK op tape is emitted link by link along the kinematic chain, each link's ops just before its checks,
    so only the chain state and the spheres of links that appear on the A side of a self-collision group stay
    live in VGPRs;
  * the current link's spheres (bounding first) are written to the wave's LDS slab (lane-contiguous) and
    the environment group loop / self-collision B-side loops index that slab with a wave-uniform index;
  * self-collision groups (A, B) run when B is the current link: A's spheres are named registers, unrolled;
    (r_a + r_b)^2 comes from a constant table computed here with the same two fp32 roundings.
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


def op_expr(op, a, b, sin="vmv::vsin", cos="vmv::vcos"):
    if op == "in":
        return f"q[{a}]"
    if op == "sin":
        return f"{sin}(t{a})"
    if op == "cos":
        return f"{cos}(t{a})"
    if op == "neg":
        return f"-t{a}"
    if op == "const":
        return flit(a)
    if op == "mul":
        return f"t{a} * t{b}"
    if op == "add":
        return f"t{a} + t{b}"
    if op == "sub":
        return f"t{a} - t{b}"
    if op == "cmul":
        return f"{flit(a)} * t{b}"
    if op == "cadd":
        return f"{flit(a)} + t{b}"
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
    def __init__(self, m):
        self.m = m
        self.ops = m["ops"]
        self.done = [False] * len(self.ops)
        self.lines = []

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
            self.lines.append(f"        const float t{i} = {op_expr(op, a, b)};")
            self.done[i] = True

    def coord(self, s, k):
        kind, v = self.m["outputs"][s][k]
        return f"t{v}" if kind == "op" else flit(v)


def f32(x):
    return np.float32(x)


def emit_robot(m):
    n = m["name"]
    L = []
    dim = m["dimension"]
    links = m["links"]
    radii = m["radii"]
    env_by_link = {g["link"]: g for g in m["env_groups"]}
    self_by_b = {}
    for g in m["self_groups"]:
        self_by_b.setdefault(g["b"], []).append(g)

    # constant tables
    radii_tab, radii_off = [], {}
    for ln in links:
        g = env_by_link[ln]
        radii_off[ln] = len(radii_tab)
        radii_tab += [radii[g["bound"]]] + [radii[s] for s in g["fine"]]
    rs2_tab, rs2_off = [], {}
    for gi, g in enumerate(m["self_groups"]):
        a_sph = sorted({p[0] for p in g["pairs"]})
        b_sph = sorted({p[1] for p in g["pairs"]})
        assert g["pairs"] == [[s, t] for s in a_sph for t in b_sph]
        assert b_sph == env_by_link[g["b"]]["fine"], "B side must be the link's fine spheres in slab order"
        rs2_off[gi] = len(rs2_tab)
        for t in b_sph:  # [t][a]
            for s in a_sph:
                rs = f32(f32(radii[s]) + f32(radii[t]))
                rs2_tab.append(float(f32(rs * rs)))
    max_group = max(len(g["fine"]) for g in m["env_groups"]) + 1

    L.append(f"namespace {n}")
    L.append("{")
    L.append(f"    constexpr int kDim = {dim};")
    L.append(f"    constexpr int kNSpheres = {m['n_spheres']};")
    L.append(f"    constexpr int kResolution = {m['resolution']};")
    L.append(f"    constexpr int kSlabSpheres = {max_group};  // largest link group incl. its bounding sphere")
    L.append(f"    __constant__ float kRadii[{len(radii_tab)}] = {{" + ", ".join(flit(v) for v in radii_tab) + "};")
    L.append(f"    __constant__ float kRs2[{max(len(rs2_tab), 1)}] = {{" +
             (", ".join(flit(v) for v in rs2_tab) or "0.0f") + "};")
    L.append("    struct Tab")
    L.append("    {")
    L.append("        static __device__ __forceinline__ float radius(int i) { return kRadii[i]; }")
    L.append("    };")
    L.append("")

    # ---- fkcc -------------------------------------------------------------------------------------------
    L.append("    // Robot::fkcc<rake> (reference robots/%s.hh, `fkcc`): true = rake in collision." % n)
    L.append("    template <int G>")
    L.append("    // `skip` (rake-uniform): this rake's answer is not needed; it only keeps the lanes converged.")
    L.append("    __device__ __forceinline__ bool")
    L.append("    fkcc(const vmv::EnvView &E, const float (&q)[kDim], vmv::lds_ptr slab, const bool skip)")
    L.append("    {")
    L.append("        bool bad = skip;")
    L.append("        // per-wave scratch words live right behind the sphere slab")
    L.append("        const vmv::lds_ptr scratch = slab - __lane_id() + kSlabSpheres * 3 * vmv::kWave;")
    em = Emitter(m)
    for ln in links:
        g = env_by_link[ln]
        group_spheres = [g["bound"]] + g["fine"]
        em.lines.append(f"        // ---- {ln}: {len(g['fine'])} spheres")
        em.need(group_spheres)
        for si, s in enumerate(group_spheres):
            for k in range(3):
                em.lines.append(f"        slab[{(3 * si + k)} * vmv::kWave] = {em.coord(s, k)};")
        em.lines.append(f"        bad |= vmv::env_group<G, Tab>(E, slab, scratch, {len(g['fine'])}, {radii_off[ln]}, !bad);")
        for sg in self_by_b.get(ln, []):
            gi = m["self_groups"].index(sg)
            a_sph = sorted({p[0] for p in sg["pairs"]})
            ba = sg["bound_a"]
            rs = f32(f32(radii[ba]) + f32(radii[sg["bound_b"]]))
            rs2b = float(f32(rs * rs))
            em.need(a_sph + [ba])  # already emitted (A precedes B), kept for safety
            em.lines.append(f"        {{  // {sg['a']} vs. {ln}")
            em.lines.append(
                f"            const bool gate = vmv::group_any<G>(vmv::neg(vmv::sql2_3({em.coord(ba, 0)}, {em.coord(ba, 1)}, "
                f"{em.coord(ba, 2)}, slab[0], slab[vmv::kWave], slab[2 * vmv::kWave]) - {flit(rs2b)}));")
            em.lines.append("            if (vmv::wave_any(gate && !bad))")
            em.lines.append("            {")
            em.lines.append("                bool h = false;")
            em.lines.append(f"                for (int s = 1; s <= {len(g['fine'])}; ++s)")
            em.lines.append("                {")
            em.lines.append("                    vmv::lds_cptr p = slab + 3 * s * vmv::kWave;")
            em.lines.append("                    const float bx = p[0], by = p[vmv::kWave], bz = p[2 * vmv::kWave];")
            em.lines.append(f"                    const float *rs2 = kRs2 + {rs2_off[gi]} + (s - 1) * {len(a_sph)};")
            for ai, s in enumerate(a_sph):
                em.lines.append(
                    f"                    h |= vmv::neg(vmv::sql2_3({em.coord(s, 0)}, {em.coord(s, 1)}, {em.coord(s, 2)}, "
                    f"bx, by, bz) - rs2[{ai}]);")
            em.lines.append("                }")
            em.lines.append("                bad |= (gate && vmv::group_any<G>(h));")
            em.lines.append("            }")
            em.lines.append("        }")
    L += em.lines
    L.append("        return bad;")
    L.append("    }")
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
    L.append("    template <int G>")
    L.append("    static __device__ __forceinline__ bool")
    L.append("    fkcc(const vmv::EnvView &E, const float (&q)[kDim], vmv::lds_ptr slab, const bool skip)")
    L.append("    {")
    L.append(f"        return {n}::fkcc<G>(E, q, slab, skip);")
    L.append("    }")
    L.append("    static __device__ __forceinline__ void sphere_fk(const float (&q)[kDim], float4 *out)")
    L.append("    {")
    L.append(f"        {n}::sphere_fk(q, out);")
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
    host.append("static const vmv_robot_info kRobots[] = {")
    for m in models:
        lo = ", ".join(flit(v) for v in m["lower"] + [0.0] * (16 - m["dimension"]))
        sp = ", ".join(flit(v) for v in m["span"] + [0.0] * (16 - m["dimension"]))
        ds = ", ".join(flit(v) for v in m["descale"] + [0.0] * (16 - m["dimension"]))
        jn = ", ".join('"%s"' % j for j in m["joint_names"])
        host.append(f'    {{"{m["name"]}", {m["dimension"]}, {m["n_spheres"]}, {m["resolution"]}, '
                    f'{flit(m["min_radius"])}, {flit(m["max_radius"])}, {{{lo}}}, {{{sp}}}, {{{ds}}}, '
                    f'"{m["end_effector"]}", {{{jn}}}}},')
    host.append("};")
    host.append(f"static const int kNumRobots = {len(models)};")
    with open(os.path.join(d, "robots_host.inc"), "w") as f:
        f.write("\n".join(host) + "\n")
    print("wrote vamp_mvt_amd/csrc/gen/{<robot>_dev.inc, tu_<robot>.hip, robots_host.inc}")


if __name__ == "__main__":
    import gen_code
    main([gen_code.load(n) for n in gen_code.ROBOTS])
