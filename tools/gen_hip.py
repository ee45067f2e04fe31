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

    def coord(self, s, k):
        kind, v = self.m["outputs"][s][k]
        return f"{self.prefix}{v}" if kind == "op" else flit(v)


def f32(x):
    return np.float32(x)


SELF_DEAL_MAX_A = 40  # A-side spheres kept in registers for the re-dealt self-collision form
CHUNK = 8  # fine spheres staged in the LDS slab at a time (slab = 1 bounding + CHUNK fine spheres per wave)


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
    self_links = {g["a"] for g in m["self_groups"]} | {g["b"] for g in m["self_groups"]}

    # constant table: radii per link group, [bounding, fine...]
    radii_tab, radii_off = [], {}
    for ln in links:
        g = env_by_link[ln]
        radii_off[ln] = len(radii_tab)
        radii_tab += [radii[g["bound"]]] + [radii[s] for s in g["fine"]]
    max_group = max(len(g["fine"]) for g in m["env_groups"])
    slab_spheres = 1 + min(CHUNK, max_group)

    L.append(f"namespace {n}")
    L.append("{")
    L.append(f"    constexpr int kDim = {dim};")
    L.append(f"    constexpr int kNSpheres = {m['n_spheres']};")
    L.append(f"    constexpr int kResolution = {m['resolution']};")
    L.append(f"    constexpr int kSlabSpheres = {slab_spheres};  // bounding sphere + one chunk of fine spheres")
    L.append(f"    __constant__ float kRadii[{len(radii_tab)}] = {{" + ", ".join(flit(v) for v in radii_tab) + "};")
    L.append("    struct Tab")
    L.append("    {")
    L.append("        static __device__ __forceinline__ float radius(int i) { return kRadii[i]; }")
    L.append("    };")
    L.append("")

    # ---- environment half of fkcc ------------------------------------------------------------------------
    L.append("    // Environment half of Robot::fkcc<rake> (reference robots/%s.hh `fkcc`, \"environment vs. robot" % n)
    L.append("    // collisions\"): true = some link group of this rake reports a collision.")
    L.append("    // `skip` (rake-uniform): this rake's answer is not needed; it only keeps the lanes converged.")
    L.append("    template <int G>")
    L.append("    __device__ __forceinline__ bool")
    L.append("    fkcc_env(const vmv::EnvView &E, const float (&q)[kDim], vmv::lds_ptr slab, const bool skip)")
    L.append("    {")
    L.append("        bool bad = skip;")
    L.append("        // per-wave scratch words live right behind the sphere slab")
    L.append("        const vmv::lds_ptr scratch = slab - __lane_id() + kSlabSpheres * 3 * vmv::kWave;")
    em = Emitter(m)
    for ln in links:
        g = env_by_link[ln]
        fine = g["fine"]
        chunks = [fine[i:i + CHUNK] for i in range(0, len(fine), CHUNK)]
        em.lines.append(f"        // ---- {ln}: {len(fine)} spheres")
        em.need([g["bound"]] + fine)

        def stage(slot, s, indent):
            for k in range(3):
                em.lines.append(f"{indent}slab[{3 * slot + k} * vmv::kWave] = {em.coord(s, k)};")

        stage(0, g["bound"], "        ")
        for si, s in enumerate(chunks[0]):
            stage(1 + si, s, "        ")
        em.lines.append("        {")
        em.lines.append(f"            const bool gate = vmv::env_gate<G, Tab>(E, slab, scratch, {radii_off[ln]}, !bad);")
        em.lines.append("            if (vmv::wave_any(gate))")
        em.lines.append("            {")
        done = 0
        for ci, ch in enumerate(chunks):
            if ci > 0:
                for si, s in enumerate(ch):
                    stage(1 + si, s, "                ")
            em.lines.append(f"                vmv::env_fine<G, Tab>(E, slab, scratch, {len(ch)}, {radii_off[ln] + 1 + done});")
            done += len(ch)
        em.lines.append("                bad |= gate && vmv::group_any<G>(vmv::env_flag(scratch));")
        em.lines.append("            }")
        em.lines.append("        }")
    L += em.lines
    L.append("        return bad;")
    L.append("    }")
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
    L.append("    // Self-collision half of Robot::fkcc<rake> (\"robot self-collisions\").")
    L.append("    // Groups (A, B) run when B is the current link.  Gates (bounding pair) are per lane; the fine pairs of the")
    L.append("    // few rakes whose gate fired are re-dealt over the 64 lanes as (passing lane, B sphere) items: B's")
    L.append("    // sphere comes from the LDS slab column of the lane the item belongs to, A's spheres from that lane's")
    L.append("    // registers through ds_bpermute (__shfl); hits return through LDS flags.")
    L.append(f"    // {len(batches)} pass(es) over the chain; each keeps one batch of A links in registers.")
    L.append("    template <int G>")
    L.append("    __device__ __forceinline__ bool fkcc_self(const float (&q)[kDim], vmv::lds_ptr slab, const bool skip)")
    L.append("    {")
    L.append("        bool bad = skip;")
    L.append("        const unsigned lane = __lane_id();")
    L.append("        const vmv::lds_cptr wave_slab = vmv::uniform((vmv::lds_cptr) (slab - lane));")
    L.append("        vmv::lds_u32 *const list = (vmv::lds_u32 *) (slab - lane + kSlabSpheres * 3 * vmv::kWave);")
    L.append("        vmv::lds_u32 *const flags = list + vmv::kWave;")
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
        I = "            "
        for ln in links:
            groups = [sg for sg in self_by_b.get(ln, []) if sg["a"] in batch_set]
            if not groups:
                continue
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
                em.lines.append(
                    f"{I}const bool {gn} = vmv::group_any<G>(vmv::neg(vmv::sql2_3({em.coord(ba, 0)}, {em.coord(ba, 1)}, "
                    f"{em.coord(ba, 2)}, {em.coord(bb, 0)}, {em.coord(bb, 1)}, {em.coord(bb, 2)}) - {flit(float(f32(rs * rs)))}))"
                    f" && !bad;  // {sg['a']} vs. {ln}")
            em.lines.append(f"{I}if (vmv::wave_any(" + " || ".join(gate_names) + "))")
            em.lines.append(f"{I}{{")
            em.lines.append(f"{I}    flags[lane] = 0u;")
            done = 0
            for ci, ch in enumerate(chunks):
                for si, s in enumerate(ch):
                    for k in range(3):
                        em.lines.append(f"{I}    slab[{3 * si + k} * vmv::kWave] = {em.coord(s, k)};")
                em.lines.append(f"{I}    vmv::wave_lds_sync();")
                for gi, sg in enumerate(groups):
                    a_sph = sorted({p[0] for p in sg["pairs"]})
                    b_sph = sorted({p[1] for p in sg["pairs"]})
                    assert b_sph == fine and sg["pairs"] == [[s, t] for s in a_sph for t in b_sph]
                    J = I + "    "
                    em.lines.append(f"{J}if (vmv::wave_any({gate_names[gi]}))  // {sg['a']} vs. {ln}, chunk {ci}")
                    em.lines.append(f"{J}{{")
                    em.lines.append(f"{J}    const int k = vmv::deal_list(list, {gate_names[gi]});")
                    em.lines.append(f"{J}    const int items = k * {len(ch)};")
                    em.lines.append(f"{J}    const float inv_k = 1.0f / (float) k;")
                    em.lines.append(f"{J}    for (int base = 0; base < items; base += vmv::kWave)")
                    em.lines.append(f"{J}    {{")
                    em.lines.append(f"{J}        const int i = base + (int) lane;")
                    em.lines.append(f"{J}        const bool act = i < items;")
                    em.lines.append(f"{J}        const int t = act ? (int) (((float) i + 0.5f) * inv_k) : 0;")
                    em.lines.append(f"{J}        const int j = act ? (i - t * k) : 0;")
                    em.lines.append(f"{J}        const unsigned src = list[j];")
                    em.lines.append(f"{J}        const vmv::lds_cptr p = wave_slab + 3 * t * vmv::kWave + src;")
                    em.lines.append(f"{J}        const float bx = p[0], by = p[vmv::kWave], bz = p[2 * vmv::kWave];")
                    em.lines.append(f"{J}        const float rb = kRadii[{radii_off[ln] + 1 + done} + t];")
                    em.lines.append(f"{J}        bool h = false;")
                    for s in a_sph:
                        cs = []
                        for k in range(3):
                            kind, v = m["outputs"][s][k]
                            cs.append(f"__shfl({em.prefix}{v}, (int) src)" if kind == "op" else flit(v))
                        em.lines.append(f"{J}        {{")
                        em.lines.append(f"{J}            const float rs = {flit(radii[s])} + rb;")
                        em.lines.append(f"{J}            h |= vmv::neg(vmv::sql2_3({cs[0]}, {cs[1]}, {cs[2]}, bx, by, bz) - rs * rs);")
                        em.lines.append(f"{J}        }}")
                    em.lines.append(f"{J}        if (h && act) flags[src] = 1u;")
                    em.lines.append(f"{J}    }}")
                    em.lines.append(f"{J}    vmv::wave_lds_sync();")
                    em.lines.append(f"{J}}}")
                done += len(ch)
            em.lines.append(f"{I}    bad |= vmv::group_any<G>(flags[lane] != 0u);")
            em.lines.append(f"{I}}}")
        L += em.lines
        L.append("        }")
    L.append("        return bad;")
    L.append("    }")
    L.append(f"    constexpr int kSelfPasses = {len(batches)};")
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
    L.append("    fkcc_env(const vmv::EnvView &E, const float (&q)[kDim], vmv::lds_ptr slab, const bool skip)")
    L.append("    {")
    L.append(f"        return {n}::fkcc_env<G>(E, q, slab, skip);")
    L.append("    }")
    L.append("    template <int G>")
    L.append("    static __device__ __forceinline__ bool")
    L.append("    fkcc_self(const float (&q)[kDim], vmv::lds_ptr slab, const bool skip)")
    L.append("    {")
    L.append(f"        return {n}::fkcc_self<G>(q, slab, skip);")
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
        max_bound = max(m["radii"][m["n_spheres"]:])
        host.append(f'    {{"{m["name"]}", {m["dimension"]}, {m["n_spheres"]}, {m["resolution"]}, '
                    f'{flit(m["min_radius"])}, {flit(m["max_radius"])}, {flit(max_bound)}, {{{lo}}}, {{{sp}}}, {{{ds}}}, '
                    f'"{m["end_effector"]}", {{{jn}}}}},')
    host.append("};")
    host.append(f"static const int kNumRobots = {len(models)};")
    with open(os.path.join(d, "robots_host.inc"), "w") as f:
        f.write("\n".join(host) + "\n")
    print("wrote vamp_mvt_amd/csrc/gen/{<robot>_dev.inc, tu_<robot>.hip, robots_host.inc}")


if __name__ == "__main__":
    import gen_code
    main([gen_code.load(n) for n in gen_code.ROBOTS])
