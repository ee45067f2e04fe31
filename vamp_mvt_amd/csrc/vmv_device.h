// vmv_device.h — gfx950 device primitives for the motion-validation hot path.
//
// One wavefront lane = one rake lane (one configuration).  All arithmetic is
// IEEE fp32 with one rounding per written operation: this translation unit is
// compiled with -ffp-contract=off and without fast-math, so that the per-lane
// results are bit-identical to the reference's AVX2 lanes (SURVEY.md §2 table of
// SIMD primitives).  Reference citations are file:line under
// /root/reference/src/impl/vamp/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vmv
{
    constexpr int kWave = 64;  // gfx950 wavefront
    // Row stride (words) of the per-wave sphere slab: row r of lane l sits at r * kRow + l.  The odd stride puts the
    // rows of one lane on different LDS banks (ds_read_b32 banks = word mod 32), so re-dealt items that read
    // different spheres of the same configuration do not collide (a stride of 64 made that an N-way conflict).
    constexpr int kRow = kWave + 1;

    // explicit LDS address space: pointers that cross a (non-inlined) call keep ds_read/ds_write addressing
    using lds_float = __attribute__((address_space(3))) float;
    using lds_ptr = lds_float *;
    using lds_cptr = const lds_float *;
    typedef float v4f __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ v4f lds_load4(lds_cptr p)
    {
        return *(const __attribute__((address_space(3))) v4f *) p;  // 16-byte aligned record -> ds_read_b128
    }

    // ------------------------------------------------------------------------
    // L0: value-type semantics (vector/avx.hh, vector/interface.hh), per lane
    // ------------------------------------------------------------------------
    __device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
    __device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
    // `not v.test_zero()` for one lane: the sign bit (interface.hh:257-277, avx.hh:385-389)
    __device__ __forceinline__ bool neg(float f) { return (f2u(f) >> 31) != 0u; }
    // _mm256_max_ps / _mm256_min_ps operand order and NaN rule (avx.hh:429-439)
    __device__ __forceinline__ float x86_max(float a, float b) { return a > b ? a : b; }
    __device__ __forceinline__ float x86_min(float a, float b) { return a < b ? a : b; }
    __device__ __forceinline__ float vclamp(float v, float lo, float hi) { return x86_min(x86_max(v, lo), hi); }
    __device__ __forceinline__ float vabs(float v) { return u2f(f2u(v) & 0x7fffffffu); }

    // vector/avx.hh:455-548: cephes sine.  cvtps_epi32 is round-to-nearest-even = v_rndne + v_cvt.
    __device__ __forceinline__ float vsin(float x)
    {
        uint32_t sign_bit = f2u(x) & 0x80000000u;
        x = vabs(x);
        float y = x * 1.27323954473516f;
        int32_t j = (y < 2147483648.0f) ? __float2int_rn(y) : (int32_t) 0x80000000;
        j = (int32_t) (((uint32_t) j + 1u) & ~1u);
        y = (float) j;
        const uint32_t swap_sign = ((uint32_t) j & 4u) << 29;
        const bool poly_sin = (((uint32_t) j & 2u) == 0u);
        sign_bit ^= swap_sign;
        const float xmm1 = y * -0.78515625f;
        const float xmm2 = y * -2.4187564849853515625e-4f;
        const float xmm3 = y * -3.77489497744594108e-8f;
        x = x + xmm1;
        x = x + xmm2;
        x = x + xmm3;
        const float z = x * x;
        float yc = 2.443315711809948E-005f;
        yc = yc * z;
        yc = yc + -1.388731625493765E-003f;
        yc = yc * z;
        yc = yc + 4.166664568298827E-002f;
        yc = yc * z;
        yc = yc * z;
        const float tmp = z * 0.5f;
        yc = yc - tmp;
        yc = yc + 1.0f;
        float y2 = -1.9515295891E-4f;
        y2 = y2 * z;
        y2 = y2 + 8.3321608736E-3f;
        y2 = y2 * z;
        y2 = y2 + -1.6666654611E-1f;
        y2 = y2 * z;
        y2 = y2 * x;
        y2 = y2 + x;
        // and/andnot select followed by an add of +0.0f (turns a selected -0.0f into +0.0f)
        const float sel = (poly_sin ? y2 : yc) + 0.0f;
        return u2f(f2u(sel) ^ sign_bit);
    }

    // vector/interface.hh:447-458
    __device__ __forceinline__ float vcos(float x)
    {
        const float PI = 3.14159265359f;
        const float v_sq = x + (float) (PI / 2.);
        const float sub = (v_sq >= PI) ? (float) (2 * PI) : 0.0f;
        return vsin(v_sq - sub);
    }

    // ------------------------------------------------------------------------
    // wave / rake-group helpers.  G = 1: every lane is its own rake (one configuration replicated over
    // the reference's 8 lanes).  G = 8: lanes 8k..8k+7 form one reference rake (planning/validate.hh).
    // ------------------------------------------------------------------------
    __device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

    template <int G>
    __device__ __forceinline__ bool group_any(bool p)
    {
        if constexpr (G == 1)
        {
            return p;
        }
        else
        {
            const uint64_t b = __ballot(p);
            const unsigned lane = __lane_id();
            return ((b >> (lane & ~7u)) & 0xffull) != 0ull;
        }
    }

    template <int G>
    __device__ __forceinline__ float group_max(float v)
    {
        if constexpr (G == 1)
        {
            return v;
        }
        else
        {
            // butterfly over the 8 lanes of the rake (xor 1, 2, 4)
            v = fmaxf(v, __shfl_xor(v, 1));
            v = fmaxf(v, __shfl_xor(v, 2));
            v = fmaxf(v, __shfl_xor(v, 4));
            return v;
        }
    }

    // ------------------------------------------------------------------------
    // L1a: environment as the kernels see it.  Primitive records live in LDS (staged once per workgroup);
    // CAPT arrays stay in HBM/L2 except the split planes, which are staged in LDS when they fit.
    // Record layouts (floats), every list sorted by min_distance (collision/environment.hh:46-72):
    //   sphere     [8]  x y z r | min_d 0 0 0
    //   capsule    [12] x1 y1 z1 xv | yv zv r rdv | min_d 0 0 0
    //   z_capsule  [8]  x1 y1 z1 zv | r rdv min_d 0
    //   cuboid     [16] x y z a1x | a1y a1z a2x a2y | a2z a3x a3y a3z | r1 r2 r3 min_d
    //   z_cuboid   [12] x y z a1x | a1y a2x a2y r1 | r2 r3 min_d 0
    // ------------------------------------------------------------------------
    constexpr int kSphereRec = 8, kCapsuleRec = 12, kZCapsuleRec = 8, kCuboidRec = 16, kZCuboidRec = 12;

    struct CaptDev
    {
        // what the query walks (vmv_capt_build.h, CaptQueryDev): the planes in 3-level blocks, one 128-byte record per
        // leaf, and each leaf's points sorted by their distance to the leaf's cell.  Kept together and first: a query
        // fetches these 22 dwords with a few wide scalar loads.
        float aabb_top[6];
        float r_point;
        uint32_t nlog2;
        const float *q_planes;
        const uint32_t *q_leaves;
        const float *q_x, *q_y, *q_z;
        uint32_t n_tests;
        float cut_t0, cut_inv_step;
        uint32_t pad_;
        const float *q_dist;   // distance grid (nullptr: none): lower bound of the distance to the nearest cloud point per cell
        uint32_t dist_dims[3];
        float dist_inv_cell;
        float dist_origin[3];
        uint32_t pad2_;
        // the reference's layout (what inspection returns; `tests` also feeds the LDS copy of the top levels)
        const float *tests;          // 2^nlog2 - 1
        const uint32_t *aff_starts;  // 2^nlog2 + 1
        const float *aabbs;          // 2^nlog2 * 6
        const float *aff_x, *aff_y, *aff_z;  // n_aff * 8
    };

    constexpr int kMaxCapt = 4;
    constexpr int kCaptCutBuckets = 32;       // radius buckets of a leaf record
    constexpr float kCaptCutMargin = 1e-4f;   // metres
    constexpr int kCaptLeafWords = 32;        // [0..5] box, [6] first vector, [7] vectors, [8..23] 32 x uint16 counts
    constexpr int kCaptPlaneLevels = 3;       // tree levels per block of CaptDev::q_planes
    constexpr int kCaptPlaneBlock = 8;        // floats per block: the 7 planes of a 3-level subtree in local heap order

    // The blocked copy of the split planes: the tree's levels are cut into groups of 3, bottom-aligned (the first group
    // holds nlog2 mod 3 levels when that is not 0), a block is the subtree under one node of the group's first level —
    // [root | lo child, hi child | their four children | pad] — and the blocks of a group are stored left to right, so
    // the block number inside its group is the path (one bit per level, 1 = hi side) that leads to its root.  Groups
    // follow each other from the root down: group g starts at float 8 * (number of blocks before it).
    __host__ __device__ inline uint32_t capt_group_levels(const uint32_t nlog2, const uint32_t first_level)
    {
        const uint32_t s0 = nlog2 % (uint32_t) kCaptPlaneLevels;
        return (first_level == 0u && s0 != 0u) ? s0 : (uint32_t) kCaptPlaneLevels;
    }
    // where plane `i` (heap index, level l = floor(log2(i + 1))) sits: float offset from the start of the copy
    __host__ __device__ inline uint32_t capt_plane_slot(const uint32_t nlog2, const uint32_t l, const uint32_t i)
    {
        uint32_t base = 0u, gs = 0u;  // group that holds level l
        for (;;)
        {
            const uint32_t nl = capt_group_levels(nlog2, gs);
            if (l < gs + nl) break;
            base += (uint32_t) kCaptPlaneBlock << gs;
            gs += nl;
        }
        const uint32_t lw = l - gs, t = i + 1u;
        const uint32_t local1 = (1u << lw) | (t & ((1u << lw) - 1u));
        const uint32_t block = (t >> lw) - (1u << gs);
        return base + block * (uint32_t) kCaptPlaneBlock + local1 - 1u;
    }
    // floats of the whole copy (limit = nlog2) or of its leading whole groups that fit `budget_floats`
    __host__ __device__ inline uint32_t capt_plane_floats(const uint32_t nlog2, const uint32_t budget_floats = 0xffffffffu)
    {
        uint32_t base = 0u, gs = 0u;
        while (gs < nlog2)
        {
            const uint32_t next = base + ((uint32_t) kCaptPlaneBlock << gs);
            if (next > budget_floats) break;
            base = next;
            gs += capt_group_levels(nlog2, gs);
        }
        return base;
    }

    // Multi-level Voxel Table (collision/mvt.hh) as the query reads it: one dense grid of voxel indices (the
    // reference's three pointer levels collapsed; see vmv_mvt_build.h), voxel boxes, compact SoA points.
    struct MvtDev
    {
        const uint32_t *cells;       // grid_width^3 -> voxel index or 0xffffffff
        const float *vox_bbox;       // [n_vox][6]
        const uint32_t *vox_offset;  // [n_vox + 1]
        const float *px, *py, *pz;
        float gmin[3], gmax[3], ws_min[3];
        float inv_scale, r_point;
        uint32_t grid_width;
    };
    constexpr int kMaxMvt = 4;

    struct HeightFieldDev  // collision/shapes.hh:250-312
    {
        const float *data;  // row-major [yd][xd]
        float x, y, z, xs, ys, zs;
        float xd, yd, xd2, yd2;  // image size and half size, as the floats the reference converts them to
        uint32_t last;           // xd * yd - 1
    };
    constexpr int kMaxHeightFields = 4;

    struct GridDev
    {
        const uint32_t *cells;
        uint32_t dims[3];
        float origin[3], inv_cell;
    };
    constexpr int kGridClasses = 4;

    struct EnvDev  // kernel argument (by value)
    {
        const float *prims;  // HBM image of the LDS primitive block
        uint32_t n_floats;   // size of that block
        uint32_t n_sphere, n_capsule, n_zcapsule, n_cuboid, n_zcuboid;
        uint32_t off_sphere, off_capsule, off_zcapsule, off_cuboid, off_zcuboid;  // float offsets in the block
        uint32_t off_md_sphere, off_md_capsule, off_md_zcapsule, off_md_cuboid, off_md_zcuboid;  // min_distance arrays
        // candidate words (32 primitives each) per list: first word index, and whether all lists fit kMaskWords
        uint32_t wbase_sphere, wbase_capsule, wbase_zcapsule, wbase_cuboid, wbase_zcuboid, masked_fine;
        // shared candidate words: a list of at most 32 primitives may start at bit `wshift` of a word whose lower bits
        // belong to the lists before it (five short lists then take one or two words instead of five; a list longer
        // than one word always starts at bit 0 of a word of its own).  All 0 unless the word-aligned layout does not
        // fit kMaskWords; always 0 in environments the three-list variant serves (vmv_api.hip: finalize)
        uint32_t wshift_sphere, wshift_capsule, wshift_zcapsule, wshift_cuboid, wshift_zcuboid;
        // broad-phase grids of the gate pass (vmv_grid_build.h; robot specific): the robot's links are split into
        // kGridClasses classes by bounding radius (tools/gen_hip.py) and each class has a grid built for its largest
        // radius - a finger (r = 0.024 m) walks far fewer candidates than in a grid sized for the 0.18 m upper arm.
        // Candidate words per cell; grid[0].cells == nullptr -> counted loops over the whole lists.
        GridDev grid[kGridClasses];
        uint32_t grid_words;
        // "some static link of the robot (its spheres do not depend on the configuration) collides with this
        // environment": evaluated once per (environment, robot) by static_links_kernel with the same device functions
        uint32_t static_hit;
        // an ill-formed cuboid / capsule (axes not orthonormal, rdv != 1 / |v|^2): its "distance" is not 1-Lipschitz, so a
        // bounding sphere can test clear while a sphere inside it tests colliding — the reference's answer then depends on
        // which spheres ITS gates guard.  Such environments run the variant that keeps the reference's groups and the
        // full sorted loops (the launchers pick it), never the merged gates of the primitive-only variants.
        uint32_t ill_formed;
        unsigned long long link_skip;  // bit g: environment group g can never touch this environment (reach certificates)
        uint32_t n_capt;
        uint32_t capt0_n_tests;  // size of capt[0].tests (candidate for LDS staging behind the primitive block)
        uint32_t n_mvt;
        CaptDev capt[kMaxCapt];
        MvtDev mvt[kMaxMvt];
        uint32_t n_heightfield;
        HeightFieldDev heightfield[kMaxHeightFields];
        // collision/attachments.hh: spheres rigidly attached to the end-effector frame.  attach_tf = the attachment's
        // frame relative to the end effector (row-major 3 x 4: R | t), attach_spheres = [n_attach][4] x y z r in it.
        uint32_t n_attach;
        float attach_tf[12];
        const float *attach_spheres;
    };
    constexpr int kMaxAttachSpheres = 256;

    // explicit address spaces for everything that crosses a non-inlined call: the constant space makes the
    // environment header scalar loads (s_load), the global space makes per-lane CAPT reads global_load (not flat)
    using env_cptr = const __attribute__((address_space(4))) EnvDev *;
    using gf_cptr = const __attribute__((address_space(1))) float *;
    using cf_cptr = const __attribute__((address_space(4))) float *;
    using gu_cptr = const __attribute__((address_space(1))) uint32_t *;
    typedef __attribute__((address_space(1))) v4f g_v4f;
    typedef __attribute__((address_space(3))) v4f lds_v4f;

    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    // per-workgroup LDS words in front of the radius table: one 64-word hit-flag row per wave for the CAPT walk
    constexpr int kCaptFlagWords = 4 * kWave;

    struct EnvView
    {
        env_cptr dev;              // device memory, wave-uniform (scalar loads)
        lds_cptr lds;              // primitive block in LDS
        uint32_t capt0_planes_in_lds;  // how many leading split planes of point cloud 0 sit in LDS behind the block
        lds_cptr radii;            // the robot's radius table (gen: kRadii) copied to LDS once per workgroup, for
                                   // re-dealt items whose sphere index differs per lane (a __constant__ table read
                                   // with a per-lane index is a global load on the critical path of every round)
    };

    // this wave's 64-word hit-flag row of the CAPT walk: kCaptFlagWords words sit right in front of the radius table
    __device__ __forceinline__ lds_u32 *capt_flag_row(const EnvView &E)
    {
        return (lds_u32 *) E.radii - kCaptFlagWords + (threadIdx.x / kWave) * kWave;
    }

    // collision/math.hh:10-42
    __device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
    {
        return (ax * bx) + (ay * by) + (az * bz);
    }
    __device__ __forceinline__ float sql2_3(float ax, float ay, float az, float bx, float by, float bz)
    {
        const float xs = ax - bx, ys = ay - by, zs = az - bz;
        return dot3(xs, ys, zs, xs, ys, zs);
    }
    // collision/sphere_sphere.hh:9-23
    __device__ __forceinline__ float
    sphere_sphere_sql2(float ax, float ay, float az, float ar, float bx, float by, float bz, float br)
    {
        const float sum = sql2_3(ax, ay, az, bx, by, bz);
        const float rs = ar + br;
        return sum - rs * rs;
    }

    __device__ __forceinline__ void wave_lds_sync_()
    {
        // same-wave LDS hand-off (DS ops of one wave retire in order; this pins the compiler's order too)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // ------------------------------------------------------------------------
    // wave helpers for re-dealing variable-length per-lane work over the 64 lanes.  Every __shfl below runs with all
    // lanes enabled (ds_bpermute returns 0 from a disabled source lane).
    // ------------------------------------------------------------------------
    __device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
    {
        const uint32_t lane = __lane_id();
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1)
        {
            const uint32_t up = (uint32_t) __shfl_up((int) v, d);
            v += (lane >= (uint32_t) d) ? up : 0u;
        }
        return v;
    }
    // first lane o with ends[o] > t, for a non-decreasing `ends` (one value per lane) and t < ends[63]
    __device__ __forceinline__ uint32_t wave_upper_bound(const uint32_t ends, const uint32_t t)
    {
        uint32_t lo = 0u, hi = (uint32_t) kWave - 1u;
#pragma unroll
        for (int s = 0; s < 6; ++s)
        {
            const uint32_t mid = (lo + hi) >> 1;
            const bool right = (uint32_t) __shfl((int) ends, (int) mid) <= t;
            lo = right ? mid + 1u : lo;
            hi = right ? hi : mid;
        }
        return lo < (uint32_t) kWave ? lo : (uint32_t) kWave - 1u;
    }

#ifndef VMV_ABLATE_ENV
#define VMV_ABLATE_ENV 0  // measurement aid (tools only): 1 = environment kernel without fine phase, 2 = FK only,
                          // 6 / 7 / 8 / 9 = CAPT query stops after the top box / the descent / the leaf test / the first vector
#endif

    // CAPT::collides_simd (collision/capt.hh:428-512) for the wave's 64 sphere queries (one per lane; the reference's
    // `inbounds.none()` early returns do not change any lane's answer, so lanes are independent queries).
    //   1. per lane: top-AABB test (without r_point, as the reference), nlog2 plane descents (top levels in LDS), leaf
    //      AABB test with r + r_point -> this lane's affordance vectors [start, start + count).
    //   2. the (query, affordance vector) pairs of ALL lanes are re-dealt over the 64 lanes: 64 vectors of 8 points per
    //      round, whatever their owner.  A leaf of a 10,000-point cloud holds 18 (Fetch radii) to 87 (Baxter) vectors
    //      and only the lanes that pass the leaf test have any, so a per-lane walk ran at ~40 % lane utilisation for
    //      max-over-lanes iterations; re-dealt it is sum / 64 rounds.  Every point is tested with the reference's
    //      expression (sql2_3 <= (r + r_point)^2); hits are OR-ed per owner through `flags` (this wave's LDS row).
    // Q queries per lane (Q = 1: a sphere; Q = 2: the bounding spheres of two consecutive links, vmv::capt_gate_pair):
    // the stages below run for all of a lane's queries together, so the dependent fetches of one query (distance-grid
    // cell, plane blocks, leaf record, first vector) are in flight next to the other's — the walk is a chain of memory
    // latencies, not arithmetic.  Queries are independent: out[q] is exactly what a Q = 1 call on query q returns.
    template <int Q>
    __device__ __forceinline__ void
    capt_collides_q(env_cptr D, const uint32_t ci, lds_cptr planes_lds, const uint32_t n_lds, lds_u32 *flags, const float (&x)[Q],
                    const float (&y)[Q], const float (&z)[Q], const float (&r)[Q], const bool (&active)[Q], bool (&out)[Q])
    {
        // the cloud's header in one go (a few wide scalar loads, pinned here: left to itself the compiler sinks every
        // field behind the branch that first needs it, one scalar-cache round trip each)
        const float t0 = D->capt[ci].aabb_top[0], t1 = D->capt[ci].aabb_top[1], t2 = D->capt[ci].aabb_top[2];
        const float t3 = D->capt[ci].aabb_top[3], t4 = D->capt[ci].aabb_top[4], t5 = D->capt[ci].aabb_top[5];
        const float r_point = D->capt[ci].r_point, cut_t0 = D->capt[ci].cut_t0, cut_inv_step = D->capt[ci].cut_inv_step;
        const uint32_t nlog2 = D->capt[ci].nlog2;
        const gf_cptr bplanes = (gf_cptr) D->capt[ci].q_planes;
        typedef const uint32_t __attribute__((address_space(1))) *gw_cptr;
        typedef const uint16_t __attribute__((address_space(1))) *gh_cptr;
        const gw_cptr leaves = (gw_cptr) D->capt[ci].q_leaves;
        const gf_cptr ax = (gf_cptr) D->capt[ci].q_x, ay = (gf_cptr) D->capt[ci].q_y, az = (gf_cptr) D->capt[ci].q_z;
        asm volatile("" ::"s"(t0), "s"(t1), "s"(t2), "s"(t3), "s"(t4), "s"(t5), "s"(r_point), "s"(cut_t0), "s"(cut_inv_step),
                     "s"(nlog2), "s"(bplanes), "s"(leaves), "s"(ax), "s"(ay), "s"(az));
        bool inb[Q];
        bool any = false;
#pragma unroll
        for (int q = 0; q < Q; ++q)
        {
            out[q] = false;
            inb[q] = active[q];
            inb[q] = inb[q] && (x[q] + r[q] >= t0) && (x[q] - r[q] <= t3);
            inb[q] = inb[q] && (y[q] + r[q] >= t1) && (y[q] - r[q] <= t4);
            inb[q] = inb[q] && (z[q] + r[q] >= t2) && (z[q] - r[q] <= t5);
            any = any || inb[q];
        }
        if (!wave_any(any) || VMV_ABLATE_ENV == 6) return;
        // distance grid: a centre whose cell is farther from every cloud point than r + r_point (+ 1e-4 m) cannot hit
        // whatever leaf it descends to (the leaf lists are subsets of the cloud).  The load is issued here and used after
        // the LDS part of the descent; if no lane is left, the wave skips the rest of the descent, the leaf record and the walk.
        const gf_cptr dgrid = (gf_cptr) D->capt[ci].q_dist;
        float dist_lb[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) dist_lb[q] = 0.0f;
        if (dgrid != nullptr)
        {
            const float inv = D->capt[ci].dist_inv_cell;
            const uint32_t g0 = D->capt[ci].dist_dims[0], g1 = D->capt[ci].dist_dims[1], g2 = D->capt[ci].dist_dims[2];
#pragma unroll
            for (int q = 0; q < Q; ++q)
            {
                const float fx = (x[q] - D->capt[ci].dist_origin[0]) * inv, fy = (y[q] - D->capt[ci].dist_origin[1]) * inv,
                            fz = (z[q] - D->capt[ci].dist_origin[2]) * inv;
                const bool in_grid = inb[q] && fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx < (float) g0 && fy < (float) g1 &&
                                     fz < (float) g2;
                const size_t cell = in_grid ? ((size_t) (uint32_t) fx * g1 + (uint32_t) fy) * g2 + (uint32_t) fz : 0;
                dist_lb[q] = in_grid ? dgrid[cell] : 0.0f;
            }
        }
        auto cut_by_distance = [&]() -> bool  // false: no lane of the wave has a query left
        {
            bool left = false;
#pragma unroll
            for (int q = 0; q < Q; ++q)
            {
                inb[q] = inb[q] && !(dist_lb[q] > (r[q] + r_point) + kCaptCutMargin);
                left = left || inb[q];
            }
            return wave_any(left);
        };

        // descent through the blocked copy of the planes (capt_plane_slot): three levels per step — one block of 7
        // planes fetched at once (LDS for the leading groups staged there, n_lds floats; one 32-byte read through
        // L1 / L2 below), then three compares on values already in registers.  `path` (one bit per level) is the block
        // number inside the next group and, after the last level, the leaf.
        uint32_t path[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) path[q] = 0u;
        uint32_t k = 0u, base = 0u;
        bool checked = false;
        for (uint32_t gs = 0u; gs < nlog2;)
        {
            const uint32_t nl = capt_group_levels(nlog2, gs);
            const uint32_t next = base + ((uint32_t) kCaptPlaneBlock << gs);
            v4f pa[Q], pb[Q];
            if (next <= n_lds)
            {
#pragma unroll
                for (int q = 0; q < Q; ++q)
                {
                    const lds_v4f *b = (const lds_v4f *) (planes_lds + base + path[q] * (uint32_t) kCaptPlaneBlock);
                    pa[q] = b[0], pb[q] = b[1];
                }
            }
            else
            {
                if (!checked)  // first group outside LDS: the distance bound has had the LDS groups to arrive
                {
                    checked = true;
                    if (!cut_by_distance()) return;
                }
#pragma unroll
                for (int q = 0; q < Q; ++q)
                {
                    // (a query that is out — top box, distance bound, inactive lane — reads block 0 of the group: its
                    // lanes share one cache line instead of fetching 64 different ones nobody uses)
                    const g_v4f *b = (const g_v4f *) (bplanes + base + (size_t) (inb[q] ? path[q] : 0u) * kCaptPlaneBlock);
                    pa[q] = b[0], pb[q] = b[1];
                }
            }
            const uint32_t k0 = k, k1 = (k0 == 2) ? 0 : k0 + 1, k2 = (k1 == 2) ? 0 : k1 + 1;
#pragma unroll
            for (int q = 0; q < Q; ++q)
            {
                const float c0k = (k0 == 0) ? x[q] : (k0 == 1) ? y[q] : z[q];
                const bool c0 = c0k >= pa[q].x;
                path[q] = (path[q] << 1) | (uint32_t) c0;
                if (nl >= 2u)
                {
                    const float c1k = (k1 == 0) ? x[q] : (k1 == 1) ? y[q] : z[q];
                    const bool c1 = c1k >= (c0 ? pa[q].z : pa[q].y);
                    path[q] = (path[q] << 1) | (uint32_t) c1;
                    if (nl >= 3u)
                    {
                        const float c2k = (k2 == 0) ? x[q] : (k2 == 1) ? y[q] : z[q];
                        // children of the lo child: 3, 4; of the hi child: 5, 6
                        const float lo = c1 ? pb[q].x : pa[q].w, hi = c1 ? pb[q].z : pb[q].y;
                        const bool c2 = c2k >= (c0 ? hi : lo);
                        path[q] = (path[q] << 1) | (uint32_t) c2;
                    }
                }
            }
            k = (k + nl) % 3u;
            base = next;
            gs += nl;
        }
        if (!checked)
        {
            if (!cut_by_distance()) return;
        }
        if (VMV_ABLATE_ENV == 7)
        {
#pragma unroll
            for (int q = 0; q < Q; ++q) out[q] = path[q] == 0x12345u;
            return;
        }
        // the leaf's record: box, first vector, vector count and the bucket counts in one cache line
        float rc_sq[Q];
        uint32_t start[Q], count[Q];
        any = false;
        {
            v4f r0[Q], r1[Q];
            uint32_t cut[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q)
            {
                const float rr = r[q] + r_point;
                rc_sq[q] = rr * rr;
                const gw_cptr rec = leaves + (size_t) (inb[q] ? path[q] : 0u) * kCaptLeafWords;  // (out: leaf 0, as above)
                const g_v4f *recv = (const g_v4f *) rec;
                r0[q] = recv[0], r1[q] = recv[1];
                // the leaf's points are sorted by their distance to the leaf's cell (a lower bound of their distance to
                // this centre): only the leading vectors that hold a point within r + r_point (+ 1e-4 m) can hit
                // (vmv_capt_build.h).  First bucket whose threshold t0 + b * step exceeds rr + margin; radii beyond the
                // table (and NaN: the comparison is false) take the last bucket = the whole list
                const float bf = ((rr + kCaptCutMargin) - cut_t0) * cut_inv_step;
                const int b = (bf < (float) (kCaptCutBuckets - 2)) ? (int) fmaxf(floorf(bf), -1.0f) + 1 : kCaptCutBuckets - 1;
                cut[q] = (uint32_t) ((gh_cptr) (rec + 8))[b];
            }
#pragma unroll
            for (int q = 0; q < Q; ++q)
            {
                const float d0 = x[q] - vclamp(x[q], r0[q].x, r0[q].w);
                const float d1 = y[q] - vclamp(y[q], r0[q].y, r1[q].x);
                const float d2 = z[q] - vclamp(z[q], r0[q].z, r1[q].y);
                const float distsq = d0 * d0 + d1 * d1 + d2 * d2;
                // (cut == 0: no vector can matter, same answer as walking none; in the condition so that the bucket
                // count is fetched with the record instead of one memory latency later)
                inb[q] = inb[q] && (distsq <= rc_sq[q]) && (cut[q] != 0u);
                any = any || inb[q];
                start[q] = inb[q] ? __float_as_uint(r1[q].z) : 0u;
                count[q] = inb[q] ? __float_as_uint(r1[q].w) : 0u;
                count[q] = (cut[q] == 0xffffu) ? count[q] : min(count[q], cut[q]);
            }
        }
        if (!wave_any(any)) return;
        if (VMV_ABLATE_ENV == 8)
        {
#pragma unroll
            for (int q = 0; q < Q; ++q) out[q] = count[q] == 0x1234u;
            return;
        }
        // every query tests its own FIRST vector (the leaf's representative point and the first afforded points) before
        // anything is re-dealt: no owner search, and a sphere well inside the cloud usually hits right there - the
        // reference's early exit for the common case
        bool first_hit[Q];
        {
            v4f x0[Q], x1[Q], y0[Q], y1[Q], z0[Q], z1[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q)
            {
                // (a query without vectors reads vector 0 and discards it: straight-line loads, all queries in flight)
                const size_t off = count[q] != 0u ? 8 * (size_t) start[q] : 0;
                const g_v4f *px = (const g_v4f *) (ax + off);
                const g_v4f *py = (const g_v4f *) (ay + off);
                const g_v4f *pz = (const g_v4f *) (az + off);
                x0[q] = px[0], x1[q] = px[1], y0[q] = py[0], y1[q] = py[1], z0[q] = pz[0], z1[q] = pz[1];
            }
#pragma unroll
            for (int q = 0; q < Q; ++q)
            {
                bool h = false;
                h |= sql2_3(x0[q].x, y0[q].x, z0[q].x, x[q], y[q], z[q]) <= rc_sq[q];
                h |= sql2_3(x0[q].y, y0[q].y, z0[q].y, x[q], y[q], z[q]) <= rc_sq[q];
                h |= sql2_3(x0[q].z, y0[q].z, z0[q].z, x[q], y[q], z[q]) <= rc_sq[q];
                h |= sql2_3(x0[q].w, y0[q].w, z0[q].w, x[q], y[q], z[q]) <= rc_sq[q];
                h |= sql2_3(x1[q].x, y1[q].x, z1[q].x, x[q], y[q], z[q]) <= rc_sq[q];
                h |= sql2_3(x1[q].y, y1[q].y, z1[q].y, x[q], y[q], z[q]) <= rc_sq[q];
                h |= sql2_3(x1[q].z, y1[q].z, z1[q].z, x[q], y[q], z[q]) <= rc_sq[q];
                h |= sql2_3(x1[q].w, y1[q].w, z1[q].w, x[q], y[q], z[q]) <= rc_sq[q];
                first_hit[q] = h && count[q] != 0u;
                start[q] += 1u;
                count[q] = (first_hit[q] || count[q] == 0u) ? 0u : count[q] - 1u;
                out[q] = first_hit[q];
            }
        }
        if (VMV_ABLATE_ENV == 9) return;
        const uint32_t lane = __lane_id();
        // the remaining vectors, query by query: the (lane, vector) pairs of one query index are re-dealt over the wave
#pragma unroll
        for (int q = 0; q < Q; ++q)
        {
            if (!wave_any(count[q] != 0u)) continue;
            const uint32_t cnt = count[q];
            const float qx_ = x[q], qy_ = y[q], qz_ = z[q], qr_ = rc_sq[q];
            const uint32_t ends = wave_inclusive_scan(cnt);
            const uint32_t total = (uint32_t) __builtin_amdgcn_readlane((int) ends, kWave - 1);
            const uint32_t vbase = start[q] - (ends - cnt);  // vector index of item t of this lane's query = vbase + t
            flags[lane] = 0u;
            wave_lds_sync_();
            // software-pipelined rounds: the owner search of round k + 1 (six dependent ds_bpermute) is issued while the
            // twelve 16-byte loads of round k are in flight; the walk is bound by these latencies, not by arithmetic
            struct Who
            {
                uint32_t o, i;
                float qx, qy, qz, qr;
                bool act;
            };
            auto who = [&](const uint32_t base_t) -> Who
            {
                Who w;
                const uint32_t t = base_t + lane;
                w.act = t < total;
                w.o = wave_upper_bound(ends, w.act ? t : 0u);
                w.i = (uint32_t) __shfl((int) vbase, (int) w.o) + t;
                w.qx = __shfl(qx_, (int) w.o), w.qy = __shfl(qy_, (int) w.o), w.qz = __shfl(qz_, (int) w.o);
                w.qr = __shfl(qr_, (int) w.o);
                return w;
            };
            Who cur = who(0u);
            for (uint32_t base_t = 0; base_t < total; base_t += (uint32_t) kWave)
            {
                const size_t off = cur.act ? 8 * (size_t) cur.i : 0;  // lanes past the end read vector 0 and discard it
                const g_v4f *px = (const g_v4f *) (ax + off);
                const g_v4f *py = (const g_v4f *) (ay + off);
                const g_v4f *pz = (const g_v4f *) (az + off);
                const v4f x0 = px[0], x1 = px[1], y0 = py[0], y1 = py[1], z0 = pz[0], z1 = pz[1];
                const Who nxt = who(base_t + (uint32_t) kWave);  // (past the last round: all lanes inactive, harmless)
                bool h = false;
                h |= sql2_3(x0.x, y0.x, z0.x, cur.qx, cur.qy, cur.qz) <= cur.qr;
                h |= sql2_3(x0.y, y0.y, z0.y, cur.qx, cur.qy, cur.qz) <= cur.qr;
                h |= sql2_3(x0.z, y0.z, z0.z, cur.qx, cur.qy, cur.qz) <= cur.qr;
                h |= sql2_3(x0.w, y0.w, z0.w, cur.qx, cur.qy, cur.qz) <= cur.qr;
                h |= sql2_3(x1.x, y1.x, z1.x, cur.qx, cur.qy, cur.qz) <= cur.qr;
                h |= sql2_3(x1.y, y1.y, z1.y, cur.qx, cur.qy, cur.qz) <= cur.qr;
                h |= sql2_3(x1.z, y1.z, z1.z, cur.qx, cur.qy, cur.qz) <= cur.qr;
                h |= sql2_3(x1.w, y1.w, z1.w, cur.qx, cur.qy, cur.qz) <= cur.qr;
                if (h && cur.act) flags[cur.o] = 1u;
                cur = nxt;
            }
            wave_lds_sync_();
            out[q] = out[q] || (inb[q] && flags[lane] != 0u);
            wave_lds_sync_();  // (the row is cleared again for the next query)
        }
    }

    __device__ __forceinline__ bool
    capt_collides(env_cptr D, const uint32_t ci, lds_cptr planes_lds, const uint32_t n_lds, lds_u32 *flags, float x, float y,
                  float z, float r, bool active)
    {
        const float xs[1] = {x}, ys[1] = {y}, zs[1] = {z}, rs[1] = {r};
        const bool act[1] = {active};
        bool out[1];
        capt_collides_q<1>(D, ci, planes_lds, n_lds, flags, xs, ys, zs, rs, act, out);
        return out[0];
    }

    // MVT::collides (collision/mvt.hh:204-279) == one lane of collides_simd (mvt.hh:282-403; lanes are
    // independent there, the 8-point chunks only add +inf padding that never collides).  Per lane: global box
    // test, the <= 3 x 3 x 3 cells around the centre (grid_query_radius is clamped to one cell, which is what makes
    // the answer structure-dependent for radii above r_max), voxel box cull, then the voxel's points with `<=`.
    // Float -> uint16 casts apply to in-range values as written; where the reference's cast is undefined
    // (negative upper bound, outside the global-box pre-test only for radii > 1 cell) the range is empty.
    __device__ __forceinline__ bool
    mvt_collides(env_cptr D, const uint32_t mi, float x, float y, float z, float r, bool active)
    {
        const float qr = r + D->mvt[mi].r_point;
        const float qr2 = qr * qr;
        bool in = active;
        in = in && !(x + qr < D->mvt[mi].gmin[0] || x - qr > D->mvt[mi].gmax[0]);
        in = in && !(y + qr < D->mvt[mi].gmin[1] || y - qr > D->mvt[mi].gmax[1]);
        in = in && !(z + qr < D->mvt[mi].gmin[2] || z - qr > D->mvt[mi].gmax[2]);
        if (!wave_any(in)) return false;

        const float isf = D->mvt[mi].inv_scale;
        const uint32_t gw = D->mvt[mi].grid_width;
        const float top = (float) (gw - 1);
        const float gqr = fminf(1.0f, qr * isf);
        const float gx = (x - D->mvt[mi].ws_min[0]) * isf, gy = (y - D->mvt[mi].ws_min[1]) * isf,
                    gz = (z - D->mvt[mi].ws_min[2]) * isf;
        const float bx = fminf(top, gx + gqr), by = fminf(top, gy + gqr), bz = fminf(top, gz + gqr);
        in = in && !(bx < 0.0f || by < 0.0f || bz < 0.0f);
        const uint32_t x0 = (uint32_t) (uint16_t) fmaxf(0.0f, gx - gqr), y0 = (uint32_t) (uint16_t) fmaxf(0.0f, gy - gqr),
                       z0 = (uint32_t) (uint16_t) fmaxf(0.0f, gz - gqr);
        const uint32_t x1 = (uint32_t) (uint16_t) fmaxf(bx, 0.0f), y1 = (uint32_t) (uint16_t) fmaxf(by, 0.0f),
                       z1 = (uint32_t) (uint16_t) fmaxf(bz, 0.0f);
        // flattened walk over this lane's cell range (x outer, z inner, as the reference)
        const uint32_t ny = (y1 >= y0) ? y1 - y0 + 1 : 0, nz = (z1 >= z0) ? z1 - z0 + 1 : 0;
        const uint32_t nx = (x1 >= x0) ? x1 - x0 + 1 : 0;
        const uint32_t n_cells = in ? nx * ny * nz : 0u;
        const gu_cptr cells = (gu_cptr) D->mvt[mi].cells;
        const gf_cptr boxes = (gf_cptr) D->mvt[mi].vox_bbox;
        const gu_cptr offs = (gu_cptr) D->mvt[mi].vox_offset;
        const gf_cptr px = (gf_cptr) D->mvt[mi].px, py = (gf_cptr) D->mvt[mi].py, pz = (gf_cptr) D->mvt[mi].pz;
        bool hit = false;
        for (uint32_t c = 0; wave_any(!hit && c < n_cells); ++c)
        {
            uint32_t i = 0, end = 0;
            if (!hit && c < n_cells)
            {
                const uint32_t cz = c % nz, cy = (c / nz) % ny, cx = c / (nz * ny);
                const uint32_t vi = cells[((size_t) (x0 + cx) * gw + (y0 + cy)) * gw + (z0 + cz)];
                if (vi != 0xffffffffu)
                {
                    const gf_cptr bb = boxes + 6 * (size_t) vi;
                    const bool cull = x + qr < bb[0] || x - qr > bb[3] || y + qr < bb[1] || y - qr > bb[4] ||
                                      z + qr < bb[2] || z - qr > bb[5];
                    if (!cull)
                    {
                        i = offs[vi];
                        end = offs[vi + 1];
                    }
                }
            }
            while (wave_any(i < end))
            {
                if (i < end)
                {
                    const float dx = x - px[i], dy = y - py[i], dz = z - pz[i];
                    if (dx * dx + dy * dy + dz * dz <= qr2)
                    {
                        hit = true;
                        i = end;
                    }
                    else
                        ++i;
                }
            }
        }
        return hit;
    }

    // Values that are the same in every lane but reach a non-inlined function in VGPRs (the calling convention
    // passes arguments in vector registers): re-materialise them as scalars so that loads through them are
    // s_load and loops over them are scalar loops instead of exec-masked vector loops.
    __device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
    __device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); }
    __device__ __forceinline__ uint64_t uniform(uint64_t v)
    {
        const uint32_t lo = uniform((uint32_t) v), hi = uniform((uint32_t) (v >> 32));
        return ((uint64_t) hi << 32) | lo;
    }
    __device__ __forceinline__ env_cptr uniform(env_cptr p) { return (env_cptr) uniform((uint64_t) p); }
    __device__ __forceinline__ lds_cptr uniform(lds_cptr p)
    {
        const uint32_t a = (uint32_t) (uintptr_t) p;  // LDS addresses are 32-bit
        return (lds_cptr) (uintptr_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) a);
    }

    // ------------------------------------------------------------------------
    // wave-level helpers for the counted primitive loops
    // ------------------------------------------------------------------------
    // max over the wavefront of a NON-NEGATIVE float (compared through its bit pattern), result wave-uniform.
    // 4 DPP steps (quad xor 1, quad xor 2, half-row mirror, row mirror) leave each 16-lane row holding its
    // maximum; 4 v_readlane + scalar max finish.  No LDS traffic.
    __device__ __forceinline__ float wave_max_nonneg(float v)
    {
        uint32_t u = f2u(v);
        u = max(u, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) u, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
        u = max(u, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) u, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
        u = max(u, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) u, 0x141, 0xF, 0xF, false));  // row_half_mirror
        u = max(u, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) u, 0x140, 0xF, 0xF, false));  // row_mirror
        const uint32_t a = (uint32_t) __builtin_amdgcn_readlane((int) u, 0);
        const uint32_t b = (uint32_t) __builtin_amdgcn_readlane((int) u, 16);
        const uint32_t c = (uint32_t) __builtin_amdgcn_readlane((int) u, 32);
        const uint32_t d = (uint32_t) __builtin_amdgcn_readlane((int) u, 48);
        return u2f(max(max(a, b), max(c, d)));
    }

    // How many leading entries of a min_distance-sorted list can matter to any lane of the wave:
    // #{ p : neg(md[p] - ext_wave) }  (the predicate is monotone along the sorted list).
    __device__ __forceinline__ uint32_t live_prefix(lds_cptr md, const uint32_t n, const float ext_wave)
    {
        const uint32_t lane = __lane_id();
        uint32_t count = 0;
        for (uint32_t base = 0; base < n; base += kWave)
        {
            const uint32_t i = base + lane;
            const bool p = (i < n) && neg(md[i < n ? i : 0] - ext_wave);
            const uint32_t c = (uint32_t) __popcll(__ballot(p));
            count += c;
            if (c < (uint32_t) kWave) break;
        }
        return count;
    }

    // Where the wave-uniform primitive records are read from inside the counted loops.
    //   VMV_PRIMS_SCALAR = 1: straight from the environment block through the scalar cache (s_load_dwordx4 into
    //       SGPRs that the VALU consumes directly) - keeps the LDS pipe free for the per-lane traffic (sphere slab,
    //       re-dealt items, min_distance prefix counts, candidate gathers, CAPT split planes);
    //   VMV_PRIMS_SCALAR = 0: from the LDS copy of the block (64-lane broadcast ds_read_b128).
#ifndef VMV_PRIMS_SCALAR
#define VMV_PRIMS_SCALAR 1
#ifndef VMV_ABLATE_SELF
#define VMV_ABLATE_SELF 0  // measurement aid (tools only): 1 = no dense pair tests, 2 = gates only, 4 = no sparse groups, 8 = no dense groups at all,
                           // 16 = clearance tables ignored (every bit 1; right answers)
#endif
#endif
#if VMV_PRIMS_SCALAR
    using rec_cptr = const __attribute__((address_space(4))) float *;
    typedef __attribute__((address_space(4))) v4f c_v4f;
    __device__ __forceinline__ v4f rec_load4(rec_cptr p) { return *(const c_v4f *) p; }
#define VMV_REC_BASE(E, D) ((rec_cptr) (D).prims)
#else
    using rec_cptr = lds_cptr;
#define VMV_REC_BASE(E, D) ((E).lds)
#endif

    // ------------------------------------------------------------------------
    // one primitive against one sphere: the reference's signed test value (collision iff its sign bit is set),
    // the primitive's min_distance, and the length the candidate margin scales with.
    // ------------------------------------------------------------------------
    enum PrimType
    {
        kSphere = 0,
        kCapsule = 1,
        kZCapsule = 2,
        kCuboid = 3,
        kZCuboid = 4
    };
    // what an environment-kernel variant compiles in: bits 0..4 = primitive lists (PrimType), bit 5 = heightfields and
    // point clouds
    constexpr int kEnvFull = 63, kEnvPrims = 31, kEnvZOnly = (1 << kSphere) | (1 << kZCapsule) | (1 << kZCuboid);
    // point clouds and nothing else (no primitive, heightfield or MVT): no list code at all, and the paired walk
    // lists the lanes whose cloud query hit without a gate call (the launchers pick it: clouds_only)
    constexpr int kEnvClouds = 32;
    template <int T>
    struct PrimTraits;
    template <> struct PrimTraits<kSphere> { static constexpr int rec = kSphereRec; };
    template <> struct PrimTraits<kCapsule> { static constexpr int rec = kCapsuleRec; };
    template <> struct PrimTraits<kZCapsule> { static constexpr int rec = kZCapsuleRec; };
    template <> struct PrimTraits<kCuboid> { static constexpr int rec = kCuboidRec; };
    template <> struct PrimTraits<kZCuboid> { static constexpr int rec = kZCuboidRec; };

    __device__ __forceinline__ v4f load4(lds_cptr p) { return lds_load4(p); }
#if VMV_PRIMS_SCALAR
    __device__ __forceinline__ v4f load4(rec_cptr p) { return rec_load4(p); }
#endif

    template <int T, typename PTR>
    __device__ __forceinline__ void
    prim_eval(PTR rec, float x, float y, float z, float r, float rsq, float &v, float &md, float &reach)
    {
        if constexpr (T == kSphere)
        {
            const v4f a = load4(rec);  // x y z r | min_d
            md = rec[4];
            reach = a.w + r;  // collision/sphere_sphere.hh:9-23: rs = ar + br
            v = sql2_3(a.x, a.y, a.z, x, y, z) - reach * reach;
        }
        else if constexpr (T == kCapsule)
        {
            const v4f a = load4(rec);      // x1 y1 z1 xv
            const v4f b = load4(rec + 4);  // yv zv r rdv
            md = rec[8];
            // collision/sphere_capsule.hh:8-23
            const float dot = dot3(x - a.x, y - a.y, z - a.z, a.w, b.x, b.y);
            const float cdf = vclamp(dot * b.w, 0.F, 1.F);
            const float sum = sql2_3(x, y, z, a.x + a.w * cdf, a.y + b.x * cdf, a.z + b.y * cdf);
            reach = r + b.z;
            v = sum - reach * reach;
        }
        else if constexpr (T == kZCapsule)
        {
            const v4f a = load4(rec);      // x1 y1 z1 zv
            const v4f b = load4(rec + 4);  // r rdv min_d 0
            md = b.z;
            // collision/sphere_capsule.hh:31-45
            const float dot = (z - a.z) * a.w;
            const float cdf = vclamp(dot * b.y, 0.F, 1.F);
            const float sum = sql2_3(x, y, z, a.x, a.y, a.z + a.w * cdf);
            reach = r + b.x;
            v = sum - reach * reach;
        }
        else if constexpr (T == kCuboid)
        {
            const v4f a = load4(rec);       // x y z a1x
            const v4f b = load4(rec + 4);   // a1y a1z a2x a2y
            const v4f c = load4(rec + 8);   // a2z a3x a3y a3z
            const v4f d = load4(rec + 12);  // r1 r2 r3 min_d
            md = d.w;
            // collision/sphere_cuboid.hh:8-27
            const float xs = x - a.x, ys = y - a.y, zs = z - a.z;
            const float a1 = x86_max(vabs(dot3(a.w, b.x, b.y, xs, ys, zs)) - d.x, 0.f);
            const float a2 = x86_max(vabs(dot3(b.z, b.w, c.x, xs, ys, zs)) - d.y, 0.f);
            const float a3 = x86_max(vabs(dot3(c.y, c.z, c.w, xs, ys, zs)) - d.z, 0.f);
            reach = r;
            v = dot3(a1, a2, a3, a1, a2, a3) - rsq;
        }
        else
        {
            const v4f a = load4(rec);      // x y z a1x
            const v4f b = load4(rec + 4);  // a1y a2x a2y r1
            const v4f c = load4(rec + 8);  // r2 r3 min_d 0
            md = c.z;
            // collision/sphere_cuboid.hh:35-52
            const float xs = x - a.x, ys = y - a.y, zs = z - a.z;
            const float a1 = x86_max(vabs((a.w * xs) + (b.x * ys)) - b.w, 0.f);
            const float a2 = x86_max(vabs((b.y * xs) + (b.z * ys)) - c.x, 0.f);
            const float a3 = x86_max(vabs(zs) - c.y, 0.f);
            reach = r;
            v = dot3(a1, a2, a3, a1, a2, a3) - rsq;
        }
    }

    // Candidate margin.  A fine sphere lies inside its link's bounding sphere (the bounding sphere is the smallest
    // ball enclosing the link's spheres; tools/robot_trace.py asserts the enclosure in the link frame, and rigid fp32
    // FK preserves it to ~1e-6 m).  Every primitive test is `distance(centre, primitive)^2 - reach^2` with a
    // 1-Lipschitz distance, so a primitive a fine sphere can collide with satisfies, for the bounding sphere,
    // distance < reach_bounding + 1e-6.  The gate therefore records as candidates all primitives with
    // distance < reach + kCandidateMargin (v < 2 * margin * reach + margin^2), two orders of magnitude wider than any
    // fp32 error in FK or in the tests, and the fine spheres evaluate the reference's exact predicates on candidates
    // only.  This prunes work; it cannot change an answer.
    constexpr float kCandidateMargin = 1e-4f;
    constexpr int kMaskWords = 4;  // 32-primitive candidate words per lane kept in LDS (environments with more use full loops)

    // the bits of a candidate word that belong to a list of n primitives: all 32 unless the list is shorter than a word
    // (only such a list may share its word, EnvDev::wshift_*)
    __device__ __forceinline__ uint32_t list_word_mask(const uint32_t n) { return n < 32u ? (1u << n) - 1u : 0xffffffffu; }

    // One sorted list, full counted loop over the live prefix (records through the scalar cache).  MASK: also
    // write this lane's candidate words for the list to mask_lane[(word) * 64].
    template <int G, int T, bool MASK>
    __device__ __forceinline__ void list_full(const EnvView &E, const uint32_t n, const uint32_t off, const uint32_t off_md,
                                              const uint32_t wbase, const uint32_t shift, float x, float y, float z, float r,
                                              float rsq, float ext, float ext_wave, bool &hit, lds_u32 *mask_lane)
    {
        if (n == 0) return;
        constexpr int REC = PrimTraits<T>::rec;
        const env_cptr Dp = E.dev;
        // candidates need the loop to cover what any FINE sphere of the link may still test: a hair beyond the wave's
        // largest bounding max_extent (fine max_extent <= bounding max_extent + ~1e-6)
        const uint32_t n_live = live_prefix(E.lds + off_md, n, MASK ? ext_wave + 1e-3f : ext_wave);
        rec_cptr rec = VMV_REC_BASE(E, (*Dp)) + off;
        if constexpr (!MASK)
        {
#pragma unroll 2
            for (uint32_t i = 0; i < n_live; ++i, rec += REC)
            {
                float v, md, reach;
                prim_eval<T>(rec, x, y, z, r, rsq, v, md, reach);
                hit |= neg(md - ext) && neg(v);
            }
        }
        else
        {
            for (uint32_t b = 0; b < n_live; b += 32)
            {
                uint32_t m = 0;
                const uint32_t e = (b + 32 < n_live) ? b + 32 : n_live;
                for (uint32_t i = b; i < e; ++i, rec += REC)
                {
                    float v, md, reach;
                    prim_eval<T>(rec, x, y, z, r, rsq, v, md, reach);
                    hit |= neg(md - ext) && neg(v);
                    const float tau = reach * (2.0f * kCandidateMargin) + kCandidateMargin * kCandidateMargin;
                    m |= (v < tau) ? (1u << (i - b)) : 0u;
                }
                // (shift != 0: a one-word list in a shared word, the gate cleared the words before the lists ran)
                if (shift == 0u) mask_lane[(wbase + (b >> 5)) * kWave] = m;
                else mask_lane[wbase * kWave] |= m << shift;
            }
        }
    }

    // One sorted list, candidates only: mask_src points at the candidate words of the lane this item works for.
    template <int T, bool SHARED>  // SHARED: lists may share candidate words (every variant but the three-list one)
    __device__ __forceinline__ void list_masked(const EnvView &E, const uint32_t n, const uint32_t off, const uint32_t wbase,
                                                const uint32_t shift, float x, float y, float z, float r, float rsq, float ext,
                                                bool active, bool &hit, const lds_u32 *mask_src)
    {
        if (n == 0) return;
        constexpr int REC = PrimTraits<T>::rec;
        const uint32_t words = (n + 31) / 32;
        for (uint32_t w = 0; w < words; ++w)
        {
            uint32_t m = active ? mask_src[(wbase + w) * kWave] : 0u;
            if constexpr (SHARED) m = (m >> shift) & list_word_mask(n);  // (this list's bits; shift 0, all bits for longer lists)
            // branch-free body: a lane that has run out evaluates record 0 and discards the result (straight-line code
            // instead of an exec-masked region with its copies of every loop-carried value)
            while (wave_any(m != 0u))
            {
                const bool live = m != 0u;
                const uint32_t bit = live ? (uint32_t) __ffs((int) m) - 1u : 0u;
                m &= m - 1u;  // 0 stays 0
                lds_cptr rec = E.lds + off + (w * 32u + bit) * REC;  // per-lane record: LDS gather
                float v, md, reach;
                prim_eval<T>(rec, x, y, z, r, rsq, v, md, reach);
                hit |= live && neg(md - ext) && neg(v);
            }
        }
    }

    // sphere_heightfield (collision/sphere_heightfield.hh:8-31), per lane; true = collides.  The reference clamps the
    // cell index to [0, xd] x [0, yd], one past the image on both axes, and gathers up to xd + 1 floats beyond the data
    // (undefined behaviour); here such an index reads the last pixel.  Every in-bounds index is the reference's.
    __device__ __forceinline__ bool heightfield_collides(const env_cptr Dp, const uint32_t hi, float x, float y, float z, float r)
    {
        const HeightFieldDev __attribute__((address_space(4))) *a = &Dp->heightfield[hi];
        const float xo = a->x - x, yo = a->y - y;
        const float xs = floorf(vclamp(a->xs * xo + a->xd2, 0.F, a->xd));
        const float ys = floorf(vclamp(a->ys * yo + a->yd2, 0.F, a->yd));
        const float index = ys * a->xd + xs;
        int idx = (int) rintf(index);  // _mm256_cvtps_epi32 (vector/avx.hh:629)
        idx = idx < 0 ? 0 : idx;
        const uint32_t u = (uint32_t) idx > a->last ? a->last : (uint32_t) idx;
        const float zh = a->data[u];
        const float zhs = a->zs * zh + a->z;
        return neg(z - r - zhs);
    }

    // The list part of the environment header (counts, block offsets, candidate-word bases), fetched ONCE per gate /
    // fine call and pinned in SGPRs.  Read where they are used, each field was its own scalar load behind the branch
    // that first needs it, and — the wave-level LDS hand-offs between re-dealt rounds are memory fences to the compiler —
    // fetched again in every round: six dependent scalar-cache round trips per round of a kernel that is latency-bound.
    // (only the three-list variant works from the pinned copy: with all five lists compiled in, fifteen more live SGPRs
    // push the gate / fine bodies into 65 - 90 SGPR spills per kernel, so those variants read the fields where they use them)
#define VMV_HN(T, name) ((V == kEnvZOnly) ? H.n[T] : D.n_##name)
#define VMV_HOFF(T, name) ((V == kEnvZOnly) ? H.off[T] : D.off_##name)
#define VMV_HWB(T, name) ((V == kEnvZOnly) ? H.wbase[T] : D.wbase_##name)
#define VMV_HSH(T, name) ((V == kEnvZOnly) ? 0u : D.wshift_##name)
    struct ListHdr
    {
        uint32_t n[5], off[5], wbase[5];  // indexed by PrimType; lists a variant does not compile in stay 0
    };
    template <int V, bool PIN = true>
    __device__ __forceinline__ ListHdr load_list_hdr(const env_cptr D)
    {
        ListHdr H;
#pragma unroll
        for (int t = 0; t < 5; ++t) H.n[t] = H.off[t] = H.wbase[t] = 0u;
        if constexpr (V == kEnvZOnly)
        {
            H.n[kSphere] = D->n_sphere, H.off[kSphere] = D->off_sphere, H.wbase[kSphere] = D->wbase_sphere;
            H.n[kZCapsule] = D->n_zcapsule, H.off[kZCapsule] = D->off_zcapsule, H.wbase[kZCapsule] = D->wbase_zcapsule;
            H.n[kZCuboid] = D->n_zcuboid, H.off[kZCuboid] = D->off_zcuboid, H.wbase[kZCuboid] = D->wbase_zcuboid;
#ifndef VMV_NO_HDR_HOIST  // (A/B knob, tools/build_variant.py: without the pin every field is fetched where it is used;
                          // measured on the Panda bench: environment kernel 0.1559 -> 0.1534 ms, profiles/r03_hdr_hoist_ab.txt)
            if constexpr (PIN)
                asm volatile("" ::"s"(H.n[kSphere]), "s"(H.n[kZCapsule]), "s"(H.n[kZCuboid]), "s"(H.off[kSphere]), "s"(H.off[kZCapsule]),
                             "s"(H.off[kZCuboid]), "s"(H.wbase[kSphere]), "s"(H.wbase[kZCapsule]), "s"(H.wbase[kZCuboid]));
#endif
        }
        return H;
    }

    // sphere_environment_in_collision (collision/validity.hh:47-158) for one robot sphere, per lane.
    //  * returns this lane's own "hits something" flag; the caller folds it over the rake with group_any.
    //  * the sorted early-break is rake-wide in the reference (all 8 lanes must agree).  Lists are sorted
    //    by min_distance, so "all lanes have min_distance - max_extent >= 0" is the same predicate
    //    evaluated on the rake's largest max_extent: ext = group_max(max_extent).
    //  * max_extent uses the correctly rounded sqrt (the reference's v*rsqrt_ps(v) is vendor-defined).
    //  * MODE 0 (plain) / 1 (gate: also record candidate words): counted loops over the live prefix for the
    //    largest max_extent in the wave; each lane still applies its own break predicate.
    //    MODE 2 (fine): only the candidates the lane's bounding sphere recorded (see kCandidateMargin).
    //  * `active` only prunes work: inactive lanes do not extend the trip counts and report no hit.
    //  * V selects what is compiled in (kEnvFull / kEnvPrims / kEnvZOnly below): environments made of primitives only
    //    run kernels without the heightfield / CAPT / MVT tail, and those without general cuboids and capsules also
    //    without those two lists (their code costs registers even when the lists are empty: 6-8 % and another 5 % on
    //    the environment kernel).
    //  * CAPT = false: the point clouds are left out (the caller holds their answer for this sphere: capt_gate_pair).
    template <int G, int MODE, int V = kEnvFull, bool CAPT = true>
    __device__ __forceinline__ bool
    env_hit(const EnvView &E, const ListHdr &H, float x, float y, float z, float r, bool active, lds_u32 *mask_src)
    {
        const env_cptr Dp = E.dev;
#define D (*Dp)
        const float ext = group_max<G>(sqrtf(dot3(x, y, z, x, y, z)) + r);
        const float rsq = r * r;
        bool hit = false;
        if constexpr (MODE == 2)
        {
            lds_u32 *const mask = mask_src;
            if constexpr ((V >> kSphere) & 1) list_masked<kSphere, V != kEnvZOnly>(E, VMV_HN(kSphere, sphere), VMV_HOFF(kSphere, sphere), VMV_HWB(kSphere, sphere), VMV_HSH(kSphere, sphere), x, y, z, r, rsq, ext, active, hit, mask);
            if constexpr ((V >> kCapsule) & 1) list_masked<kCapsule, V != kEnvZOnly>(E, VMV_HN(kCapsule, capsule), VMV_HOFF(kCapsule, capsule), VMV_HWB(kCapsule, capsule), VMV_HSH(kCapsule, capsule), x, y, z, r, rsq, ext, active, hit, mask);
            if constexpr ((V >> kZCapsule) & 1) list_masked<kZCapsule, V != kEnvZOnly>(E, VMV_HN(kZCapsule, zcapsule), VMV_HOFF(kZCapsule, zcapsule), VMV_HWB(kZCapsule, zcapsule), VMV_HSH(kZCapsule, zcapsule), x, y, z, r, rsq, ext, active, hit, mask);
            if constexpr ((V >> kCuboid) & 1) list_masked<kCuboid, V != kEnvZOnly>(E, VMV_HN(kCuboid, cuboid), VMV_HOFF(kCuboid, cuboid), VMV_HWB(kCuboid, cuboid), VMV_HSH(kCuboid, cuboid), x, y, z, r, rsq, ext, active, hit, mask);
            if constexpr ((V >> kZCuboid) & 1) list_masked<kZCuboid, V != kEnvZOnly>(E, VMV_HN(kZCuboid, zcuboid), VMV_HOFF(kZCuboid, zcuboid), VMV_HWB(kZCuboid, zcuboid), VMV_HSH(kZCuboid, zcuboid), x, y, z, r, rsq, ext, active, hit, mask);
        }
        else
        {
            constexpr bool MASK = (MODE == 1);
            const float ext_wave = wave_max_nonneg(active ? ext : 0.0f);
            if constexpr ((V >> kSphere) & 1) list_full<G, kSphere, MASK>(E, VMV_HN(kSphere, sphere), VMV_HOFF(kSphere, sphere), D.off_md_sphere, VMV_HWB(kSphere, sphere), VMV_HSH(kSphere, sphere), x, y, z, r, rsq, ext,
                                        ext_wave, hit, mask_src);
            if constexpr ((V >> kCapsule) & 1) list_full<G, kCapsule, MASK>(E, VMV_HN(kCapsule, capsule), VMV_HOFF(kCapsule, capsule), D.off_md_capsule, VMV_HWB(kCapsule, capsule), VMV_HSH(kCapsule, capsule), x, y, z, r, rsq,
                                         ext, ext_wave, hit, mask_src);
            if constexpr ((V >> kZCapsule) & 1) list_full<G, kZCapsule, MASK>(E, VMV_HN(kZCapsule, zcapsule), VMV_HOFF(kZCapsule, zcapsule), D.off_md_zcapsule, VMV_HWB(kZCapsule, zcapsule), VMV_HSH(kZCapsule, zcapsule), x, y, z, r,
                                          rsq, ext, ext_wave, hit, mask_src);
            if constexpr ((V >> kCuboid) & 1) list_full<G, kCuboid, MASK>(E, VMV_HN(kCuboid, cuboid), VMV_HOFF(kCuboid, cuboid), D.off_md_cuboid, VMV_HWB(kCuboid, cuboid), VMV_HSH(kCuboid, cuboid), x, y, z, r, rsq, ext,
                                        ext_wave, hit, mask_src);
            if constexpr ((V >> kZCuboid) & 1) list_full<G, kZCuboid, MASK>(E, VMV_HN(kZCuboid, zcuboid), VMV_HOFF(kZCuboid, zcuboid), D.off_md_zcuboid, VMV_HWB(kZCuboid, zcuboid), VMV_HSH(kZCuboid, zcuboid), x, y, z, r, rsq,
                                         ext, ext_wave, hit, mask_src);
        }
        hit = hit && active;
        if constexpr (((V >> 5) & 1) == 0) return hit;
        for (uint32_t hi = 0; hi < D.n_heightfield; ++hi)  // validity.hh:131-137
        {
            if (!wave_any(active && !hit)) break;
            hit |= active && heightfield_collides(Dp, hi, x, y, z, r);
        }
        for (uint32_t ci = 0; CAPT && ci < D.n_capt; ++ci)
        {
            const bool act = active && !hit;
            if (!wave_any(act)) break;
            hit |= capt_collides(Dp, ci, E.lds + D.n_floats, (ci == 0) ? E.capt0_planes_in_lds : 0u, capt_flag_row(E), x, y, z, r, act);
        }
        for (uint32_t mi = 0; mi < D.n_mvt; ++mi)  // validity.hh:149-155
        {
            const bool act = active && !hit;
            if (!wave_any(act)) break;
            hit |= mvt_collides(Dp, mi, x, y, z, r, act);
        }
        return hit;
#undef D
    }
    // (callers outside the hot loops: the header is fetched here)
    template <int G, int MODE, int V = kEnvFull, bool CAPT = true>
    __device__ __forceinline__ bool
    env_hit(const EnvView &E, float x, float y, float z, float r, bool active, lds_u32 *mask)
    {
        const ListHdr H = load_list_hdr<V, false>(E.dev);
        return env_hit<G, MODE, V, CAPT>(E, H, x, y, z, r, active, mask);
    }

    // Contact report (Robot::fkcc_debug -> sphere_environment_get_collisions, collision/validity.hh:161-256): which
    // primitives one sphere collides with, as bits in the candidate-word layout (wbase_* per list, 32 primitives per
    // word, sorted-list positions) plus one word for the heightfields.  Replicated-configuration semantics (G = 1).
    template <int T>
    __device__ __forceinline__ void list_hit_words(const EnvView &E, const uint32_t n, const uint32_t off, const uint32_t wbase,
                                                   float x, float y, float z, float r, float rsq, float ext, uint32_t *words)
    {
        constexpr int REC = PrimTraits<T>::rec;
        for (uint32_t i = 0; i < n; ++i)
        {
            float v, md, reach;
            prim_eval<T>(E.lds + off + i * REC, x, y, z, r, rsq, v, md, reach);
            if (!neg(md - ext)) break;  // sorted: nothing behind it can be reached either (validity.hh:179-183)
            if (neg(v)) words[wbase + (i >> 5)] |= 1u << (i & 31u);
        }
    }
    constexpr int kReportWords = 8;  // 32-primitive words of a contact report (its own layout: lists back to back)
    __device__ __forceinline__ void sphere_hit_words(const EnvView &E, float x, float y, float z, float r,
                                                     uint32_t (&words)[kReportWords + 1])
    {
        const env_cptr Dp = E.dev;
#define D (*Dp)
        const float ext = sqrtf(dot3(x, y, z, x, y, z)) + r, rsq = r * r;
#pragma unroll
        for (int w = 0; w <= kReportWords; ++w) words[w] = 0u;
        uint32_t base = 0;
        list_hit_words<kSphere>(E, D.n_sphere, D.off_sphere, base, x, y, z, r, rsq, ext, words);
        base += (D.n_sphere + 31u) / 32u;
        list_hit_words<kCapsule>(E, D.n_capsule, D.off_capsule, base, x, y, z, r, rsq, ext, words);
        base += (D.n_capsule + 31u) / 32u;
        list_hit_words<kZCapsule>(E, D.n_zcapsule, D.off_zcapsule, base, x, y, z, r, rsq, ext, words);
        base += (D.n_zcapsule + 31u) / 32u;
        list_hit_words<kCuboid>(E, D.n_cuboid, D.off_cuboid, base, x, y, z, r, rsq, ext, words);
        base += (D.n_cuboid + 31u) / 32u;
        list_hit_words<kZCuboid>(E, D.n_zcuboid, D.off_zcuboid, base, x, y, z, r, rsq, ext, words);
        for (uint32_t hi = 0; hi < D.n_heightfield; ++hi)
            if (heightfield_collides(Dp, hi, x, y, z, r)) words[kReportWords] |= 1u << hi;
#undef D
    }

    // One sorted list in the gate pass, driven by the broad-phase grid: this lane evaluates the reference's exact
    // predicates on the candidates of its own cell only, and records the fine-phase candidates (kCandidateMargin)
    // among them.  Lanes walk their own candidate bits; the loop runs until the busiest lane is done.
    template <int T, bool SHARED>
    __device__ __forceinline__ void list_grid(const EnvView &E, const uint32_t n, const uint32_t off, const uint32_t wbase,
                                              const uint32_t shift, const uint32_t (&cw)[kMaskWords], float x, float y, float z,
                                              float r, float rsq, float ext, bool &hit, lds_u32 *mask_lane)
    {
        if (n == 0) return;
        constexpr int REC = PrimTraits<T>::rec;
        const uint32_t words = (n + 31) / 32;
        for (uint32_t w = 0; w < words; ++w)
        {
            // (wave-uniform index into the preloaded words: selects, not a scratch array)
            const uint32_t k = wbase + w;
            uint32_t m = (k == 0u) ? cw[0] : (k == 1u) ? cw[1] : (k == 2u) ? cw[2] : cw[3];
            if constexpr (SHARED) m = (m >> shift) & list_word_mask(n);  // (shared word: this list's bits)
            uint32_t fine = 0u;
            while (wave_any(m != 0u))  // branch-free body, see list_masked
            {
                const bool live = m != 0u;
                const uint32_t bit = live ? (uint32_t) __ffs((int) m) - 1u : 0u;
                m &= m - 1u;
                lds_cptr rec = E.lds + off + (w * 32u + bit) * REC;
                float v, md, reach;
                prim_eval<T>(rec, x, y, z, r, rsq, v, md, reach);
                hit |= live && neg(md - ext) && neg(v);
                const float tau = reach * (2.0f * kCandidateMargin) + kCandidateMargin * kCandidateMargin;
                fine |= (live && v < tau) ? (1u << bit) : 0u;
            }
            if (VMV_ABLATE_ENV == 4 || VMV_ABLATE_ENV == 5) fine = 0u;
            // (the first list of a word — shift 0 — stores, so the word needs no clearing; the lists that share it OR in)
            if (shift == 0u) mask_lane[(wbase + w) * kWave] = fine;
            else mask_lane[wbase * kWave] |= fine << shift;
        }
    }

    // Gate pass of one bounding sphere through the broad-phase grid (same answer as env_hit<G, 1>).
    template <int G, int V = kEnvFull, bool CAPT = true>
    __device__ __forceinline__ bool
    env_hit_grid(const EnvView &E, const ListHdr &H, const uint32_t cls, float x, float y, float z, float r, bool active,
                 lds_u32 *mask_lane)
    {
        const env_cptr Dp = E.dev;
#define D (*Dp)
        const GridDev __attribute__((address_space(4))) *Gd = &Dp->grid[cls];
        // the grid's header in one go, pinned (as the list header: otherwise one scalar round trip per field, in sequence)
        const float inv_cell = Gd->inv_cell, o0 = Gd->origin[0], o1 = Gd->origin[1], o2 = Gd->origin[2];
        const uint32_t d0 = Gd->dims[0], d1 = Gd->dims[1], d2 = Gd->dims[2], gw = D.grid_words;
        const gu_cptr cells = (gu_cptr) Gd->cells;
        if constexpr (V == kEnvZOnly)
            asm volatile("" ::"s"(inv_cell), "s"(o0), "s"(o1), "s"(o2), "s"(d0), "s"(d1), "s"(d2), "s"(gw), "s"(cells));
        const float ext = group_max<G>(sqrtf(dot3(x, y, z, x, y, z)) + r);
        const float rsq = r * r;
        // this lane's cell; outside the grid box nothing can be touched (vmv_grid_build.h)
        const float fx = (x - o0) * inv_cell, fy = (y - o1) * inv_cell, fz = (z - o2) * inv_cell;
        const bool inside = active && fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx < (float) d0 && fy < (float) d1 &&
                            fz < (float) d2;
        const uint32_t ix = inside ? (uint32_t) fx : 0u, iy = inside ? (uint32_t) fy : 0u, iz = inside ? (uint32_t) fz : 0u;
        const gu_cptr cell = cells + ((size_t) (ix * d1 + iy) * d2 + iz) * gw;
        bool hit = false;
        // all candidate words of the cell at once (independent loads, one memory latency per gate instead of one per
        // list and word: the walks below were waiting on these loads, not on arithmetic)
        static_assert(kMaskWords == 4, "list_grid selects among four preloaded words");
        uint32_t cw[kMaskWords];
#pragma unroll
        for (int w = 0; w < kMaskWords; ++w) cw[w] = (inside && (uint32_t) w < gw) ? cell[w] : 0u;
        if (VMV_ABLATE_ENV == 3) return inside && cw[0] == 0x12345u;  // measurement aid: gate overhead without the walks
        if constexpr ((V >> kSphere) & 1) list_grid<kSphere, V != kEnvZOnly>(E, VMV_HN(kSphere, sphere), VMV_HOFF(kSphere, sphere), VMV_HWB(kSphere, sphere), VMV_HSH(kSphere, sphere), cw, x, y, z, r, rsq, ext, hit, mask_lane);
        if constexpr ((V >> kCapsule) & 1) list_grid<kCapsule, V != kEnvZOnly>(E, VMV_HN(kCapsule, capsule), VMV_HOFF(kCapsule, capsule), VMV_HWB(kCapsule, capsule), VMV_HSH(kCapsule, capsule), cw, x, y, z, r, rsq, ext, hit, mask_lane);
        if constexpr ((V >> kZCapsule) & 1) list_grid<kZCapsule, V != kEnvZOnly>(E, VMV_HN(kZCapsule, zcapsule), VMV_HOFF(kZCapsule, zcapsule), VMV_HWB(kZCapsule, zcapsule), VMV_HSH(kZCapsule, zcapsule), cw, x, y, z, r, rsq, ext, hit, mask_lane);
        if constexpr ((V >> kCuboid) & 1) list_grid<kCuboid, V != kEnvZOnly>(E, VMV_HN(kCuboid, cuboid), VMV_HOFF(kCuboid, cuboid), VMV_HWB(kCuboid, cuboid), VMV_HSH(kCuboid, cuboid), cw, x, y, z, r, rsq, ext, hit, mask_lane);
        if constexpr ((V >> kZCuboid) & 1) list_grid<kZCuboid, V != kEnvZOnly>(E, VMV_HN(kZCuboid, zcuboid), VMV_HOFF(kZCuboid, zcuboid), VMV_HWB(kZCuboid, zcuboid), VMV_HSH(kZCuboid, zcuboid), cw, x, y, z, r, rsq, ext, hit, mask_lane);
        hit = hit && active;
        if constexpr (((V >> 5) & 1) == 0) return hit;
        for (uint32_t hi = 0; hi < D.n_heightfield; ++hi)  // validity.hh:131-137
        {
            if (!wave_any(active && !hit)) break;
            hit |= active && heightfield_collides(Dp, hi, x, y, z, r);
        }
        for (uint32_t ci = 0; CAPT && ci < D.n_capt; ++ci)
        {
            const bool act = active && !hit;
            if (!wave_any(act)) break;
            hit |= capt_collides(Dp, ci, E.lds + D.n_floats, (ci == 0) ? E.capt0_planes_in_lds : 0u, capt_flag_row(E), x, y, z, r, act);
        }
        for (uint32_t mi = 0; mi < D.n_mvt; ++mi)
        {
            const bool act = active && !hit;
            if (!wave_any(act)) break;
            hit |= mvt_collides(Dp, mi, x, y, z, r, act);
        }
        return hit;
#undef D
    }

    __device__ __forceinline__ void wave_lds_sync() { wave_lds_sync_(); }

    // per-wave LDS scratch behind the slab: lane list [64], hit flags [64], k [4], candidate words [kMaskWords][64]
    constexpr int kScratchWords = 2 * kWave + 4 + kMaskWords * kWave;

    // One link's environment group (robots/panda.hh:5629-6010): `if (hit(bounding)) { any fine sphere hits }`.
    //
    // The link's FINE sphere centres are staged in this wave's LDS slab, slab[(3*s + k) * 65 + lane] (`slab` already
    // points at this lane's column), in chunks of at most kChunk spheres so that the slab stays small enough for 4+
    // waves per SIMD; the bounding sphere goes to the gate in registers.
    //   env_gate   every lane tests its own bounding sphere (lane = configuration); the lanes of the rakes whose
    //              gate fired are listed in LDS.
    //   env_fine   for one staged chunk: only rakes whose gate fired matter, typically a few of the 64 lanes.
    //              Instead of running n rounds with most lanes idle, the (passing lane, fine sphere) pairs are
    //              re-dealt over the 64 lanes, sphere-major (item = s * k + j), each lane fetching "its" sphere
    //              from the slab column of the configuration it now works for.  k is a multiple of G and passing
    //              lanes come in whole rakes, so the 8 lanes of a rake stay adjacent and aligned for the rake-wide
    //              max_extent.  Hits are OR-ed back per configuration through LDS flags.
    //   env_flag   this lane's "some fine sphere of my configuration hit".
    // `active` (rake-uniform) only prunes work.  Tab::radius(i) reads the robot's __constant__ radius table.
    // The point-cloud part of TWO gates at once (generated fkcc_env_paired): the bounding spheres of two consecutive links,
    // both queried against every cloud with their dependent fetches interleaved (capt_collides_q<2>).  Returns this lane's
    // answers: bit 0 = sphere a collides with a cloud, bit 1 = sphere b.  The gates then run with PRE = true: everything
    // but the clouds, OR the bit.  (env_hit asks the clouds only when nothing else hit; asking anyway changes no OR.)
    template <int G, typename Tab>
    __device__ __noinline__ unsigned capt_gate_pair(const EnvView E_, float xa, float ya, float za, const int ra_index_, float xb,
                                                    float yb, float zb, const int rb_index_, const bool active)
    {
        const EnvView E{uniform(E_.dev), uniform(E_.lds), uniform(E_.capt0_planes_in_lds), uniform(E_.radii)};
        const env_cptr Dp = E.dev;
        const uint32_t n_capt = Dp->n_capt;
        if (n_capt == 0u) return 0u;
        const float x[2] = {xa, xb}, y[2] = {ya, yb}, z[2] = {za, zb};
        const float r[2] = {Tab::radius(uniform(ra_index_)), Tab::radius(uniform(rb_index_))};
        bool hit[2] = {false, false};
        for (uint32_t ci = 0; ci < n_capt; ++ci)
        {
            const bool act[2] = {active && !hit[0], active && !hit[1]};
            if (!wave_any(act[0] || act[1])) break;
            bool out[2];
            capt_collides_q<2>(Dp, ci, E.lds + Dp->n_floats, (ci == 0) ? E.capt0_planes_in_lds : 0u, capt_flag_row(E), x, y, z, r,
                               act, out);
            hit[0] |= out[0];
            hit[1] |= out[1];
        }
        return (hit[0] ? 1u : 0u) | (hit[1] ? 2u : 0u);
    }

    template <int G, typename Tab, int V = kEnvFull, bool PRE = false>
    __device__ __noinline__ bool
    env_gate(const EnvView E_, const float bx, const float by, const float bz, lds_ptr scratch_, const int radius_index_,
             const int grid_class_, const bool active,
             const bool pre = false /* PRE: this lane's bounding sphere collides with a point cloud */)
    {
        // (bx, by, bz: this lane's bounding-sphere centre, in registers — it used to travel through the slab's row 0:
        // three LDS stores in the caller, a load and a wait here, per gate)
        const uint32_t lane = __lane_id();
        const EnvView E{uniform(E_.dev), uniform(E_.lds), uniform(E_.capt0_planes_in_lds), uniform(E_.radii)};
        lds_u32 *list = (lds_u32 *) uniform((lds_cptr) scratch_);
        lds_u32 *mask_lane = list + 2 * kWave + 4 + lane;
        if (VMV_ABLATE_ENV == 2) return bx + by + bz > 1e30f;  // measurement aid: FK only (keeps the FK results alive)
        const ListHdr H = load_list_hdr<V>(E.dev);
        bool own;
        if (E.dev->masked_fine && E.dev->grid[0].cells != nullptr)
            own = env_hit_grid<G, V, !PRE>(E, H, (uint32_t) uniform(grid_class_), bx, by, bz,
                                  Tab::radius(uniform(radius_index_)), active, mask_lane);
        else if (E.dev->masked_fine)
        {
#pragma unroll
            for (int w = 0; w < kMaskWords; ++w) mask_lane[w * kWave] = 0u;
            own = env_hit<G, 1, V, !PRE>(E, H, bx, by, bz, Tab::radius(uniform(radius_index_)), active, mask_lane);
        }
        else
            own = env_hit<G, 0, V, !PRE>(E, H, bx, by, bz, Tab::radius(uniform(radius_index_)), active, nullptr);
        if constexpr (PRE) own = own || (pre && active);
        const bool gate = group_any<G>(own);
        const uint64_t mask = __ballot(gate);
        list[kWave + lane] = 0u;  // flags
        if (gate) list[__popcll(mask & ((1ull << lane) - 1ull))] = lane;
        if (lane == 0) list[2 * kWave] = (uint32_t) __popcll(mask);
        wave_lds_sync();
        return gate;
    }

    // Lists the lanes (whole rakes) that still need an answer, as env_gate does for the lanes whose gate fired.
    template <int G>
    __device__ __forceinline__ bool env_list_active(lds_ptr scratch_, const bool active)
    {
        const uint32_t lane = __lane_id();
        lds_u32 *list = (lds_u32 *) uniform((lds_cptr) scratch_);
        const bool a = group_any<G>(active);
        const uint64_t mask = __ballot(a);
        list[kWave + lane] = 0u;  // flags
        if (a) list[__popcll(mask & ((1ull << lane) - 1ull))] = lane;
        if (lane == 0) list[2 * kWave] = (uint32_t) __popcll(mask);
        wave_lds_sync();
        return a;
    }

    template <int G, typename Tab, int V = kEnvFull>
    __device__ __noinline__ void
    env_fine(const EnvView E_, lds_cptr slab, lds_ptr scratch_, const int n_fine_, const int radii_offset_,
             const int full_ = 0 /* 1: no candidate words (spheres without a bounding-sphere pass: attachments) */,
             const int k_ = -1 /* listed lanes, if the caller knows (else read from the list's header word in LDS) */)
    {
        const uint32_t lane = __lane_id();
        const EnvView E{uniform(E_.dev), uniform(E_.lds), uniform(E_.capt0_planes_in_lds), uniform(E_.radii)};
        lds_u32 *list = (lds_u32 *) uniform((lds_cptr) scratch_);
        lds_u32 *flags = list + kWave;
        const int n_fine = uniform(n_fine_), radii_offset = uniform(radii_offset_);
        const int k_known = uniform(k_);
        const int k = k_known >= 0 ? k_known : (int) uniform(list[2 * kWave]);
        if (k == 0) return;
        const bool masked = E.dev->masked_fine != 0u && uniform(full_) == 0;
        // (the list header is NOT hoisted out of the rounds here: fetched once up front, pinned or not, it takes env_fine's
        // body from 62 to the limit of 96 SGPRs and into SGPR spills through a VGPR it first has to save to scratch, once
        // per call; the gate, which has SGPRs to spare, works from the pinned copy)
        lds_cptr wave_slab = uniform(slab - lane);
        const int items = k * n_fine;
        const float inv_k = 1.0f / (float) k;
        for (int base = 0; base < items; base += kWave)
        {
            const int i = base + (int) lane;
            const bool act = i < items;
            // s = i / k (exact: (i + 0.5) / k is at least 0.5 / 64 away from an integer, far above fp32 error)
            int s = (int) (((float) i + 0.5f) * inv_k);
            s = act ? s : 0;
            const int j = act ? (i - s * k) : 0;
            const uint32_t src = list[j];
            lds_cptr p = wave_slab + 3 * s * kRow + src;  // (row s of the slab: fine sphere s of the staged chunk)
            bool hit;
            if (masked)
                hit = env_hit<G, 2, V>(E, p[0], p[kRow], p[2 * kRow], E.radii[radii_offset + s], act,
                                    list + 2 * kWave + 4 + src);
            else
                hit = env_hit<G, 0, V>(E, p[0], p[kRow], p[2 * kRow], E.radii[radii_offset + s], act, nullptr);
            if (hit) flags[src] = 1u;
        }
        wave_lds_sync();
    }

    // Re-dealing support for the self-collision groups: list the lanes whose predicate holds; returns how many.
    __device__ __forceinline__ int deal_list(lds_u32 *list, const bool pred)
    {
        const uint32_t lane = __lane_id();
        const uint64_t mask = __ballot(pred);
        if (pred) list[__popcll(mask & ((1ull << lane) - 1ull))] = lane;
        wave_lds_sync();
        return __popcll(mask);
    }

    // Appends the lanes whose predicate holds to an LDS list as (lane | tag) entries; returns the new length.
    // The caller issues wave_lds_sync() before reading the list.
    __device__ __forceinline__ int deal_append(lds_u32 *list, const int n, const bool pred, const uint32_t tag)
    {
        const uint64_t mask = __ballot(pred);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
        if (pred) list[n + (int) below] = __lane_id() | tag;
        return n + __popcll(mask);
    }

    // per-wave LDS scratch of the self-collision kernels: lane list [64], hit flags [64], pad [4], A-side candidate
    // words [64], compacted items [8 * 64]
    constexpr int kSelfScratchWords = 3 * kWave + 4 + 8 * kWave;

#undef VMV_HN
#undef VMV_HOFF
#undef VMV_HWB
#undef VMV_HSH
    __device__ __forceinline__ bool env_flag(lds_ptr scratch)
    {
        return ((lds_u32 *) scratch)[kWave + __lane_id()] != 0u;
    }
}  // namespace vmv
