// vmv_filter.hip — point-cloud filters on the GPU: vamp.filter_pointcloud(..., filter_type)
// (bindings/environment.cc:183-239) with both of the reference's filters,
//   "scdf"       collision/filter.hh:175-275        six space-filling-curve passes: Morton sort, greedy thinning
//   "centervox"  collision/filter_centervox.hh      one point per voxel, the one closest to the voxel centre
// The outputs (which points, in which order) are the reference's; what differs is how they are computed:
//
//  scdf.  Per curve: Morton codes + min/max reduction (one kernel), stable radix sort of (code, index) pairs
//  (rocPRIM through hipCUB; the reference's pdqsort leaves the order of equal codes open, a stable sort fixes it),
//  then the greedy pass "keep a point iff it is farther than min_dist from the last KEPT point".  That pass is a
//  chain i -> nxt(i) = first later point farther than min_dist from i, starting at 0; nxt is computed for every
//  point in parallel and the chain is marked by pointer doubling (after round k every chain node within 2^(k+1)
//  steps of the start is marked), then compacted with a prefix sum.  Restated quirks of the reference: `max` is the MIN of
//  origin + range (filter.hh:193); the index vector is resized to n before culling, so the n - hi entries behind the
//  survivors all refer to point 0 and take part in every later step (:195-216); remap_point converts a float to
//  uint32_t, which the reference's x86-64 build does with cvttss2si to 64 bits and keeps the low half (negative values
//  from those stray copies of point 0 do reach it).
//
//  centervox.  Every point atomically bids (distance² bits << 32 | index) for its voxel in a dense grid_width³ table
//  (lowest distance wins, ties go to the lowest index = the reference's strict `<` on sequential insertion), and
//  records first-appearance indices per x, per (x, y) and per voxel; the occupied voxels are then sorted by
//  (first x, first (x, y), first voxel), which is the order in which the reference's three-level tables were created
//  and therefore the order of its extract_points().
#include "../../include/vamp_mvt_amd.h"
#include "vmv_common.h"

#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

namespace vmv
{
namespace
{
    constexpr int kThreads = 256;
    inline unsigned blocks_for(size_t n) { return (unsigned) ((n + kThreads - 1) / kThreads); }

#define VMV_F(call)                                              \
    do                                                           \
    {                                                            \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return vmv::hip_status(e_, #call); \
    } while (0)

    // one allocation, carved into aligned pieces; freed on scope exit
    struct Arena
    {
        char *base = nullptr;
        size_t used = 0, size = 0;
        ~Arena()
        {
            if (base) (void) hipFree(base);
        }
        template <typename T>
        T *take(size_t count)
        {
            used = (used + 255) & ~size_t{255};
            T *p = reinterpret_cast<T *>(base + used);
            used += count * sizeof(T);
            return p;
        }
    };

    // ---------------------------------------------------------------------------------------------------------
    // scdf
    // ---------------------------------------------------------------------------------------------------------
    __device__ __forceinline__ float dist2(const float *__restrict__ pc, uint32_t a, uint32_t b)
    {
        const float xs = pc[3 * (size_t) a] - pc[3 * (size_t) b], ys = pc[3 * (size_t) a + 1] - pc[3 * (size_t) b + 1],
                    zs = pc[3 * (size_t) a + 2] - pc[3 * (size_t) b + 2];
        return (xs * xs) + (ys * ys) + (zs * zs);  // collision/math.hh sql2_3
    }

    __global__ void cull_flags_kernel(const float *__restrict__ pc, uint32_t n, float sqrange, float ox, float oy, float oz,
                                      float lx, float ly, float lz, float hx, float hy, float hz, int cull,
                                      uint32_t *__restrict__ flags)
    {
        const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
        if (i >= n) return;
        const float x = pc[3 * (size_t) i], y = pc[3 * (size_t) i + 1], z = pc[3 * (size_t) i + 2];
        const float xs = x - ox, ys = y - oy, zs = z - oz;
        const bool keep = !cull || (((xs * xs) + (ys * ys) + (zs * zs)) < sqrange && lx <= x && x <= hx && ly <= y &&
                                    y <= hy && lz <= z && z <= hz);  // filter.hh:202-215
        flags[i] = keep ? 1u : 0u;
    }

    // idx[pos[i]] = i for the kept points; the tail [hi, n) keeps the zeros it was cleared to (filter.hh:195-196)
    __global__ void scatter_kept_kernel(const uint32_t *__restrict__ flags, const uint32_t *__restrict__ pos, uint32_t n,
                                        uint32_t *__restrict__ idx)
    {
        const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
        if (i < n && flags[i]) idx[pos[i]] = i;
    }

    // float -> uint32 as the reference's x86-64 build converts it (cvttss2si to 64 bits, low half; NaN / out of range -> 0)
    __device__ __forceinline__ uint32_t cvt_f32_u32_x86(float v)
    {
        if (!(v > -9223372036854775808.0f && v < 9223372036854775808.0f)) return 0u;
        return (uint32_t) (uint64_t) (long long) v;
    }
    __device__ __forceinline__ uint32_t spread3(uint32_t v)  // bit i -> bit 3 i (the low 11 bits)
    {
        uint32_t out = 0;
#pragma unroll
        for (int b = 0; b < 11; ++b) out |= ((v >> b) & 1u) << (3 * b);
        return out;
    }
    __device__ __forceinline__ uint32_t float_order(float f)  // monotonic map for atomicMin/Max on floats
    {
        const uint32_t u = __float_as_uint(f);
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    __host__ __device__ inline float float_unorder(uint32_t k)
    {
        const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
        float f;
#if defined(__HIP_DEVICE_COMPILE__)
        f = __uint_as_float(u);
#else
        std::memcpy(&f, &u, 4);
#endif
        return f;
    }

    // filter.hh:226-236: codes of the current entries for the curve (c0, c1, c2) and the extremes over all coordinates
    __global__ void morton_kernel(const float *__restrict__ pc, const uint32_t *__restrict__ idx, uint32_t m, float mn,
                                  float mx, int c0, int c1, int c2, uint32_t *__restrict__ code,
                                  uint32_t *__restrict__ extremes /* [0] min key, [1] max key */)
    {
        const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
        float lo = INFINITY, hi = -INFINITY;
        if (i < m)
        {
            const float *p = pc + 3 * (size_t) idx[i];
            const float scale = mx - mn;
            const uint32_t a = cvt_f32_u32_x86(((p[c0] - mn) / scale) * 1000.0f);
            const uint32_t b = cvt_f32_u32_x86(((p[c1] - mn) / scale) * 1000.0f);
            const uint32_t c = cvt_f32_u32_x86(((p[c2] - mn) / scale) * 1000.0f);
            // _pdep_u32 with the masks 0x49249249 / 0x92492492 / 0x24924924 (11, 11 and 10 low bits)
            code[i] = spread3(a & 0x7ffu) | (spread3(b & 0x7ffu) << 1) | (spread3(c & 0x3ffu) << 2);
            lo = fminf(fminf(p[0], p[1]), p[2]);
            hi = fmaxf(fmaxf(p[0], p[1]), p[2]);
        }
        // wave reduce, one atomic per wave
        for (int off = 32; off > 0; off >>= 1)
        {
            lo = fminf(lo, __shfl_xor(lo, off));
            hi = fmaxf(hi, __shfl_xor(hi, off));
        }
        if ((threadIdx.x & 63) == 0)
        {
            atomicMin(&extremes[0], float_order(lo));
            atomicMax(&extremes[1], float_order(hi));
        }
    }

    // sorted copy of the current points, SoA, so that the chain kernel streams contiguous memory
    __global__ void gather_sorted_kernel(const float *__restrict__ pc, const uint32_t *__restrict__ idx, uint32_t m,
                                         float *__restrict__ sx, float *__restrict__ sy, float *__restrict__ sz)
    {
        const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
        if (i >= m) return;
        const float *p = pc + 3 * (size_t) idx[i];
        sx[i] = p[0];
        sy[i] = p[1];
        sz[i] = p[2];
    }

    // filter.hh:243-257, the greedy pass "keep a point iff it is farther than min_dist from the last KEPT point", by
    // ONE wave: the 64 lanes test the next 64 sorted points against the last kept point at once; the first far lane
    // is the next kept point and the lanes behind it are re-tested against it.  Sequential steps = m / 64 + kept
    // instead of m, and the loads (contiguous, independent of the chain) are issued a few windows ahead.
    constexpr int kChainAhead = 4;
    constexpr uint32_t kChainThreshold = 131072;  // larger passes are walked by chain_kernel (measured crossover)
    __device__ __forceinline__ float lane_value(float v, int lane)  // wave-uniform lane: v_readlane, no LDS round trip
    {
        return __uint_as_float((uint32_t) __builtin_amdgcn_readlane((int) __float_as_uint(v), lane));
    }
    __global__ __launch_bounds__(64) void chain_kernel(const float *__restrict__ sx, const float *__restrict__ sy,
                                                       const float *__restrict__ sz, const uint32_t *__restrict__ idx,
                                                       uint32_t m, float sqdist, uint32_t *__restrict__ idx_out,
                                                       uint32_t *__restrict__ count_out)
    {
        const uint32_t lane = threadIdx.x;
        uint32_t count = 0;
        int kept_buf = 0;                    // lane j holds kept entry (count & ~63) + j; stored 64 at a time
        float lx = 0.f, ly = 0.f, lz = 0.f;  // last kept point (wave-uniform)
        auto keep = [&](int id_value)
        {
            kept_buf = (lane == (count & 63u)) ? id_value : kept_buf;
            ++count;
            if ((count & 63u) == 0u) idx_out[count - 64u + lane] = (uint32_t) kept_buf;
        };
        for (uint32_t base = 0; base < m; base += 64 * kChainAhead)
        {
            float x[kChainAhead], y[kChainAhead], z[kChainAhead];
            uint32_t id[kChainAhead];
#pragma unroll
            for (int w = 0; w < kChainAhead; ++w)
            {
                const uint32_t i = base + 64 * w + lane;
                const bool in = i < m;
                x[w] = in ? sx[i] : 0.f;
                y[w] = in ? sy[i] : 0.f;
                z[w] = in ? sz[i] : 0.f;
                id[w] = in ? idx[i] : 0u;
            }
#pragma unroll
            for (int w = 0; w < kChainAhead; ++w)
            {
                const uint32_t i = base + 64 * w + lane;
                bool cand = i < m;
                if (base == 0 && w == 0)
                {
                    // the first sorted point is always kept (filter.hh:245)
                    keep(__builtin_amdgcn_readlane((int) id[0], 0));
                    lx = lane_value(x[0], 0), ly = lane_value(y[0], 0), lz = lane_value(z[0], 0);
                    cand = cand && lane > 0;
                }
                while (true)
                {
                    const float xs = x[w] - lx, ys = y[w] - ly, zs = z[w] - lz;
                    const unsigned long long far = __ballot(cand && ((xs * xs) + (ys * ys) + (zs * zs)) > sqdist);
                    if (far == 0ull) break;
                    const int f = __builtin_amdgcn_readfirstlane(__ffsll((long long) far) - 1);
                    keep(__builtin_amdgcn_readlane((int) id[w], f));
                    lx = lane_value(x[w], f), ly = lane_value(y[w], f), lz = lane_value(z[w], f);
                    cand = cand && (int) lane > f;
                }
            }
        }
        if (lane < (count & 63u)) idx_out[(count & ~63u) + lane] = (uint32_t) kept_buf;
        if (lane == 0) *count_out = count;
    }

    // Small clouds (and every pass after the first, when the points are already min_dist apart) take the parallel
    // route instead: nxt[i] = first j > i farther than min_dist from point i, for every i at once (short scans when
    // the neighbourhoods are sparse), then the chain 0 -> nxt(0) -> ... is marked by pointer doubling (after round k
    // every chain node within 2^(k+1) steps of the start is marked) and compacted with a prefix sum.
    __global__ void next_far_kernel(const float *__restrict__ pc, const uint32_t *__restrict__ idx, uint32_t m, float sqdist,
                                    uint32_t *__restrict__ nxt, uint32_t *__restrict__ mark)
    {
        const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
        if (i > m) return;
        if (i == m)
        {
            nxt[m] = m;  // sentinel
            mark[m] = 0u;
            return;
        }
        const uint32_t a = idx[i];
        uint32_t j = i + 1;
        while (j < m && !(dist2(pc, idx[j], a) > sqdist)) ++j;
        nxt[i] = j;
        mark[i] = (i == 0) ? 1u : 0u;
    }
    __global__ void double_kernel(const uint32_t *__restrict__ jump_in, uint32_t *__restrict__ jump_out, uint32_t m,
                                  uint32_t *__restrict__ mark)
    {
        const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
        if (i > m) return;
        const uint32_t j = jump_in[i];
        if (mark[i] && j < m) mark[j] = 1u;  // benign race: only ever written to 1, and only on chain nodes
        jump_out[i] = jump_in[j];
    }
    __global__ void compact_kernel(const uint32_t *__restrict__ mark, const uint32_t *__restrict__ pos,
                                   const uint32_t *__restrict__ idx_in, uint32_t m, uint32_t *__restrict__ idx_out,
                                   uint32_t *__restrict__ count_out)
    {
        const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
        if (i < m && mark[i]) idx_out[pos[i]] = idx_in[i];
        if (i == m) *count_out = pos[m];  // exclusive sum at m = number of marks (mark[m] is 0)
    }

    __global__ void gather_points_kernel(const float *__restrict__ pc, const uint32_t *__restrict__ idx, uint32_t m,
                                         float *__restrict__ out)
    {
        const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
        if (i >= m) return;
        const float *p = pc + 3 * (size_t) idx[i];
        out[3 * (size_t) i] = p[0];
        out[3 * (size_t) i + 1] = p[1];
        out[3 * (size_t) i + 2] = p[2];
    }

    int filter_scdf(const float *d_pc, uint32_t n, float min_dist, float max_range, const float *origin, const float *lo,
                    const float *hi, int cull, float *d_out, uint32_t *n_out, hipStream_t s)
    {
        const float sqdist = min_dist * min_dist, sqrange = max_range * max_range;
        float mn = std::fmin(std::fmin(origin[0] - max_range, origin[1] - max_range), origin[2] - max_range);
        float mx = std::fmin(std::fmin(origin[0] + max_range, origin[1] + max_range), origin[2] + max_range);  // :193

        size_t sort_bytes = 0, scan_bytes = 0;
        VMV_F(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const uint32_t *) nullptr, (uint32_t *) nullptr,
                                                 (const uint32_t *) nullptr, (uint32_t *) nullptr, (int) n, 0, 32, s));
        VMV_F(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (const uint32_t *) nullptr, (uint32_t *) nullptr,
                                               (int) n + 1, s));
        Arena A;
        A.size = (size_t) (n + 1) * 4 * 9 + sort_bytes + scan_bytes + 16 * 256 + 4096;
        VMV_F(hipMalloc((void **) &A.base, A.size));
        uint32_t *idx_a = A.take<uint32_t>(n + 1), *idx_b = A.take<uint32_t>(n + 1);
        uint32_t *code_a = A.take<uint32_t>(n + 1), *code_b = A.take<uint32_t>(n + 1);
        float *sx = A.take<float>(n + 1), *sy = A.take<float>(n + 1), *sz = A.take<float>(n + 1);
        uint32_t *mark = A.take<uint32_t>(n + 1), *pos = A.take<uint32_t>(n + 1);
        uint32_t *scalars = A.take<uint32_t>(8);
        void *sort_tmp = A.take<char>(sort_bytes), *scan_tmp = A.take<char>(scan_bytes);

        // step 1 (filter.hh:198-216): survivors first, in input order; the rest of the n entries refer to point 0
        VMV_F(hipMemsetAsync(idx_a, 0, (size_t) (n + 1) * 4, s));
        hipLaunchKernelGGL(cull_flags_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, s, d_pc, n, sqrange, origin[0],
                           origin[1], origin[2], lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], cull, mark);
        VMV_F(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, mark, pos, (int) n, s));
        hipLaunchKernelGGL(scatter_kept_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, s, mark, pos, n, idx_a);
        VMV_F(hipGetLastError());

        uint32_t m = n;
        static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
        for (int it = 0; it < 6; ++it)
        {
            VMV_F(hipMemsetAsync(scalars, 0xff, 4, s));  // min key
            VMV_F(hipMemsetAsync(scalars + 1, 0, 4, s));  // max key
            hipLaunchKernelGGL(morton_kernel, dim3(blocks_for(m)), dim3(kThreads), 0, s, d_pc, idx_a, m, mn, mx,
                               perms[it][0], perms[it][1], perms[it][2], code_a, scalars);
            VMV_F(hipcub::DeviceRadixSort::SortPairs(sort_tmp, sort_bytes, code_a, code_b, idx_a, idx_b, (int) m, 0, 32, s));
            if (m > kChainThreshold)
            {
                hipLaunchKernelGGL(gather_sorted_kernel, dim3(blocks_for(m)), dim3(kThreads), 0, s, d_pc, idx_b, m, sx, sy, sz);
                hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, s, sx, sy, sz, idx_b, m, sqdist, idx_a, scalars + 2);
            }
            else
            {
                uint32_t *jin = reinterpret_cast<uint32_t *>(sx), *jout = reinterpret_cast<uint32_t *>(sy);
                hipLaunchKernelGGL(next_far_kernel, dim3(blocks_for((size_t) m + 1)), dim3(kThreads), 0, s, d_pc, idx_b, m,
                                   sqdist, jin, mark);
                for (uint32_t reach = 1; reach < m; reach *= 2)  // after the round with `reach`, 2 * reach steps are covered
                {
                    hipLaunchKernelGGL(double_kernel, dim3(blocks_for((size_t) m + 1)), dim3(kThreads), 0, s, jin, jout, m, mark);
                    std::swap(jin, jout);
                }
                VMV_F(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, mark, pos, (int) m + 1, s));
                hipLaunchKernelGGL(compact_kernel, dim3(blocks_for((size_t) m + 1)), dim3(kThreads), 0, s, mark, pos, idx_b, m,
                                   idx_a, scalars + 2);
            }
            VMV_F(hipGetLastError());
            uint32_t back[3];  // extremes (min key, max key), kept count
            VMV_F(hipMemcpyAsync(back, scalars, 12, hipMemcpyDeviceToHost, s));
            VMV_F(hipStreamSynchronize(s));
            // new_min starts at max, new_max at min (filter.hh:224-225, :232-233)
            const float new_min = std::fmin(mx, float_unorder(back[0])), new_max = std::fmax(mn, float_unorder(back[1]));
            m = back[2];
            mx = (float) ((double) (new_max + mx) / 2.);  // :261-262
            mn = (float) ((double) (new_min + mn) / 2.);
        }
        hipLaunchKernelGGL(gather_points_kernel, dim3(blocks_for(m)), dim3(kThreads), 0, s, d_pc, idx_a, m, d_out);
        VMV_F(hipGetLastError());
        VMV_F(hipStreamSynchronize(s));
        *n_out = m;
        return VMV_OK;
    }

    // ---------------------------------------------------------------------------------------------------------
    // centervox
    // ---------------------------------------------------------------------------------------------------------
    __global__ void voxel_bid_kernel(const float *__restrict__ pc, uint32_t n, float max_range_sq, float ox, float oy,
                                     float oz, float lx, float ly, float lz, float hx, float hy, float hz, float isf,
                                     float voxel_size, uint32_t G, unsigned long long *__restrict__ best,
                                     uint32_t *__restrict__ first_xyz, uint32_t *__restrict__ first_xy,
                                     uint32_t *__restrict__ first_x)
    {
        const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
        if (i >= n) return;
        const float x = pc[3 * (size_t) i], y = pc[3 * (size_t) i + 1], z = pc[3 * (size_t) i + 2];
        const float dx = x - ox, dy = y - oy, dz = z - oz;
        if (dx * dx + dy * dy + dz * dz >= max_range_sq) return;                          // filter_centervox.hh:141-144
        if (x < lx || x > hx || y < ly || y > hy || z < lz || z > hz) return;             // :146-150
        auto coord = [&](float p, float mn) -> uint32_t
        {
            int c = (int) ((p - mn) * isf);  // :153-161
            c = c < 0 ? 0 : (c > 254 ? 254 : c);
            return (uint32_t) c;
        };
        const uint32_t vx = coord(x, lx), vy = coord(y, ly), vz = coord(z, lz);
        const float cx = lx + ((float) vx + 0.5f) * voxel_size, cy = ly + ((float) vy + 0.5f) * voxel_size,
                    cz = lz + ((float) vz + 0.5f) * voxel_size;  // :21-25
        const float ex = x - cx, ey = y - cy, ez = z - cz;
        const float dsq = ex * ex + ey * ey + ez * ez;  // :28-31
        const size_t cell = ((size_t) vx * G + vy) * G + vz;
        atomicMin(&best[cell], ((unsigned long long) __float_as_uint(dsq) << 32) | i);
        atomicMin(&first_xyz[cell], i);
        atomicMin(&first_xy[vx * G + vy], i);
        atomicMin(&first_x[vx], i);
    }

    __global__ void voxel_collect_kernel(const unsigned long long *__restrict__ best, const uint32_t *__restrict__ first_xyz,
                                         const uint32_t *__restrict__ first_xy, const uint32_t *__restrict__ first_x,
                                         uint32_t G, size_t cells, uint32_t capacity, uint32_t *__restrict__ counter,
                                         uint32_t *__restrict__ k0, uint32_t *__restrict__ k1, uint32_t *__restrict__ k2,
                                         uint32_t *__restrict__ winner)
    {
        const size_t cell = (size_t) blockIdx.x * kThreads + threadIdx.x;
        if (cell >= cells) return;
        const uint32_t f = first_xyz[cell];
        if (f == 0xffffffffu) return;
        const uint32_t slot = atomicAdd(counter, 1u);
        if (slot >= capacity) return;  // the host reports the overflow from the counter
        const uint32_t vx = (uint32_t) (cell / ((size_t) G * G)), vy = (uint32_t) ((cell / G) % G);
        k0[slot] = first_x[vx];
        k1[slot] = first_xy[vx * G + vy];
        k2[slot] = f;
        winner[slot] = (uint32_t) (best[cell] & 0xffffffffull);
    }

    __global__ void gather_u32_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ perm, uint32_t m,
                                      uint32_t *__restrict__ dst)
    {
        const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
        if (i < m) dst[i] = src[perm[i]];
    }
    __global__ void iota_kernel(uint32_t *__restrict__ v, uint32_t m)
    {
        const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
        if (i < m) v[i] = i;
    }

    int filter_centervox(const float *d_pc, uint32_t n, float voxel_size, float max_range, const float *origin,
                         const float *lo, const float *hi, float *d_out, uint32_t *n_out, hipStream_t s)
    {
        // filter_centervox.hh:94-127
        const float width = std::fmax(std::fmax(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
        int grid_width = (int) std::ceil(width / voxel_size);
        if (grid_width > 255) grid_width = 255;
        if (!(grid_width >= 1)) return VMV_ERR_INVALID_ARGUMENT;
        const float isf = (float) grid_width / width;
        size_t pool = (size_t) (std::pow(width / voxel_size, 3.0f) * 0.05f);
        if (pool > 32768) pool = 32768;
        // coordinates are clamped to 254 whatever the grid width (:153-161), and (p - min) * isf can reach grid_width
        const uint32_t G = (uint32_t) std::min(255, grid_width + 1);
        const size_t cells = (size_t) G * G * G;
        const uint32_t cap = (uint32_t) pool + 1;  // one more than the pool, to see the overflow

        size_t sort_bytes = 0;
        VMV_F(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const uint32_t *) nullptr, (uint32_t *) nullptr,
                                                 (const uint32_t *) nullptr, (uint32_t *) nullptr, (int) cap, 0, 32, s));
        Arena A;
        A.size = cells * 12 + (size_t) G * G * 4 + 256 * 4 + (size_t) cap * 4 * 8 + sort_bytes + 16 * 256 + 4096;
        VMV_F(hipMalloc((void **) &A.base, A.size));
        unsigned long long *best = A.take<unsigned long long>(cells);
        uint32_t *first_xyz = A.take<uint32_t>(cells), *first_xy = A.take<uint32_t>((size_t) G * G);
        uint32_t *first_x = A.take<uint32_t>(256), *counter = A.take<uint32_t>(4);
        uint32_t *k0 = A.take<uint32_t>(cap), *k1 = A.take<uint32_t>(cap), *k2 = A.take<uint32_t>(cap);
        uint32_t *winner = A.take<uint32_t>(cap), *perm_a = A.take<uint32_t>(cap), *perm_b = A.take<uint32_t>(cap);
        uint32_t *key_a = A.take<uint32_t>(cap), *key_b = A.take<uint32_t>(cap);
        void *sort_tmp = A.take<char>(sort_bytes);
        // everything the bids touch starts at all-ones (one memset: the pieces are contiguous up to `counter`)
        VMV_F(hipMemsetAsync(best, 0xff, (size_t) ((char *) counter - (char *) best), s));
        VMV_F(hipMemsetAsync(counter, 0, 16, s));

        hipLaunchKernelGGL(voxel_bid_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, s, d_pc, n, max_range * max_range,
                           origin[0], origin[1], origin[2], lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], isf, voxel_size, G,
                           best, first_xyz, first_xy, first_x);
        hipLaunchKernelGGL(voxel_collect_kernel, dim3(blocks_for(cells)), dim3(kThreads), 0, s, best, first_xyz, first_xy,
                           first_x, G, cells, cap, counter, k0, k1, k2, winner);
        VMV_F(hipGetLastError());
        uint32_t count = 0;
        VMV_F(hipMemcpyAsync(&count, counter, 4, hipMemcpyDeviceToHost, s));
        VMV_F(hipStreamSynchronize(s));
        if (count > pool) return VMV_ERR_CAPACITY;  // the reference throws "Voxel pool exhausted" (:132-134)
        *n_out = count;
        if (count == 0) return VMV_OK;

        // creation order of the reference's tables = lexicographic (first x, first (x, y), first voxel): LSD passes
        hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(count)), dim3(kThreads), 0, s, perm_a, count);
        const uint32_t *keys[3] = {k2, k1, k0};
        for (int pass = 0; pass < 3; ++pass)
        {
            hipLaunchKernelGGL(gather_u32_kernel, dim3(blocks_for(count)), dim3(kThreads), 0, s, keys[pass], perm_a, count,
                               key_a);
            VMV_F(hipcub::DeviceRadixSort::SortPairs(sort_tmp, sort_bytes, key_a, key_b, perm_a, perm_b, (int) count, 0, 32, s));
            std::swap(perm_a, perm_b);
        }
        hipLaunchKernelGGL(gather_u32_kernel, dim3(blocks_for(count)), dim3(kThreads), 0, s, winner, perm_a, count, key_a);
        hipLaunchKernelGGL(gather_points_kernel, dim3(blocks_for(count)), dim3(kThreads), 0, s, d_pc, key_a, count, d_out);
        VMV_F(hipGetLastError());
        VMV_F(hipStreamSynchronize(s));
        return VMV_OK;
    }
}  // namespace
}  // namespace vmv

extern "C" int vmv_filter_pointcloud(const float *points, size_t n, float min_dist, float max_range, float voxel_size,
                                     const float *origin, const float *ws_min, const float *ws_max, int cull,
                                     int filter_type, float *out, size_t capacity, size_t *n_out, uint64_t *nanoseconds,
                                     uint64_t *device_nanoseconds)
{
    if (!n_out || !origin || !ws_min || !ws_max || (n && !points) || (filter_type != 0 && filter_type != 1) ||
        n >= (size_t{1} << 31))
        return VMV_ERR_INVALID_ARGUMENT;
    *n_out = 0;
    if (n == 0) return VMV_OK;  // filter.hh:185-188
    int devs = 0;
    if (hipGetDeviceCount(&devs) != hipSuccess || devs <= 0)
        return vmv::hip_status(hipErrorNoDevice, "vmv_filter_pointcloud (this library has no CPU path)");
    const auto t0 = std::chrono::steady_clock::now();
    float *d_pc = nullptr, *d_out = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = VMV_OK;
    uint32_t m = 0;
    auto cleanup = [&]()
    {
        if (d_pc) (void) hipFree(d_pc);
        if (d_out) (void) hipFree(d_out);
        if (e0) (void) hipEventDestroy(e0);
        if (e1) (void) hipEventDestroy(e1);
    };
#define VMV_FC(call)                                \
    do                                              \
    {                                               \
        hipError_t e_ = (call);                     \
        if (e_ != hipSuccess)                       \
        {                                           \
            cleanup();                              \
            return vmv::hip_status(e_, #call);      \
        }                                           \
    } while (0)
    VMV_FC(hipMalloc((void **) &d_pc, n * 12));
    VMV_FC(hipMalloc((void **) &d_out, n * 12));
    VMV_FC(hipMemcpy(d_pc, points, n * 12, hipMemcpyHostToDevice));
    VMV_FC(hipEventCreate(&e0));
    VMV_FC(hipEventCreate(&e1));
    VMV_FC(hipEventRecord(e0, s));
    rc = (filter_type == 0) ?
             vmv::filter_scdf(d_pc, (uint32_t) n, min_dist, max_range, origin, ws_min, ws_max, cull, d_out, &m, s) :
             vmv::filter_centervox(d_pc, (uint32_t) n, voxel_size, max_range, origin, ws_min, ws_max, d_out, &m, s);
    if (rc != VMV_OK)
    {
        cleanup();
        return rc;
    }
    VMV_FC(hipEventRecord(e1, s));
    VMV_FC(hipEventSynchronize(e1));
    float ms = 0.f;
    VMV_FC(hipEventElapsedTime(&ms, e0, e1));
    *n_out = m;
    if (out)
    {
        if (m > capacity)
        {
            cleanup();
            return VMV_ERR_CAPACITY;
        }
        VMV_FC(hipMemcpy(out, d_out, (size_t) m * 12, hipMemcpyDeviceToHost));
    }
    cleanup();
    if (device_nanoseconds) *device_nanoseconds = (uint64_t) ((double) ms * 1e6);
    if (nanoseconds)
        *nanoseconds =
            (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return VMV_OK;
}
