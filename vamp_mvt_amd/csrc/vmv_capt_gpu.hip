// vmv_capt_gpu.hip — Collision-Affording Point Tree build on the GPU (collision/capt.hh:106-369).
//
// Produces exactly the arrays of the host builder (vmv_capt_build.h, which restates the reference's recursive build
// including its quirks), level by level instead of depth first:
//
//   * three presorts give every point its rank along x, y and z (ties by point index, as the host builder orders
//     them); a level's median split is then one radix sort of (node << 24 | rank on the level's axis) — after it, node
//     k of level l owns order[k * S, (k + 1) * S), S = leaves >> l, sorted along the axis;
//   * per node (one thread): split plane = midpoint of the two middle elements, the children's cells, and how much of
//     the sibling half each child is offered (capt.hh:228-257: the hi child scans its lo sibling from the SMALL end
//     and stops at the first point outside r_max, so it gets all of it or nothing; the lo child gets the prefix of the
//     hi half within r_max, found by binary search);
//   * the inherited affordance lists of all nodes of a level live flattened in one array (node-major).  A level maps
//     every entry to "kept by the lo child / by the hi child" (capt.hh:213-226, pruned with r_max only), two prefix
//     sums give the stable positions, and the children's lists are written as [kept inherited..., sibling offer...] —
//     the order the host builder appends them in;
//   * leaves (one wave each): representative point first, then the inherited points within r_max + r_point of the
//     cell, packed 8 per vector and padded with +inf; tight AABB; affordance starts by prefix sum.
// List sizes are only known level by level, so the host reads one counter back per level (nlog2 small syncs).
#include "../../include/vamp_mvt_amd.h"
#include "vmv_capt_build.h"
#include "vmv_common.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace vmv
{
namespace
{
    constexpr int kT = 256;
    inline unsigned nblk(size_t n) { return (unsigned) ((n + kT - 1) / kT); }

#define VMV_C(call)                                              \
    do                                                           \
    {                                                            \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return vmv::hip_status(e_, #call); \
    } while (0)

    struct DevBuf  // growable device buffer
    {
        void *p = nullptr;
        size_t bytes = 0;
        ~DevBuf()
        {
            if (p) (void) hipFree(p);
        }
        hipError_t reserve(size_t need)
        {
            if (need <= bytes) return hipSuccess;
            if (p) (void) hipFree(p);
            p = nullptr;
            bytes = 0;
            const size_t want = need + need / 2 + 4096;
            hipError_t e = hipMalloc(&p, want);
            if (e == hipSuccess) bytes = want;
            return e;
        }
        template <typename T>
        T *as() const
        {
            return static_cast<T *>(p);
        }
    };

    __device__ __forceinline__ uint32_t fkey(float f)  // monotone map float -> uint32
    {
        const uint32_t u = __float_as_uint(f);
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }

    __global__ void pad_points_kernel(const float *__restrict__ in, uint32_t n, uint32_t leaves, float *__restrict__ px,
                                      float *__restrict__ py, float *__restrict__ pz)
    {
        const uint32_t i = blockIdx.x * kT + threadIdx.x;
        if (i >= leaves) return;
        const float inf = INFINITY;
        px[i] = i < n ? in[3 * (size_t) i] : inf;
        py[i] = i < n ? in[3 * (size_t) i + 1] : inf;
        pz[i] = i < n ? in[3 * (size_t) i + 2] : inf;
    }
    __global__ void axis_keys_kernel(const float *__restrict__ coord, uint32_t leaves, unsigned long long *__restrict__ keys,
                                     uint32_t *__restrict__ ids)
    {
        const uint32_t i = blockIdx.x * kT + threadIdx.x;
        if (i >= leaves) return;
        keys[i] = ((unsigned long long) fkey(coord[i]) << 32) | i;
        ids[i] = i;
    }
    __global__ void ranks_kernel(const uint32_t *__restrict__ sorted_ids, uint32_t leaves, uint32_t *__restrict__ rank)
    {
        const uint32_t i = blockIdx.x * kT + threadIdx.x;
        if (i < leaves) rank[sorted_ids[i]] = i;
    }
    __global__ void level_keys_kernel(const uint32_t *__restrict__ order, const uint32_t *__restrict__ rank, uint32_t leaves,
                                      uint32_t shift /* log2 S */, unsigned long long *__restrict__ keys)
    {
        const uint32_t i = blockIdx.x * kT + threadIdx.x;
        if (i < leaves) keys[i] = ((unsigned long long) (i >> shift) << 24) | rank[order[i]];
    }

    struct NodeInfo  // per node of the current level
    {
        float plane;
        uint32_t take_lo;  // hi-half points offered to the lo child: order[mid, take_lo)
        uint32_t take_hi;  // lo-half points offered to the hi child: order[begin, take_hi)
    };

    // capt.hh:106-123 (median split) and :228-257 (sibling offers); cells are [lo xyz, hi xyz] per node
    __global__ void node_kernel(const uint32_t *__restrict__ order, const float *__restrict__ coord, uint32_t n_nodes,
                                uint32_t S, int axis, float r_max, const float *__restrict__ cells_in,
                                float *__restrict__ cells_out, float *__restrict__ tests, uint32_t first_node,
                                NodeInfo *__restrict__ info)
    {
        const uint32_t k = blockIdx.x * kT + threadIdx.x;
        if (k >= n_nodes) return;
        const uint32_t begin = k * S, half = S / 2, mid = begin + half, end = begin + S;
        const float a = coord[order[mid - 1]], b = coord[order[mid]];
        const float plane = (float) ((double) (a + b) / 2.0);
        tests[first_node + k] = plane;
        for (int c = 0; c < 6; ++c)
        {
            const float v = cells_in[6 * (size_t) k + c];
            cells_out[6 * (size_t) (2 * k) + c] = (c == 3 + axis) ? plane : v;      // lo child: hi[axis] = plane
            cells_out[6 * (size_t) (2 * k + 1) + c] = (c == axis) ? plane : v;      // hi child: lo[axis] = plane
        }
        // hi child's offer: scanned from the small end, stops at the first failure -> all of the lo half or nothing
        const float first = coord[order[begin]];
        uint32_t take_hi = begin;
        if (first >= plane - r_max && isfinite(first))
        {
            // sorted ascending: every later element of the lo half is >= first; only +inf padding can end the scan
            uint32_t lo_i = begin, hi_i = mid;  // first index in [begin, mid) whose coordinate is not finite
            while (lo_i < hi_i)
            {
                const uint32_t m = (lo_i + hi_i) / 2;
                if (isfinite(coord[order[m]]))
                    lo_i = m + 1;
                else
                    hi_i = m;
            }
            take_hi = lo_i;
        }
        // lo child's offer: prefix of the hi half with coordinate <= plane + r_max (and finite)
        uint32_t lo_i = mid, hi_i = end;
        const float lim = plane + r_max;
        while (lo_i < hi_i)
        {
            const uint32_t m = (lo_i + hi_i) / 2;
            const float v = coord[order[m]];
            if (v <= lim && isfinite(v))
                lo_i = m + 1;
            else
                hi_i = m;
        }
        info[k] = NodeInfo{plane, lo_i, take_hi};
    }

    // capt.hh:213-226: which children keep an inherited entry (pruned with r_max, not r_max + r_point)
    __global__ void inherit_flags_kernel(const uint32_t *__restrict__ ent_id, const uint32_t *__restrict__ ent_node,
                                         size_t total, const float *__restrict__ coord, const NodeInfo *__restrict__ info,
                                         float r_max, uint32_t *__restrict__ f_lo, uint32_t *__restrict__ f_hi)
    {
        const size_t e = (size_t) blockIdx.x * kT + threadIdx.x;
        if (e > total) return;
        if (e == total)
        {
            f_lo[e] = f_hi[e] = 0u;  // so that the exclusive sums carry the totals at index `total`
            return;
        }
        const float v = coord[ent_id[e]];
        const float plane = info[ent_node[e]].plane;
        f_lo[e] = (v <= plane + r_max) ? 1u : 0u;
        f_hi[e] = (v >= plane - r_max) ? 1u : 0u;
    }

    // sizes of the children's lists: child 2k (lo) and 2k + 1 (hi) of node k; list_off[k] = start of k's parent list
    __global__ void child_sizes_kernel(const NodeInfo *__restrict__ info, const size_t *__restrict__ list_off,
                                       const uint32_t *__restrict__ p_lo, const uint32_t *__restrict__ p_hi, uint32_t n_nodes,
                                       uint32_t S, uint32_t *__restrict__ sizes /* 2 n_nodes + 1 */)
    {
        const uint32_t k = blockIdx.x * kT + threadIdx.x;
        if (k > n_nodes) return;
        if (k == n_nodes)
        {
            sizes[2 * n_nodes] = 0u;
            return;
        }
        const size_t a = list_off[k], b = list_off[k + 1];
        const uint32_t begin = k * S, mid = begin + S / 2;
        sizes[2 * k] = (p_lo[b] - p_lo[a]) + (info[k].take_lo - mid);
        sizes[2 * k + 1] = (p_hi[b] - p_hi[a]) + (info[k].take_hi - begin);
    }
    __global__ void offsets_to_size_t_kernel(const uint32_t *__restrict__ in, uint32_t count, size_t *__restrict__ out)
    {
        const uint32_t i = blockIdx.x * kT + threadIdx.x;
        if (i < count) out[i] = in[i];
    }

    __global__ void fill_inherited_kernel(const uint32_t *__restrict__ ent_id, const uint32_t *__restrict__ ent_node,
                                          size_t total, const size_t *__restrict__ list_off,
                                          const uint32_t *__restrict__ f_lo, const uint32_t *__restrict__ f_hi,
                                          const uint32_t *__restrict__ p_lo, const uint32_t *__restrict__ p_hi,
                                          const size_t *__restrict__ child_off, uint32_t *__restrict__ out_id,
                                          uint32_t *__restrict__ out_node)
    {
        const size_t e = (size_t) blockIdx.x * kT + threadIdx.x;
        if (e >= total) return;
        const uint32_t k = ent_node[e];
        const size_t a = list_off[k];
        if (f_lo[e])
        {
            const size_t o = child_off[2 * k] + (p_lo[e] - p_lo[a]);
            out_id[o] = ent_id[e];
            out_node[o] = 2 * k;
        }
        if (f_hi[e])
        {
            const size_t o = child_off[2 * k + 1] + (p_hi[e] - p_hi[a]);
            out_id[o] = ent_id[e];
            out_node[o] = 2 * k + 1;
        }
    }
    __global__ void fill_sibling_kernel(const uint32_t *__restrict__ order, uint32_t leaves, uint32_t S,
                                        const NodeInfo *__restrict__ info, const size_t *__restrict__ list_off,
                                        const uint32_t *__restrict__ p_lo, const uint32_t *__restrict__ p_hi,
                                        const size_t *__restrict__ child_off, uint32_t *__restrict__ out_id,
                                        uint32_t *__restrict__ out_node)
    {
        const uint32_t i = blockIdx.x * kT + threadIdx.x;
        if (i >= leaves) return;
        const uint32_t k = i / S, begin = k * S, mid = begin + S / 2;
        const size_t a = list_off[k], b = list_off[k + 1];
        if (i >= mid && i < info[k].take_lo)  // hi-half point offered to the lo child, after its inherited entries
        {
            const size_t o = child_off[2 * k] + (p_lo[b] - p_lo[a]) + (i - mid);
            out_id[o] = order[i];
            out_node[o] = 2 * k;
        }
        if (i < info[k].take_hi)  // (i >= begin by construction) lo-half point offered to the hi child
        {
            const size_t o = child_off[2 * k + 1] + (p_hi[b] - p_hi[a]) + (i - begin);
            out_id[o] = order[i];
            out_node[o] = 2 * k + 1;
        }
    }

    // ---- leaves (capt.hh:262-293 in the host builder's form): one wave per leaf ------------------------------------
    struct LeafParams
    {
        float near_sq, reach_sq;
    };
    __device__ __forceinline__ float box_distsq(const float *c, float x, float y, float z)
    {
        const float d0 = x - fminf(fmaxf(x, c[0]), c[3]);  // std::clamp(v, lo, hi)
        const float d1 = y - fminf(fmaxf(y, c[1]), c[4]);
        const float d2 = z - fminf(fmaxf(z, c[2]), c[5]);
        return d0 * d0 + d1 * d1 + d2 * d2;
    }
    // pass 0 counts the accepted entries per leaf (-> vectors per leaf); pass 1 writes them
    template <int PASS>
    __global__ __launch_bounds__(kT) void leaf_kernel(const uint32_t *__restrict__ order, const float *__restrict__ px,
                                                      const float *__restrict__ py, const float *__restrict__ pz,
                                                      uint32_t leaves, const float *__restrict__ cells,
                                                      const size_t *__restrict__ list_off,
                                                      const uint32_t *__restrict__ ent_id, LeafParams P,
                                                      uint32_t *__restrict__ n_vec, const uint32_t *__restrict__ aff_starts,
                                                      float *__restrict__ ax, float *__restrict__ ay, float *__restrict__ az,
                                                      float *__restrict__ aabbs)
    {
        const uint32_t leaf = (blockIdx.x * kT + threadIdx.x) / 64;
        const uint32_t lane = threadIdx.x & 63;
        if (leaf >= leaves) return;
        const uint32_t rep = order[leaf];
        const float rx = px[rep], ry = py[rep], rz = pz[rep];
        const float *c = cells + 6 * (size_t) leaf;
        const bool finite = isfinite(rx);
        // cell inside the ball of radius r_min + r_point around the representative: no afforded points needed
        const float e0 = fmaxf(rx - c[0], c[3] - rx), e1 = fmaxf(ry - c[1], c[4] - ry), e2 = fmaxf(rz - c[2], c[5] - rz);
        const bool covered = (e0 * e0 + e1 * e1 + e2 * e2) <= P.near_sq;
        const size_t a = list_off[leaf], b = list_off[leaf + 1];
        uint32_t accepted = 0;
        float lo[3] = {rx, ry, rz}, hi[3] = {rx, ry, rz};
        const size_t out_base = PASS ? (size_t) aff_starts[leaf] * 8 : 0;
        if (finite && !covered)
        {
            for (size_t base = a; base < b; base += 64)
            {
                const size_t e = base + lane;
                float x = 0.f, y = 0.f, z = 0.f;
                bool ok = false;
                if (e < b)
                {
                    const uint32_t id = ent_id[e];
                    x = px[id], y = py[id], z = pz[id];
                    ok = box_distsq(c, x, y, z) <= P.reach_sq;
                }
                const unsigned long long m = __ballot(ok);
                if (PASS && ok)
                {
                    const uint32_t slot = 1u + accepted + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
                    ax[out_base + slot] = x;
                    ay[out_base + slot] = y;
                    az[out_base + slot] = z;
                    lo[0] = fminf(lo[0], x), lo[1] = fminf(lo[1], y), lo[2] = fminf(lo[2], z);
                    hi[0] = fmaxf(hi[0], x), hi[1] = fmaxf(hi[1], y), hi[2] = fmaxf(hi[2], z);
                }
                accepted += (uint32_t) __popcll(m);
            }
        }
        if (!PASS)
        {
            if (lane == 0) n_vec[leaf] = finite ? (1u + accepted + 7u) / 8u : 0u;
            return;
        }
        // tight box: reduce over the wave
        for (int off = 32; off > 0; off >>= 1)
            for (int k = 0; k < 3; ++k)
            {
                lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
                hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
            }
        if (lane < 3) aabbs[6 * (size_t) leaf + lane] = lane == 0 ? lo[0] : lane == 1 ? lo[1] : lo[2];
        if (lane >= 3 && lane < 6) aabbs[6 * (size_t) leaf + lane] = lane == 3 ? hi[0] : lane == 4 ? hi[1] : hi[2];
        if (finite)
        {
            const uint32_t used = 1u + accepted, padded = ((used + 7u) / 8u) * 8u;
            if (lane == 0)
            {
                ax[out_base] = rx;
                ay[out_base] = ry;
                az[out_base] = rz;
            }
            if (lane < padded - used)  // at most 7 lanes
            {
                ax[out_base + used + lane] = INFINITY;
                ay[out_base + used + lane] = INFINITY;
                az[out_base + used + lane] = INFINITY;
            }
        }
    }

    __global__ void top_box_kernel(const float *__restrict__ px, const float *__restrict__ py, const float *__restrict__ pz,
                                   uint32_t leaves, uint32_t *__restrict__ keys /* 3 min keys, 3 max keys */)
    {
        const uint32_t i = blockIdx.x * kT + threadIdx.x;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (i < leaves && isfinite(px[i]))
        {
            lo[0] = hi[0] = px[i];
            lo[1] = hi[1] = py[i];
            lo[2] = hi[2] = pz[i];
        }
        for (int off = 32; off > 0; off >>= 1)
            for (int k = 0; k < 3; ++k)
            {
                lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
                hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
            }
        if ((threadIdx.x & 63) == 0)
            for (int k = 0; k < 3; ++k)
            {
                atomicMin(&keys[k], fkey(lo[k]));
                atomicMax(&keys[3 + k], fkey(hi[k]));
            }
    }

    inline float unkey(uint32_t k)
    {
        const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    }
}  // namespace

// Builds the CAPT arrays on the current device; they stay there (out.dev).  device_ns: HIP-event time from "points resident on the device"
// to "all arrays built on the device".
int build_capt_device(const float *xyz_host, size_t n, float r_min, float r_max, float r_point, CaptArrays &out,
                      uint64_t *device_ns)
{
    if (n < 2 || n > (1u << 24)) return VMV_ERR_INVALID_ARGUMENT;
    out = CaptArrays{};
    out.r_min = r_min, out.r_max = r_max, out.r_point = r_point;
    const float reach = r_max + r_point;
    const LeafParams LP{(r_min + r_point) * (r_min + r_point), reach * reach};
    while ((size_t{1} << out.nlog2) < n) ++out.nlog2;
    const uint32_t nlog2 = out.nlog2, leaves = 1u << nlog2;
    hipStream_t s = nullptr;

    DevBuf b_in, b_pts, b_rank, b_order, b_keys, b_tmp, b_cells[2], b_tests, b_info, b_listoff[2], b_ent[2], b_flags, b_sizes,
        b_scalars, b_out, b_aabbs, b_nvec;
    VMV_C(b_in.reserve(n * 12));
    VMV_C(hipMemcpy(b_in.p, xyz_host, n * 12, hipMemcpyHostToDevice));
    struct Events  // destroyed on every exit path
    {
        hipEvent_t a = nullptr, b = nullptr;
        ~Events()
        {
            if (a) (void) hipEventDestroy(a);
            if (b) (void) hipEventDestroy(b);
        }
    } ev;
    VMV_C(hipEventCreate(&ev.a));
    VMV_C(hipEventCreate(&ev.b));
    hipEvent_t e0 = ev.a, e1 = ev.b;
    VMV_C(hipEventRecord(e0, s));

    VMV_C(b_pts.reserve((size_t) leaves * 12));
    float *px = b_pts.as<float>(), *py = px + leaves, *pz = py + leaves;
    const float *coord[3] = {px, py, pz};
    hipLaunchKernelGGL(pad_points_kernel, dim3(nblk(leaves)), dim3(kT), 0, s, b_in.as<float>(), (uint32_t) n, leaves, px, py, pz);

    // presort ranks per axis
    VMV_C(b_rank.reserve((size_t) leaves * 12));
    uint32_t *rank[3] = {b_rank.as<uint32_t>(), b_rank.as<uint32_t>() + leaves, b_rank.as<uint32_t>() + 2 * (size_t) leaves};
    VMV_C(b_keys.reserve((size_t) leaves * 8 * 2 + (size_t) leaves * 4 * 2));
    unsigned long long *keys_a = b_keys.as<unsigned long long>(), *keys_b = keys_a + leaves;
    uint32_t *ids_a = reinterpret_cast<uint32_t *>(keys_b + leaves), *ids_b = ids_a + leaves;
    size_t sort_bytes = 0, scan_bytes = 0;
    VMV_C(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys_a, keys_b, ids_a, ids_b, (int) leaves, 0, 64, s));
    // the scans run over lists whose size is only known later; their temporary storage is re-queried when they grow
    VMV_C(b_tmp.reserve(sort_bytes));
    for (int ax = 0; ax < 3; ++ax)
    {
        hipLaunchKernelGGL(axis_keys_kernel, dim3(nblk(leaves)), dim3(kT), 0, s, coord[ax], leaves, keys_a, ids_a);
        VMV_C(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, sort_bytes, keys_a, keys_b, ids_a, ids_b, (int) leaves, 0, 64, s));
        hipLaunchKernelGGL(ranks_kernel, dim3(nblk(leaves)), dim3(kT), 0, s, ids_b, leaves, rank[ax]);
    }
    VMV_C(b_order.reserve((size_t) leaves * 8));
    uint32_t *order = b_order.as<uint32_t>(), *order_b = order + leaves;
    hipLaunchKernelGGL(axis_keys_kernel, dim3(nblk(leaves)), dim3(kT), 0, s, px, leaves, keys_a, order);  // order = iota

    VMV_C(b_tests.reserve((size_t) leaves * 4));
    VMV_C(b_info.reserve((size_t) leaves * sizeof(NodeInfo)));
    VMV_C(b_cells[0].reserve((size_t) leaves * 24));
    VMV_C(b_cells[1].reserve((size_t) leaves * 24));
    VMV_C(b_listoff[0].reserve(((size_t) leaves + 1) * 8));
    VMV_C(b_listoff[1].reserve(((size_t) leaves + 1) * 8));
    VMV_C(b_sizes.reserve(((size_t) leaves + 1) * 4 * 2));
    VMV_C(b_scalars.reserve(256));
    {
        const float inf = std::numeric_limits<float>::infinity();
        const float root[6] = {-inf, -inf, -inf, inf, inf, inf};
        VMV_C(hipMemcpyAsync(b_cells[0].p, root, sizeof(root), hipMemcpyHostToDevice, s));
        VMV_C(hipMemsetAsync(b_listoff[0].p, 0, 16, s));  // root: empty inherited list
    }
    size_t total = 0;  // entries of the current level's flattened lists
    int cur = 0;
    for (uint32_t l = 0; l < nlog2; ++l)
    {
        const uint32_t n_nodes = 1u << l, S = leaves >> l, shift = nlog2 - l;
        const int axis = (int) (l % 3);
        // median split of every node: sort by (node, rank along the axis)
        hipLaunchKernelGGL(level_keys_kernel, dim3(nblk(leaves)), dim3(kT), 0, s, order, rank[axis], leaves, shift, keys_a);
        VMV_C(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, sort_bytes, keys_a, keys_b, order, order_b, (int) leaves, 0,
                                                 (int) (24 + l), s));
        std::swap(order, order_b);
        hipLaunchKernelGGL(node_kernel, dim3(nblk(n_nodes)), dim3(kT), 0, s, order, coord[axis], n_nodes, S, axis, r_max,
                           b_cells[cur].as<float>(), b_cells[cur ^ 1].as<float>(), b_tests.as<float>(), n_nodes - 1,
                           b_info.as<NodeInfo>());
        // inherited lists -> children
        VMV_C(b_flags.reserve((total + 1) * 4 * 4));
        uint32_t *f_lo = b_flags.as<uint32_t>(), *f_hi = f_lo + (total + 1), *p_lo = f_hi + (total + 1), *p_hi = p_lo + (total + 1);
        hipLaunchKernelGGL(inherit_flags_kernel, dim3(nblk(total + 1)), dim3(kT), 0, s, b_ent[cur].as<uint32_t>(),
                           b_ent[cur].as<uint32_t>() + total, total, coord[axis], b_info.as<NodeInfo>(), r_max, f_lo, f_hi);
        size_t need = 0;
        VMV_C(hipcub::DeviceScan::ExclusiveSum(nullptr, need, f_lo, p_lo, (int) (total + 1), s));
        scan_bytes = std::max(scan_bytes, need);
        VMV_C(hipcub::DeviceScan::ExclusiveSum(nullptr, need, (uint32_t *) nullptr, (uint32_t *) nullptr, (int) (2 * n_nodes + 1), s));
        scan_bytes = std::max(scan_bytes, need);
        DevBuf &scan_tmp = b_nvec;  // reused as scan scratch until the leaf stage
        VMV_C(scan_tmp.reserve(scan_bytes));
        size_t sb = scan_tmp.bytes;
        VMV_C(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, sb, f_lo, p_lo, (int) (total + 1), s));
        sb = scan_tmp.bytes;
        VMV_C(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, sb, f_hi, p_hi, (int) (total + 1), s));
        uint32_t *sizes = b_sizes.as<uint32_t>(), *size_scan = sizes + (2 * (size_t) n_nodes + 1);
        hipLaunchKernelGGL(child_sizes_kernel, dim3(nblk(n_nodes + 1)), dim3(kT), 0, s, b_info.as<NodeInfo>(),
                           b_listoff[cur].as<size_t>(), p_lo, p_hi, n_nodes, S, sizes);
        sb = scan_tmp.bytes;
        VMV_C(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, sb, sizes, size_scan, (int) (2 * n_nodes + 1), s));
        hipLaunchKernelGGL(offsets_to_size_t_kernel, dim3(nblk(2 * n_nodes + 1)), dim3(kT), 0, s, size_scan, 2 * n_nodes + 1,
                           b_listoff[cur ^ 1].as<size_t>());
        uint32_t next_total = 0;
        VMV_C(hipMemcpyAsync(&next_total, size_scan + 2 * n_nodes, 4, hipMemcpyDeviceToHost, s));
        VMV_C(hipStreamSynchronize(s));
        // list positions are 32-bit.  A level at most doubles the entries and adds one sibling half per node
        // (<= leaves <= 2^24), so stopping at 2^30 entries keeps every later sum below 2^32.
        if (next_total > (1u << 30)) return VMV_ERR_CAPACITY;
        VMV_C(b_ent[cur ^ 1].reserve(((size_t) next_total + 1) * 8));
        uint32_t *out_id = b_ent[cur ^ 1].as<uint32_t>(), *out_node = out_id + next_total;
        if (total)
            hipLaunchKernelGGL(fill_inherited_kernel, dim3(nblk(total)), dim3(kT), 0, s, b_ent[cur].as<uint32_t>(),
                               b_ent[cur].as<uint32_t>() + total, total, b_listoff[cur].as<size_t>(), f_lo, f_hi, p_lo, p_hi,
                               b_listoff[cur ^ 1].as<size_t>(), out_id, out_node);
        hipLaunchKernelGGL(fill_sibling_kernel, dim3(nblk(leaves)), dim3(kT), 0, s, order, leaves, S, b_info.as<NodeInfo>(),
                           b_listoff[cur].as<size_t>(), p_lo, p_hi, b_listoff[cur ^ 1].as<size_t>(), out_id, out_node);
        VMV_C(hipGetLastError());
        total = next_total;
        cur ^= 1;
    }
    // leaves: count vectors, prefix sum, write
    VMV_C(b_aabbs.reserve((size_t) leaves * 24));
    DevBuf b_nv, b_starts, b_scan2;
    VMV_C(b_nv.reserve(((size_t) leaves + 1) * 4));
    VMV_C(b_starts.reserve(((size_t) leaves + 1) * 4));
    VMV_C(hipMemsetAsync(b_nv.p, 0, ((size_t) leaves + 1) * 4, s));
    hipLaunchKernelGGL(leaf_kernel<0>, dim3(nblk((size_t) leaves * 64)), dim3(kT), 0, s, order, px, py, pz, leaves,
                       b_cells[cur].as<float>(), b_listoff[cur].as<size_t>(), b_ent[cur].as<uint32_t>(), LP, b_nv.as<uint32_t>(),
                       (const uint32_t *) nullptr, (float *) nullptr, (float *) nullptr, (float *) nullptr, (float *) nullptr);
    size_t need = 0;
    VMV_C(hipcub::DeviceScan::ExclusiveSum(nullptr, need, b_nv.as<uint32_t>(), b_starts.as<uint32_t>(), (int) leaves + 1, s));
    VMV_C(b_scan2.reserve(need));
    VMV_C(hipcub::DeviceScan::ExclusiveSum(b_scan2.p, need, b_nv.as<uint32_t>(), b_starts.as<uint32_t>(), (int) leaves + 1, s));
    uint32_t n_vectors = 0;
    VMV_C(hipMemcpyAsync(&n_vectors, b_starts.as<uint32_t>() + leaves, 4, hipMemcpyDeviceToHost, s));
    VMV_C(hipStreamSynchronize(s));
    VMV_C(b_out.reserve((size_t) n_vectors * 8 * 4 * 3 + 64));
    float *ax = b_out.as<float>(), *ay = ax + (size_t) n_vectors * 8, *az = ay + (size_t) n_vectors * 8;
    hipLaunchKernelGGL(leaf_kernel<1>, dim3(nblk((size_t) leaves * 64)), dim3(kT), 0, s, order, px, py, pz, leaves,
                       b_cells[cur].as<float>(), b_listoff[cur].as<size_t>(), b_ent[cur].as<uint32_t>(), LP, (uint32_t *) nullptr,
                       b_starts.as<uint32_t>(), ax, ay, az, b_aabbs.as<float>());
    uint32_t *topk = b_scalars.as<uint32_t>();
    VMV_C(hipMemsetAsync(topk, 0xff, 12, s));
    VMV_C(hipMemsetAsync(topk + 3, 0, 12, s));
    hipLaunchKernelGGL(top_box_kernel, dim3(nblk(leaves)), dim3(kT), 0, s, px, py, pz, leaves, topk);
    VMV_C(hipGetLastError());
    VMV_C(hipEventRecord(e1, s));
    VMV_C(hipEventSynchronize(e1));
    float ms = 0.f;
    VMV_C(hipEventElapsedTime(&ms, e0, e1));
    if (device_ns) *device_ns = (uint64_t) ((double) ms * 1e6);

    // hand the result buffers over (the caller frees them); only the 24-byte top box is read back now
    uint32_t top[6];
    VMV_C(hipMemcpy(top, topk, 24, hipMemcpyDeviceToHost));
    for (int k = 0; k < 6; ++k) out.aabb_top[k] = unkey(top[k]);
    auto release = [](DevBuf &b)
    {
        void *p = b.p;
        b.p = nullptr;
        b.bytes = 0;
        return p;
    };
    out.dev.tests = static_cast<float *>(release(b_tests));
    out.dev.aff_starts = static_cast<uint32_t *>(release(b_starts));
    out.dev.aabbs = static_cast<float *>(release(b_aabbs));
    out.dev.aff = static_cast<float *>(release(b_out));
    out.dev.n_vectors = n_vectors;
    VMV_C(hipGetDevice(&out.dev.device));
    out.host_valid = false;
    return VMV_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// query copy: per leaf, points sorted by their distance to the leaf's k-d cell + per-radius-bucket vector counts
// ---------------------------------------------------------------------------------------------------------------------
namespace
{
    // one thread per affordance vector: its leaf (binary search in aff_starts), the leaf's cell (walk up the implicit
    // tree: node i splits axis depth(i) % 3 at tests[i]; the lo child is 2i + 1), and for each of its 8 points
    // (leaf << 32 | bits of the distance to the cell, rounded towards zero) as the sort key
    __global__ void capt_query_keys_kernel(const float *__restrict__ tests, const uint32_t *__restrict__ starts,
                                           const float *__restrict__ ax, const float *__restrict__ ay, const float *__restrict__ az,
                                           uint32_t nlog2, uint32_t n_vectors, unsigned long long *__restrict__ keys,
                                           uint32_t *__restrict__ slots)
    {
        const uint32_t v = blockIdx.x * kT + threadIdx.x;
        if (v >= n_vectors) return;
        const uint32_t leaves = 1u << nlog2;
        uint32_t lo_i = 0, hi_i = leaves;  // last leaf with starts[leaf] <= v
        while (hi_i - lo_i > 1)
        {
            const uint32_t mid = (lo_i + hi_i) >> 1;
            if (starts[mid] <= v)
                lo_i = mid;
            else
                hi_i = mid;
        }
        const uint32_t leaf = lo_i;
        double lo[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL}, hi[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL};
        uint32_t node = leaf + leaves - 1u;
        for (int depth = (int) nlog2 - 1; depth >= 0; --depth)
        {
            const uint32_t parent = (node - 1u) >> 1;
            const double t = (double) tests[parent];
            const int k = depth % 3;
            if (node == 2u * parent + 1u)
                hi[k] = fmin(hi[k], t);  // lo child: coordinate < plane
            else
                lo[k] = fmax(lo[k], t);  // hi child: coordinate >= plane
            node = parent;
        }
        for (int j = 0; j < 8; ++j)
        {
            const size_t s = (size_t) v * 8 + j;
            const double p[3] = {(double) ax[s], (double) ay[s], (double) az[s]};
            double d2 = 0.0;
            for (int k = 0; k < 3; ++k)
            {
                const double d = fmax(fmax(lo[k] - p[k], p[k] - hi[k]), 0.0);
                d2 += d * d;
            }
            float key = (float) sqrt(d2);
            // padding (+inf) and any other non-finite point sorts LAST (inf - inf above is NaN, which fmax drops: the key
            // would be 0 and the pads would lead the leaf's list); such a point never satisfies the query's `<=`
            if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) key = HUGE_VALF;
            if (!(key >= 0.f)) key = 0.f;                                         // (NaN distance: always tested)
            if (key > 0.f && key < HUGE_VALF && (double) key * (double) key > d2)  // keep it a lower bound
                key = __uint_as_float(__float_as_uint(key) - 1u);
            keys[s] = ((unsigned long long) leaf << 32) | (unsigned long long) __float_as_uint(key);
            slots[s] = (uint32_t) s;
        }
    }

    __global__ void capt_query_gather_kernel(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ slots,
                                             const float *__restrict__ ax, const float *__restrict__ ay, const float *__restrict__ az,
                                             size_t n_slots, float *__restrict__ qx, float *__restrict__ qy, float *__restrict__ qz,
                                             float *__restrict__ vkey)
    {
        const size_t s = (size_t) blockIdx.x * kT + threadIdx.x;
        if (s >= n_slots) return;
        const uint32_t src = slots[s];
        qx[s] = ax[src], qy[s] = ay[src], qz[s] = az[src];
        if ((s & 7u) == 0u) vkey[s >> 3] = __uint_as_float((uint32_t) keys[s]);  // nearest point of the vector
    }

    // one thread per (leaf, bucket): how many leading vectors of the leaf have their nearest point within T_b; written
    // as the uint16 entries of the leaf's record (0xffff = the whole list).  Thread b == 0 also writes the record's head.
    __global__ void capt_query_leaves_kernel(const uint32_t *__restrict__ starts, const float *__restrict__ aabbs,
                                             const float *__restrict__ vkey /* nullptr: no pruning */, uint32_t leaves, float t0,
                                             float step, uint32_t *__restrict__ records)
    {
        const uint32_t i = blockIdx.x * kT + threadIdx.x;
        if (i >= leaves * (uint32_t) kCaptCutBuckets) return;
        const uint32_t leaf = i / (uint32_t) kCaptCutBuckets, b = i % (uint32_t) kCaptCutBuckets;
        const uint32_t s = starts[leaf], e = starts[leaf + 1];
        uint32_t *rec = records + (size_t) leaf * kCaptLeafWords;
        uint32_t n = 0xffffu;
        if (vkey != nullptr && b + 1u < (uint32_t) kCaptCutBuckets)
        {
            const float T = t0 + (float) b * step;
            uint32_t lo = s, hi = e;  // first vector with vkey > T
            while (lo < hi)
            {
                const uint32_t mid = (lo + hi) >> 1;
                if (vkey[mid] <= T)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            n = (lo - s >= 0xffffu) ? 0xffffu : lo - s;
        }
        reinterpret_cast<uint16_t *>(rec + 8)[b] = (uint16_t) n;
        if (b == 0)
        {
            for (int k = 0; k < 6; ++k) rec[k] = __float_as_uint(aabbs[(size_t) leaf * 6 + k]);
            rec[6] = s;
            rec[7] = e - s;
            for (int k = 8 + kCaptCutBuckets / 2; k < kCaptLeafWords; ++k) rec[k] = 0u;
        }
    }

    __global__ void capt_query_planes_kernel(const float *__restrict__ tests, uint32_t nlog2, uint32_t n_tests,
                                             float *__restrict__ planes)
    {
        const uint32_t i = blockIdx.x * kT + threadIdx.x;
        if (i >= n_tests) return;
        const uint32_t l = 31u - (uint32_t) __clz((int) (i + 1u));
        planes[capt_plane_slot(nlog2, l, i)] = tests[i];
    }
    // one thread per grid cell: distance from the cell centre to the nearest cloud point (the representative point of
    // every leaf = the first slot of its first vector; padding leaves hold +inf), minus the half diagonal, rounded down
    __global__ void capt_query_dist_kernel(const uint32_t *__restrict__ starts, const float *__restrict__ ax, const float *__restrict__ ay,
                                           const float *__restrict__ az, uint32_t leaves, uint32_t d0, uint32_t d1, uint32_t d2,
                                           float ox, float oy, float oz, float h, float *__restrict__ out)
    {
        __shared__ float px[kT], py[kT], pz[kT];
        const size_t cell = (size_t) blockIdx.x * kT + threadIdx.x;
        const size_t n_cells = (size_t) d0 * d1 * d2;
        const uint32_t iz = (uint32_t) (cell % d2), iy = (uint32_t) ((cell / d2) % d1), ix = (uint32_t) (cell / ((size_t) d1 * d2));
        const float cx = ox + ((float) ix + 0.5f) * h, cy = oy + ((float) iy + 0.5f) * h, cz = oz + ((float) iz + 0.5f) * h;
        float best = HUGE_VALF;
        for (uint32_t base = 0; base < leaves; base += kT)
        {
            const uint32_t l = base + threadIdx.x;
            float x = HUGE_VALF, y = HUGE_VALF, z = HUGE_VALF;
            if (l < leaves && starts[l + 1] > starts[l])
            {
                const size_t s = (size_t) starts[l] * 8;
                x = ax[s], y = ay[s], z = az[s];
            }
            __syncthreads();
            px[threadIdx.x] = x, py[threadIdx.x] = y, pz[threadIdx.x] = z;
            __syncthreads();
            const uint32_t m = (leaves - base < (uint32_t) kT) ? leaves - base : (uint32_t) kT;
            for (uint32_t k = 0; k < m; ++k)
            {
                const float dx = px[k] - cx, dy = py[k] - cy, dz = pz[k] - cz;
                const float d = dx * dx + dy * dy + dz * dz;
                best = (d < best) ? d : best;  // (inf and NaN never win)
            }
        }
        if (cell >= n_cells) return;
        // lower bound for any centre the device maps to this cell: cells are taken 1 % larger than they are (fp32 cell
        // index at the borders), fp32 rounding of the distance itself is covered by the 1e-5
        const float lb = sqrtf(best) * (1.0f - 1e-6f) - 0.5f * 1.7320508f * h * 1.01f - 1e-5f;
        out[cell] = (best < HUGE_VALF) ? fmaxf(lb, 0.0f) : HUGE_VALF;
    }
}  // namespace

int build_capt_query(const float *d_tests, const uint32_t *d_aff_starts, const float *d_aabbs, const float *d_ax,
                     const float *d_ay, const float *d_az, uint32_t nlog2, uint32_t n_vectors, float r_min, float r_max,
                     float r_point, bool prune, const float aabb_top[6], CaptQueryDev &out)
{
    out = CaptQueryDev{};
    if (nlog2 == 0 || nlog2 > 24 || (size_t) n_vectors * 8 >= (size_t{1} << 31)) return VMV_ERR_CAPACITY;
    if (!(r_max >= r_min) || !std::isfinite(r_max) || !std::isfinite(r_min) || !std::isfinite(r_point)) prune = false;
    const uint32_t leaves = 1u << nlog2, n_tests = leaves - 1u;
    const size_t n_slots = (size_t) n_vectors * 8;
    hipStream_t s = nullptr;
    DevBuf b_keys, b_slots, b_tmp, b_pts, b_vkey, b_leaves, b_planes;
    VMV_C(b_pts.reserve(n_slots * 4 * 3 + 64));
    float *qx = b_pts.as<float>(), *qy = qx + n_slots, *qz = qy + n_slots;
    if (prune && n_slots)
    {
        VMV_C(b_keys.reserve(n_slots * 8 * 2));
        VMV_C(b_slots.reserve(n_slots * 4 * 2));
        unsigned long long *keys_a = b_keys.as<unsigned long long>(), *keys_b = keys_a + n_slots;
        uint32_t *slots_a = b_slots.as<uint32_t>(), *slots_b = slots_a + n_slots;
        hipLaunchKernelGGL(capt_query_keys_kernel, dim3(nblk(n_vectors)), dim3(kT), 0, s, d_tests, d_aff_starts, d_ax, d_ay, d_az,
                           nlog2, n_vectors, keys_a, slots_a);
        size_t sort_bytes = 0;
        const int end_bit = 32 + (int) nlog2;
        VMV_C(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys_a, keys_b, slots_a, slots_b, (int) n_slots, 0, end_bit, s));
        VMV_C(b_tmp.reserve(sort_bytes));
        VMV_C(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, sort_bytes, keys_a, keys_b, slots_a, slots_b, (int) n_slots, 0, end_bit, s));
        VMV_C(b_vkey.reserve((size_t) n_vectors * 4));
        hipLaunchKernelGGL(capt_query_gather_kernel, dim3(nblk(n_slots)), dim3(kT), 0, s, keys_b, slots_b, d_ax, d_ay, d_az, n_slots,
                           qx, qy, qz, b_vkey.as<float>());
    }
    else if (n_slots)
    {
        VMV_C(hipMemcpyAsync(qx, d_ax, n_slots * 4, hipMemcpyDeviceToDevice, s));
        VMV_C(hipMemcpyAsync(qy, d_ay, n_slots * 4, hipMemcpyDeviceToDevice, s));
        VMV_C(hipMemcpyAsync(qz, d_az, n_slots * 4, hipMemcpyDeviceToDevice, s));
    }
    // thresholds: T_b = t0 + b * step for b < B - 1 (T_{B-2} >= r_max + r_point), the last bucket is the whole list
    const float t0 = r_min + r_point;
    const float step = prune ? std::max((r_max - r_min) / (float) (kCaptCutBuckets - 2), 1e-3f) : 1.0f;
    VMV_C(b_leaves.reserve((size_t) leaves * kCaptLeafWords * 4));
    hipLaunchKernelGGL(capt_query_leaves_kernel, dim3(nblk((size_t) leaves * kCaptCutBuckets)), dim3(kT), 0, s, d_aff_starts,
                       d_aabbs, prune && n_slots ? b_vkey.as<float>() : (const float *) nullptr, leaves, t0, step,
                       b_leaves.as<uint32_t>());
    // blocked planes (vmv_device.h, capt_plane_slot)
    const size_t plane_floats = capt_plane_floats(nlog2);
    VMV_C(b_planes.reserve(plane_floats * 4));
    VMV_C(hipMemsetAsync(b_planes.p, 0, plane_floats * 4, s));
    hipLaunchKernelGGL(capt_query_planes_kernel, dim3(nblk(n_tests)), dim3(kT), 0, s, d_tests, nlog2, n_tests, b_planes.as<float>());
    // distance grid over the cloud's box (queries pass the reference's top-box test first, so their centres lie within
    // the box grown by their radius; centres outside the grid are simply not rejected)
    DevBuf b_dist;
    uint32_t dd[3] = {0, 0, 0};
    float dorg[3] = {0, 0, 0}, dh = 0.f;
    if (prune && std::getenv("VMV_CAPT_NO_DIST_GRID") == nullptr)
    {
        const double pad = (double) r_max + (double) r_point + 0.05;
        double ext[3], vol = 1.0;
        bool ok = true;
        for (int k = 0; k < 3; ++k)
        {
            ext[k] = ((double) aabb_top[3 + k] + pad) - ((double) aabb_top[k] - pad);
            ok = ok && std::isfinite(ext[k]) && ext[k] > 0.0 && ext[k] < 1e3;
            vol *= ext[k];
        }
        if (ok)
        {
            // cells: at most ~1M, and at most 4e10 point-cell pairs for the brute-force build
            const double max_cells = std::min(1.0e6, 4.0e10 / (double) leaves);
            dh = (float) std::max(0.03, std::cbrt(vol / max_cells));
            size_t n_cells = 1;
            for (int k = 0; k < 3; ++k)
            {
                dd[k] = (uint32_t) std::ceil(ext[k] / (double) dh);
                dorg[k] = (float) ((double) aabb_top[k] - pad);
                n_cells *= dd[k];
            }
            VMV_C(b_dist.reserve(n_cells * 4));
            hipLaunchKernelGGL(capt_query_dist_kernel, dim3(nblk(n_cells)), dim3(kT), 0, s, d_aff_starts, d_ax, d_ay, d_az, leaves, dd[0],
                               dd[1], dd[2], dorg[0], dorg[1], dorg[2], dh, b_dist.as<float>());
        }
    }
    VMV_C(hipGetLastError());
    VMV_C(hipStreamSynchronize(s));
    auto release = [](DevBuf &b)
    {
        void *p = b.p;
        b.p = nullptr;
        b.bytes = 0;
        return p;
    };
    out.points = static_cast<float *>(release(b_pts));
    out.leaves = static_cast<uint32_t *>(release(b_leaves));
    out.planes = static_cast<float *>(release(b_planes));
    out.t0 = t0;
    out.inv_step = 1.0f / step;
    if (b_dist.p)
    {
        out.dist = static_cast<float *>(release(b_dist));
        for (int k = 0; k < 3; ++k) out.dist_dims[k] = dd[k], out.dist_origin[k] = dorg[k];
        out.dist_inv_cell = 1.0f / dh;
    }
    return VMV_OK;
}

int download_capt(CaptArrays &a)
{
    if (a.host_valid || !a.dev.tests) return VMV_OK;
    const size_t leaves = size_t{1} << a.nlog2, nv = a.dev.n_vectors;
    a.tests.resize(leaves - 1);
    a.aff_starts.resize(leaves + 1);
    a.aabbs.resize(leaves * 6);
    VMV_C(hipMemcpy(a.tests.data(), a.dev.tests, (leaves - 1) * 4, hipMemcpyDeviceToHost));
    VMV_C(hipMemcpy(a.aff_starts.data(), a.dev.aff_starts, (leaves + 1) * 4, hipMemcpyDeviceToHost));
    VMV_C(hipMemcpy(a.aabbs.data(), a.dev.aabbs, leaves * 24, hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k)
    {
        a.aff[k].resize(nv * 8);
        if (nv) VMV_C(hipMemcpy(a.aff[k].data(), a.dev.aff + (size_t) k * nv * 8, nv * 32, hipMemcpyDeviceToHost));
    }
    a.host_valid = true;
    return VMV_OK;
}
}  // namespace vmv
