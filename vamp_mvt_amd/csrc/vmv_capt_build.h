// vmv_capt_build.h — host-side builder of the Collision-Affording Point Tree arrays the query kernel reads.
//
// The arrays are the *input* of the device query, so the build has to reproduce the reference structure
// (collision/capt.hh:106-369) including its quirks:
//   * points are padded to a power of two with +inf;
//   * splits cycle x, y, z; the split value is the midpoint of the two middle elements;
//   * inherited affordances are pruned with r_max (not r_max + r_point)        — capt.hh:217,222;
//   * the hi child scans its lo sibling from the *small* end and stops at the first point farther than
//     r_max from the plane, so most cross-boundary affordances on that side are missing — capt.hh:228-235;
//   * leaves keep the representative point first, then afforded points, 8 per vector, +inf padded.
// Equal coordinates: the reference sorts with pdqsort_branchless, which leaves the order of equal keys
// unspecified; this builder orders ties by point index.
//
// Written iteratively (explicit work stack) — the product's own implementation, independent of oracle/.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <limits>
#include <numeric>
#include <vector>

namespace vmv
{
    struct CaptArrays
    {
        uint32_t nlog2 = 0;
        std::vector<float> tests;          // 2^nlog2 - 1
        std::vector<uint32_t> aff_starts;  // 2^nlog2 + 1
        std::vector<float> aabbs;          // 2^nlog2 * 6 (lower xyz, upper xyz)
        std::array<std::vector<float>, 3> aff;  // n_vectors * 8 each
        float aabb_top[6];
        float r_min, r_max, r_point;
        // device-resident copy (GPU build): the environment owns these allocations; the host vectors above are then
        // filled on demand (inspection) and `resident` says which device holds the arrays
        struct Device
        {
            float *tests = nullptr;
            uint32_t *aff_starts = nullptr;
            float *aabbs = nullptr;
            float *aff = nullptr;  // x | y | z, n_vectors * 8 floats each
            uint32_t n_vectors = 0;
            int device = -1;
        } dev;
        bool host_valid = true;
        uint32_t n_aff_vectors() const { return dev.tests ? dev.n_vectors : static_cast<uint32_t>(aff[0].size() / 8); }
        uint32_t n_tests() const { return (1u << nlog2) - 1u; }
    };

    namespace capt_detail
    {
        struct Box
        {
            float lo[3], hi[3];
            void grow(const float *p)
            {
                for (int k = 0; k < 3; ++k)
                {
                    lo[k] = std::min(lo[k], p[k]);
                    hi[k] = std::max(hi[k], p[k]);
                }
            }
            // squared distance from p to the box (capt.hh:48-55)
            float distsq(const float *p) const
            {
                const float d0 = p[0] - std::clamp(p[0], lo[0], hi[0]);
                const float d1 = p[1] - std::clamp(p[1], lo[1], hi[1]);
                const float d2 = p[2] - std::clamp(p[2], lo[2], hi[2]);
                return d0 * d0 + d1 * d1 + d2 * d2;
            }
            // is the whole cell inside the ball of squared radius rsq around p (capt.hh:39-46)
            bool inside_ball(const float *p, float rsq) const
            {
                const float d0 = std::max(p[0] - lo[0], hi[0] - p[0]);
                const float d1 = std::max(p[1] - lo[1], hi[1] - p[1]);
                const float d2 = std::max(p[2] - lo[2], hi[2] - p[2]);
                return (d0 * d0 + d1 * d1 + d2 * d2) <= rsq;
            }
        };

        struct Work
        {
            uint32_t begin, count, node;
            std::vector<uint32_t> afford;
            Box cell;
            uint8_t axis;
        };
    }  // namespace capt_detail

    inline bool build_capt(const float *xyz, size_t n, float r_min, float r_max, float r_point, CaptArrays &out)
    {
        using namespace capt_detail;
        if (n < 2 || n > (1u << 24)) return false;
        constexpr float inf = std::numeric_limits<float>::infinity();
        out = CaptArrays{};
        out.r_min = r_min;
        out.r_max = r_max;
        out.r_point = r_point;
        const float reach = r_max + r_point;
        const float reach_sq = reach * reach;
        const float near_sq = (r_min + r_point) * (r_min + r_point);

        while ((size_t{1} << out.nlog2) < n) ++out.nlog2;
        const size_t leaves = size_t{1} << out.nlog2;
        std::vector<float> pts(leaves * 3, inf);
        std::copy(xyz, xyz + 3 * n, pts.begin());
        auto P = [&](uint32_t id) { return &pts[3 * size_t{id}]; };

        out.tests.assign(leaves - 1, std::numeric_limits<float>::quiet_NaN());
        out.aff_starts.assign(1, 0u);
        out.aff_starts.reserve(leaves + 1);
        out.aabbs.reserve(leaves * 6);
        for (int k = 0; k < 3; ++k)
        {
            out.aabb_top[k] = inf;
            out.aabb_top[3 + k] = -inf;
        }

        std::vector<uint32_t> order(leaves);
        std::iota(order.begin(), order.end(), 0u);

        // depth-first, lo child before hi child, so leaves come out in index order
        std::vector<Work> stack;
        stack.push_back(Work{0u, static_cast<uint32_t>(leaves), 0u, {}, Box{{-inf, -inf, -inf}, {inf, inf, inf}}, 0});
        while (!stack.empty())
        {
            Work w = std::move(stack.back());
            stack.pop_back();

            if (w.count == 1)
            {
                const float *rep = P(order[w.begin]);
                Box tight{{rep[0], rep[1], rep[2]}, {rep[0], rep[1], rep[2]}};
                if (std::isfinite(rep[0]))
                {
                    Box top{{out.aabb_top[0], out.aabb_top[1], out.aabb_top[2]},
                            {out.aabb_top[3], out.aabb_top[4], out.aabb_top[5]}};
                    top.grow(rep);
                    std::copy(top.lo, top.lo + 3, out.aabb_top);
                    std::copy(top.hi, top.hi + 3, out.aabb_top + 3);

                    float lane[3][8] = {{rep[0]}, {rep[1]}, {rep[2]}};
                    int fill = 1;
                    auto flush = [&]()
                    {
                        for (int k = 0; k < 3; ++k) out.aff[k].insert(out.aff[k].end(), lane[k], lane[k] + 8);
                    };
                    if (!w.cell.inside_ball(rep, near_sq))
                    {
                        for (const uint32_t id : w.afford)
                        {
                            const float *p = P(id);
                            if (w.cell.distsq(p) <= reach_sq)
                            {
                                tight.grow(p);
                                for (int k = 0; k < 3; ++k) lane[k][fill] = p[k];
                                if (++fill == 8)
                                {
                                    flush();
                                    fill = 0;
                                }
                            }
                        }
                    }
                    if (fill > 0)
                    {
                        for (int j = fill; j < 8; ++j)
                            for (int k = 0; k < 3; ++k) lane[k][j] = inf;
                        flush();
                    }
                }
                out.aabbs.insert(out.aabbs.end(), tight.lo, tight.lo + 3);
                out.aabbs.insert(out.aabbs.end(), tight.hi, tight.hi + 3);
                out.aff_starts.push_back(out.n_aff_vectors());
                continue;
            }

            // median split on w.axis (capt.hh:106-123)
            const int ax = w.axis;
            auto first = order.begin() + w.begin, last = first + w.count;
            std::sort(first, last,
                      [&](uint32_t a, uint32_t b)
                      {
                          const float fa = P(a)[ax], fb = P(b)[ax];
                          return fa < fb || (!(fb < fa) && a < b);
                      });
            const uint32_t half = w.count / 2;
            const uint32_t mid = w.begin + half;
            const float plane =
                static_cast<float>(static_cast<double>(P(order[mid - 1])[ax] + P(order[mid])[ax]) / 2.0);
            out.tests[w.node] = plane;

            Work lo{w.begin, half, 2 * w.node + 1, {}, w.cell, static_cast<uint8_t>((ax + 1) % 3)};
            Work hi{mid, half, 2 * w.node + 2, {}, w.cell, static_cast<uint8_t>((ax + 1) % 3)};
            lo.cell.hi[ax] = plane;
            hi.cell.lo[ax] = plane;

            // inherited affordances: keep on each side what is within r_max of the plane (capt.hh:213-226)
            for (const uint32_t id : w.afford)
            {
                const float v = P(id)[ax];
                if (v <= plane + r_max) lo.afford.push_back(id);
                if (v >= plane - r_max) hi.afford.push_back(id);
            }
            // new affordances from the sibling half (capt.hh:228-257)
            uint32_t take_hi = w.begin;  // lo-half points offered to the hi child, scanned from the small end
            while (take_hi < mid && P(order[take_hi])[ax] >= plane - r_max && std::isfinite(P(order[take_hi])[ax]))
                ++take_hi;
            uint32_t take_lo = mid;  // hi-half points offered to the lo child, scanned from the plane outwards
            while (take_lo < w.begin + w.count && P(order[take_lo])[ax] <= plane + r_max &&
                   std::isfinite(P(order[take_lo])[ax]))
                ++take_lo;
            hi.afford.insert(hi.afford.end(), order.begin() + w.begin, order.begin() + take_hi);
            lo.afford.insert(lo.afford.end(), order.begin() + mid, order.begin() + take_lo);

            stack.push_back(std::move(hi));  // popped second
            stack.push_back(std::move(lo));  // popped first
        }
        return true;
    }

    // The same arrays built on the current HIP device (vmv_capt_gpu.hip).  They stay on the device (out.dev, owned by
    // the caller from then on); download_capt fills the host vectors when somebody wants to look at them.
    int build_capt_device(const float *xyz_host, size_t n, float r_min, float r_max, float r_point, CaptArrays &out,
                          uint64_t *device_ns);
    int download_capt(CaptArrays &a);

    // Query copy of a cloud (vmv_capt_gpu.hip, built at vmv_env_finalize from the arrays above, which stay what inspection
    // returns).  Not part of the reference's structure — the same planes, boxes and points, laid out for the device walk:
    //  * points: a query centre that descends to leaf L lies in L's k-d cell, so |p - c| >= dist(p, cell_L) for every point
    //    p of L's list.  The copy holds each leaf's points sorted by that distance (same vector range, +inf padding
    //    last) and per (leaf, radius bucket b) how many leading vectors hold a point with dist(p, cell_L) <= T_b,
    //    T_b = t0 + b * step (the last bucket = the whole list).  A query of radius r takes the first bucket with
    //    T_b >= r + r_point + 1e-4 m and tests those vectors with the reference's predicate; every point it skips is
    //    farther than r + r_point by more than 1e-4 m, five orders of magnitude above fp32 rounding at metre scale
    //    (`prune` = false — VMV_CAPT_NO_PREFIX=1, coordinates beyond +-1e2 m, non-finite radii — keeps the order and
    //    makes every bucket the whole list);
    //  * leaves: one 128-byte record per leaf = [box lo xyz, hi xyz | first vector | vector count | 32 x uint16 bucket
    //    counts (0xffff = whole list) | pad]: one cache line answers the leaf test and says what to walk;
    //  * planes: the split planes in blocks of 3 tree levels (7 planes + pad = 32 bytes, local heap order; groups of
    //    levels bottom-aligned, the blocks of a group stored left to right): a descent step fetches one block and
    //    resolves three levels from registers — five dependent fetches for a 16,384-leaf tree instead of fourteen.
    // (kCaptCutBuckets, kCaptCutMargin, kCaptLeafWords, capt_plane_slot(): vmv_device.h)
    struct CaptQueryDev
    {
        float *points = nullptr;     // x | y | z, n_vectors * 8 floats each
        uint32_t *leaves = nullptr;  // 2^nlog2 * kCaptLeafWords
        float *planes = nullptr;     // blocked copy of `tests` (vmv_device.h: capt_plane_slot)
        float t0 = 0.f, inv_step = 0.f;
        // distance grid (optional, nullptr = none): for every cell of a uniform grid over the cloud's box a LOWER bound of
        // the distance from any point of the cell to the nearest cloud point (distance at the cell centre - half a cell
        // diagonal, rounded down).  A query whose cell's bound exceeds r + r_point + 1e-4 m cannot touch any point of
        // the cloud, whatever leaf it descends to: it is rejected before the descent (pruning only).
        float *dist = nullptr;
        uint32_t dist_dims[3] = {0, 0, 0};
        float dist_origin[3] = {0, 0, 0}, dist_inv_cell = 0.f;
    };
    int build_capt_query(const float *d_tests, const uint32_t *d_aff_starts, const float *d_aabbs, const float *d_ax,
                         const float *d_ay, const float *d_az, uint32_t nlog2, uint32_t n_vectors, float r_min, float r_max,
                         float r_point, bool prune, const float aabb_top[6], CaptQueryDev &out);
}  // namespace vmv
