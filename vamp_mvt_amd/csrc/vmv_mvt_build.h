// vmv_mvt_build.h — host-side builder of the Multi-level Voxel Table as the query kernel reads it.
//
// Reference: collision/mvt.hh (the fork's own point-cloud structure).  Semantics that define the query's answer
// and are reproduced here:
//   * uniform grid over the workspace box: grid_width = floor(workspace_width_x / r_max) (x extent only!),
//     inverse_scale_factor = grid_width / workspace_width_x, voxel of a point = trunc(clamp((p - ws_min) * isf,
//     0, grid_width - 1)) per axis (mvt.hh:438-447, 536-549);
//   * every voxel keeps its points and their bounding box; the global box is the union (mvt.hh:86-98, 595-609);
//   * the reference sizes fixed pools up front and THROWS inside a noexcept constructor when one runs out
//     (SURVEY.md §5): the same conditions are detected here and reported as a status instead:
//       - a voxel holding more than next_pow2(4 * trunc((r_max / 0.02)^3)) / 4 points (min 8)   (mvt.hh:455-468, 66-70)
//       - more than 10 % of the grid's voxels occupied (point pool)                                (mvt.hh:463-467, 644-648)
//       - more than half of the grid's (x, y) columns occupied (z-table pool)                      (mvt.hh:494-508, 634-637)
//
// MI355X layout: the reference's three dependent pointer tables (x -> y -> z -> voxel) exist to save host memory;
// with 288 GB of HBM the device image is ONE dense grid_width^3 table of voxel indices (a single load per cell),
// voxel records, and compact SoA point arrays addressed by (offset, count).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace vmv
{
    struct MvtArrays
    {
        uint32_t grid_width = 0, capacity = 0;
        float inv_scale = 0.f, r_min = 0.f, r_max = 0.f, r_point = 0.f;
        float ws_min[3], ws_max[3], gmin[3], gmax[3];
        std::vector<uint32_t> cells;        // grid_width^3, index (x * gw + y) * gw + z -> voxel or 0xffffffff
        std::vector<float> vox_bbox;        // [n_vox][6]
        std::vector<uint32_t> vox_offset;   // [n_vox + 1] into the point arrays
        std::vector<float> px, py, pz;      // compact, voxel-major, insertion order inside a voxel
        uint32_t n_voxels() const { return (uint32_t) (vox_offset.empty() ? 0 : vox_offset.size() - 1); }
    };

    enum class MvtStatus
    {
        ok = 0,
        voxel_capacity = 1,
        point_pool = 2,
        ztable_pool = 3,
        degenerate = 4,
        too_large = 5
    };

    inline MvtStatus build_mvt(const float *xyz, size_t n, float r_min, float r_max, const float *ws_min,
                               const float *ws_max, float r_point, MvtArrays &out)
    {
        constexpr uint32_t kNone = 0xffffffffu;
        out = MvtArrays{};
        out.r_min = r_min;
        out.r_max = r_max;
        out.r_point = r_point;
        std::copy(ws_min, ws_min + 3, out.ws_min);
        std::copy(ws_max, ws_max + 3, out.ws_max);
        const float width = ws_max[0] - ws_min[0];
        const float cells_f = std::floor(width / r_max);
        if (!(cells_f >= 1.0f) || n == 0) return MvtStatus::degenerate;
        const uint32_t gw = (uint32_t) std::min<double>(cells_f, 65535.0);
        if ((uint64_t) gw * gw * gw > (1ull << 28)) return MvtStatus::too_large;  // dense table limit (1 GiB)
        out.grid_width = gw;
        out.inv_scale = (float) gw / width;

        // the reference's pool arithmetic (mvt.hh:455-508)
        const size_t est = (size_t) std::pow((double) r_max / 0.02, (double) 3.0f);
        unsigned bytes = (unsigned) est * 4u;
        {  // next power of two
            unsigned v = bytes ? bytes - 1 : 0;
            v |= v >> 1, v |= v >> 2, v |= v >> 4, v |= v >> 8, v |= v >> 16;
            bytes = bytes ? v + 1 : 1;
        }
        bytes = std::max(bytes, 32u);
        out.capacity = bytes / 4u;
        const size_t pool_floats = (size_t) ((double) ((size_t) gw * gw * gw) * 0.1 * (double) (size_t) bytes * 3) / 4;
        const size_t max_voxels = pool_floats / (3 * (size_t) out.capacity);
        const size_t max_columns = (size_t) ((double) ((size_t) gw * gw) * 0.5);

        // pass 1: voxel of every point, voxel ids in order of first appearance, per-voxel counts
        out.cells.assign((size_t) gw * gw * gw, kNone);
        std::vector<uint8_t> column_used((size_t) gw * gw, 0);
        std::vector<uint32_t> vox_of(n), counts;
        size_t columns = 0;
        for (size_t i = 0; i < n; ++i)
        {
            uint32_t v[3];
            for (int k = 0; k < 3; ++k)
            {
                const float f = (xyz[3 * i + k] - ws_min[k]) * out.inv_scale;
                v[k] = (uint32_t) (uint16_t) std::clamp(f, 0.0f, (float) (gw - 1));
            }
            uint8_t &col = column_used[(size_t) v[0] * gw + v[1]];
            if (!col)
            {
                if (++columns > max_columns) return MvtStatus::ztable_pool;
                col = 1;
            }
            uint32_t &cell = out.cells[((size_t) v[0] * gw + v[1]) * gw + v[2]];
            if (cell == kNone)
            {
                if (counts.size() + 1 > max_voxels) return MvtStatus::point_pool;
                cell = (uint32_t) counts.size();
                counts.push_back(0);
            }
            if (counts[cell] >= out.capacity) return MvtStatus::voxel_capacity;
            ++counts[cell];
            vox_of[i] = cell;
        }
        // pass 2: compact SoA fill + boxes
        const size_t nv = counts.size();
        out.vox_offset.assign(nv + 1, 0);
        for (size_t v = 0; v < nv; ++v) out.vox_offset[v + 1] = out.vox_offset[v] + counts[v];
        out.px.resize(n), out.py.resize(n), out.pz.resize(n);
        constexpr float inf = std::numeric_limits<float>::infinity();
        out.vox_bbox.resize(nv * 6);
        for (size_t v = 0; v < nv; ++v)
            for (int k = 0; k < 3; ++k)
            {
                out.vox_bbox[6 * v + k] = inf;
                out.vox_bbox[6 * v + 3 + k] = -inf;
            }
        std::vector<uint32_t> cursor(out.vox_offset.begin(), out.vox_offset.end() - 1);
        for (size_t i = 0; i < n; ++i)
        {
            const uint32_t v = vox_of[i];
            const uint32_t at = cursor[v]++;
            out.px[at] = xyz[3 * i], out.py[at] = xyz[3 * i + 1], out.pz[at] = xyz[3 * i + 2];
            for (int k = 0; k < 3; ++k)
            {
                out.vox_bbox[6 * (size_t) v + k] = std::min(out.vox_bbox[6 * (size_t) v + k], xyz[3 * i + k]);
                out.vox_bbox[6 * (size_t) v + 3 + k] = std::max(out.vox_bbox[6 * (size_t) v + 3 + k], xyz[3 * i + k]);
            }
        }
        for (int k = 0; k < 3; ++k)
        {
            out.gmin[k] = std::numeric_limits<float>::max();
            out.gmax[k] = std::numeric_limits<float>::lowest();
        }
        for (size_t v = 0; v < nv; ++v)
            for (int k = 0; k < 3; ++k)
            {
                out.gmin[k] = std::min(out.gmin[k], out.vox_bbox[6 * v + k]);
                out.gmax[k] = std::max(out.gmax[k], out.vox_bbox[6 * v + 3 + k]);
            }
        return MvtStatus::ok;
    }
}  // namespace vmv
