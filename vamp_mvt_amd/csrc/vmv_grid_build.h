// vmv_grid_build.h — host-side broad-phase grid for the bounding-sphere ("gate") pass.
//
// Not part of the reference's algorithm: a conservative index that only decides WHICH primitives a lane evaluates
// the reference's exact predicates on.  For a query sphere of radius <= R whose centre falls in a cell, every
// primitive it can collide with is listed in that cell's candidate words:
//
//   primitive p is listed for cell c  <=>  g_p(centre_c) < R + margin + L_p * half_diagonal
//
// where g_p(x) is the reference's own distance expression for p evaluated in double precision (collision of a
// sphere (x, r) is g_p(x) < r up to fp32 rounding), and L_p bounds how fast g_p can change with x (1 for spheres
// and well-formed cuboids/capsules; larger if the caller supplied non-unit axes or an inconsistent rdv).  The
// margin (1e-3 m) is three orders of magnitude above any fp32 effect.  Outside the grid box — the primitives'
// bounding box inflated by R + margin — no primitive can be touched, and the candidate set is empty.
// Word layout = the per-list 32-primitive candidate words of vmv_device.h (wbase_* of EnvDev).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace vmv
{
    struct GridArrays
    {
        uint32_t dims[3] = {0, 0, 0};
        float origin[3] = {0, 0, 0};
        float inv_cell = 0.f;
        uint32_t words = 0;               // candidate words per cell
        std::vector<uint32_t> cells;      // dims[0]*dims[1]*dims[2] * words
    };

    struct GridPrim  // one primitive in list order with its candidate word/bit
    {
        int type;          // 0 sphere, 1 capsule, 2 z-capsule, 3 cuboid, 4 z-cuboid
        const float *p;    // canonical parameters (sphere: x y z r; capsule: 8; cuboid: 15)
        uint32_t word, bit;
    };

    namespace grid_detail
    {
        inline double capsule_g(const float *c, const double x[3], bool z_aligned, double &lip)
        {
            const double p1[3] = {c[0], c[1], c[2]};
            const double v[3] = {z_aligned ? 0.0 : c[3], z_aligned ? 0.0 : c[4], c[5]};
            const double rdv = c[7];
            double dot = 0, vv = 0;
            for (int k = 0; k < 3; ++k)
            {
                dot += (x[k] - p1[k]) * v[k];
                vv += v[k] * v[k];
            }
            const double t = std::min(std::max(dot * rdv, 0.0), 1.0);
            double d2 = 0;
            for (int k = 0; k < 3; ++k)
            {
                const double q = x[k] - (p1[k] + v[k] * t);
                d2 += q * q;
            }
            lip = 1.0 + vv * std::fabs(rdv);  // the clamped projection moves the foot point at most this fast
            return std::sqrt(d2) - c[6];
        }

        inline double cuboid_g(const float *c, const double x[3], bool z_aligned, double &lip)
        {
            double a[3][3];
            for (int i = 0; i < 3; ++i)
                for (int k = 0; k < 3; ++k) a[i][k] = c[3 + 3 * i + k];
            if (z_aligned)  // collision/sphere_cuboid.hh:35-52 uses only the xy parts of axes 1, 2 and |zs|
            {
                a[0][2] = a[1][2] = 0.0;
                a[2][0] = a[2][1] = 0.0;
                a[2][2] = 1.0;
            }
            double s = 0;
            for (int i = 0; i < 3; ++i)
            {
                double d = 0;
                for (int k = 0; k < 3; ++k) d += a[i][k] * (x[k] - c[k]);
                const double e = std::max(std::fabs(d) - (double) c[12 + i], 0.0);
                s += e * e;
            }
            // Lipschitz bound: sqrt of the largest absolute row sum of A A^T (= 1 for orthonormal axes)
            double worst = 0;
            for (int i = 0; i < 3; ++i)
            {
                double row = 0;
                for (int j = 0; j < 3; ++j)
                {
                    double d = 0;
                    for (int k = 0; k < 3; ++k) d += a[i][k] * a[j][k];
                    row += std::fabs(d);
                }
                worst = std::max(worst, row);
            }
            lip = std::sqrt(std::max(worst, 1.0));
            return std::sqrt(s);
        }
    }  // namespace grid_detail

    // bbox of the primitive (for the grid extent), conservative
    inline void grid_prim_box(const GridPrim &g, double lo[3], double hi[3])
    {
        const float *p = g.p;
        if (g.type == 0)
            for (int k = 0; k < 3; ++k) lo[k] = p[k] - p[3], hi[k] = p[k] + p[3];
        else if (g.type <= 2)
            for (int k = 0; k < 3; ++k)
            {
                const double a = p[k], b = (double) p[k] + ((g.type == 2 && k < 2) ? 0.0 : (double) p[3 + k]);
                lo[k] = std::min(a, b) - p[6], hi[k] = std::max(a, b) + p[6];
            }
        else
        {
            double reach = 0;  // half diagonal measured with the caller's axes
            for (int i = 0; i < 3; ++i) reach += (double) p[12 + i] * (double) p[12 + i];
            reach = std::sqrt(reach) * 1.8 + 1e-6;  // generous for slightly non-orthonormal axes
            for (int k = 0; k < 3; ++k) lo[k] = p[k] - reach, hi[k] = p[k] + reach;
        }
    }

    inline bool build_grid(const std::vector<GridPrim> &prims, uint32_t words, double R, GridArrays &out)
    {
        out = GridArrays{};
        out.words = words;
        if (prims.empty() || words == 0) return false;
        constexpr double kMargin = 1e-3;
        double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};
        for (const auto &g : prims)
        {
            double a[3], b[3];
            grid_prim_box(g, a, b);
            for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], a[k]), hi[k] = std::max(hi[k], b[k]);
        }
        double vol = 1;
        for (int k = 0; k < 3; ++k)
        {
            lo[k] -= R + kMargin;
            hi[k] += R + kMargin;
            if (!(hi[k] - lo[k] < 1e4)) return false;  // absurd extents: no grid, full loops
            vol *= hi[k] - lo[k];
        }
        // cell edge: at least kMinCell, and no more than kMaxCells cells (VMV_GRID_CELLS / VMV_GRID_MIN_CELL: tuning knobs)
        double max_cells = 24000.0, min_cell = 0.06;
        if (const char *e = std::getenv("VMV_GRID_CELLS")) max_cells = std::max(1000.0, std::atof(e));
        if (const char *e = std::getenv("VMV_GRID_MIN_CELL")) min_cell = std::max(0.005, std::atof(e));
        double h = std::max(min_cell, std::cbrt(vol / max_cells));
        for (int k = 0; k < 3; ++k)
        {
            out.dims[k] = (uint32_t) std::max(1.0, std::ceil((hi[k] - lo[k]) / h));
            out.origin[k] = (float) lo[k];
        }
        // the device computes the cell as floor((x - origin) * inv_cell) in fp32: use the fp32 values here too and
        // let every cell claim a slightly larger cube so that rounding at cell borders stays covered
        out.inv_cell = (float) (1.0 / h);
        const double hf = 1.0 / (double) out.inv_cell;
        const double half_diag = 0.5 * hf * std::sqrt(3.0) * 1.001 + 1e-5;
        out.cells.assign((size_t) out.dims[0] * out.dims[1] * out.dims[2] * words, 0u);
        for (uint32_t ix = 0; ix < out.dims[0]; ++ix)
            for (uint32_t iy = 0; iy < out.dims[1]; ++iy)
                for (uint32_t iz = 0; iz < out.dims[2]; ++iz)
                {
                    const double c[3] = {(double) out.origin[0] + (ix + 0.5) * hf, (double) out.origin[1] + (iy + 0.5) * hf,
                                         (double) out.origin[2] + (iz + 0.5) * hf};
                    uint32_t *cell = &out.cells[(((size_t) ix * out.dims[1] + iy) * out.dims[2] + iz) * words];
                    for (const auto &g : prims)
                    {
                        double lip = 1.0, d;
                        if (g.type == 0)
                        {
                            const double dx = c[0] - g.p[0], dy = c[1] - g.p[1], dz = c[2] - g.p[2];
                            d = std::sqrt(dx * dx + dy * dy + dz * dz) - g.p[3];
                        }
                        else if (g.type <= 2)
                            d = grid_detail::capsule_g(g.p, c, g.type == 2, lip);
                        else
                            d = grid_detail::cuboid_g(g.p, c, g.type == 4, lip);
                        if (d < R + kMargin + lip * half_diag) cell[g.word] |= 1u << g.bit;
                    }
                }
        return true;
    }
}  // namespace vmv
