// Test program: the product's host-side builders (plain C++ headers under vamp_mvt_amd/csrc: CAPT, MVT, broad-phase grid)
// under AddressSanitizer + UBSan, with the CAPT arrays compared against the oracle's.  Built and run by
// tests/test_native_sanitize.py (CPU only; GPU sanitizers are not available on this pool).
#include "../../vamp_mvt_amd/csrc/vmv_capt_build.h"
#include "../../vamp_mvt_amd/csrc/vmv_grid_build.h"
#include "../../vamp_mvt_amd/csrc/vmv_mvt_build.h"

extern "C"
{
#include "../../oracle/vamp_oracle.h"
}

#include <cstdio>
#include <cstring>
#include <random>

static int fail(const char *what)
{
    std::printf("FAIL: %s\n", what);
    return 1;
}

int main()
{
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> ang(0.f, 6.2831853f), rad(0.5f, 1.1f), zz(0.f, 1.3f);
    for (size_t n : {2u, 17u, 1000u, 4096u, 9000u})
    {
        std::vector<float> pts(3 * n);
        for (size_t i = 0; i < n; ++i)
        {
            const float a = ang(rng), r = rad(rng);
            pts[3 * i] = r * std::cos(a), pts[3 * i + 1] = r * std::sin(a), pts[3 * i + 2] = zz(rng);
        }
        for (const float r_max : {0.08f, 0.24f})
        {
            vmv::CaptArrays a;
            if (!vmv::build_capt(pts.data(), n, 0.012f, r_max, 0.0025f, a)) return fail("build_capt");
            vo_env *e = vo_env_create();
            if (vo_env_add_capt(e, pts.data(), n, 0.012f, r_max, 0.0025f) != 0) return fail("vo_env_add_capt");
            vo_capt_view v;
            vo_env_capt_view(e, 0, &v);
            if (v.nlog2 != a.nlog2 || v.n_aff_vectors != a.n_aff_vectors()) return fail("capt sizes");
            if (std::memcmp(v.aff_starts, a.aff_starts.data(), a.aff_starts.size() * 4)) return fail("aff_starts");
            if (std::memcmp(v.aabbs, a.aabbs.data(), a.aabbs.size() * 4)) return fail("aabbs");
            if (std::memcmp(v.aff_x, a.aff[0].data(), a.aff[0].size() * 4) ||
                std::memcmp(v.aff_y, a.aff[1].data(), a.aff[1].size() * 4) ||
                std::memcmp(v.aff_z, a.aff[2].data(), a.aff[2].size() * 4))
                return fail("affordances");
            vo_env_destroy(e);
        }
        // MVT: the builder either succeeds or reports one of the reference's pool limits, never touches memory it
        // does not own
        const float lo[3] = {-1.2f, -1.2f, -0.1f}, hi[3] = {1.2f, 1.2f, 2.3f};
        vmv::MvtArrays m;
        const vmv::MvtStatus st = vmv::build_mvt(pts.data(), n, 0.012f, 0.08f, lo, hi, 0.0025f, m);
        if (st == vmv::MvtStatus::ok && m.vox_offset.back() == 0) return fail("mvt holds no points");
    }
    // grid: a handful of primitives, every class radius
    std::vector<float> sph = {0.5f, 0.1f, 0.3f, 0.1f, -0.4f, 0.6f, 0.9f, 0.05f};
    std::vector<vmv::GridPrim> prims = {{0, sph.data(), 0, 0}, {0, sph.data() + 4, 0, 1}};
    for (const double R : {0.02, 0.1, 0.5})
    {
        vmv::GridArrays g;
        if (!vmv::build_grid(prims, 1, R, g)) return fail("build_grid");
        if (g.cells.size() != (size_t) g.dims[0] * g.dims[1] * g.dims[2]) return fail("grid size");
    }
    std::printf("OK\n");
    return 0;
}
