// TEST INFRASTRUCTURE ONLY (oracle/_ref): thin extern "C" exports over the
// reference's own dependency-free SIMD layer, compiled from where it lies
// under /root/reference (never copied).  Only headers that need no third-party
// library are used: vamp/vector.hh (+vector/{interface,avx,utils}.hh,
// constants.hh, utils.hh) and vamp/random/halton.hh.  Everything that pulls
// Eigen / pdqsort / nigh (collision/*, robots/*, planning/*) is unbuildable in
// this image and is NOT stubbed.
//
// Used by tests/ and tools/ to pin the arithmetic contract of SURVEY.md §2
// (sin, cos, approximate sqrt, hsum / l2_norm, test_zero, Halton sequence).
#include <array>
#include <cstddef>
#include <cstdint>
#include <cmath>
#include <vamp/vector.hh>
#include <vamp/random/halton.hh>

using V8 = vamp::FloatVector<8>;

namespace
{
    template <typename F>
    void map8(const float *in, float *out, std::size_t n, F f)
    {
        alignas(32) float a[8], b[8];
        for (std::size_t i = 0; i < n; i += 8)
        {
            for (int k = 0; k < 8; ++k) a[k] = (i + k < n) ? in[i + k] : 0.F;
            V8 v(a);
            V8 r = f(v);
            r.to_array(b);
            for (int k = 0; k < 8 && i + k < n; ++k) out[i + k] = b[k];
        }
    }

    // Template argument for the reference's Halton<Robot>: only the members
    // halton.hh itself names.  Scale/offset are supplied at run time.
    template <std::size_t DIM>
    struct RobotSpace
    {
        static constexpr std::size_t dimension = DIM;
        using Configuration = vamp::FloatVector<DIM>;
        static inline std::array<float, DIM> s_m{};
        static inline std::array<float, DIM> s_a{};
        inline static void scale_configuration(Configuration &q) noexcept
        {
            q = q * Configuration(s_m) + Configuration(s_a);
        }
    };

    template <std::size_t DIM>
    void halton_run(const float *s_m, const float *s_a, std::size_t count, float *out)
    {
        using R = RobotSpace<DIM>;
        for (std::size_t j = 0; j < DIM; ++j)
        {
            R::s_m[j] = s_m[j];
            R::s_a[j] = s_a[j];
        }
        vamp::rng::Halton<R> h;
        alignas(32) float buf[R::Configuration::num_scalars_rounded];
        for (std::size_t i = 0; i < count; ++i)
        {
            auto q = h.next();
            q.to_array(buf);
            for (std::size_t j = 0; j < DIM; ++j) out[i * DIM + j] = buf[j];
        }
    }
}  // namespace

extern "C"
{
    // reference: vector/avx.hh:455-548 via interface.hh sin()
    void ref_sin(const float *in, float *out, std::size_t n)
    {
        map8(in, out, n, [](V8 v) { return v.sin(); });
    }
    // reference: vector/interface.hh:447-458
    void ref_cos(const float *in, float *out, std::size_t n)
    {
        map8(in, out, n, [](V8 v) { return v.cos(); });
    }
    // reference: vector/avx.hh:411-415 (v * rsqrt_ps(v); vendor-defined low bits)
    void ref_sqrt_approx(const float *in, float *out, std::size_t n)
    {
        map8(in, out, n, [](V8 v) { return v.sqrt(); });
    }
    // reference: interface.hh:397-420 (l2_norm of a Configuration of `dim` scalars)
    float ref_l2_norm(const float *q, std::size_t dim)
    {
        alignas(32) float a[16] = {0};
        for (std::size_t j = 0; j < dim && j < 16; ++j) a[j] = q[j];
        if (dim <= 8)
        {
            vamp::FloatVector<8> v(a);
            return v.l2_norm();
        }
        vamp::FloatVector<16> v(a);
        return v.l2_norm();
    }
    // reference: interface.hh:257-277 — true iff no lane has its sign bit set
    int ref_test_zero(const float *lanes8)
    {
        alignas(32) float a[8];
        for (int k = 0; k < 8; ++k) a[k] = lanes8[k];
        return V8(a).test_zero() ? 1 : 0;
    }
    // reference: random/halton.hh:75-108, default bases; returns count*dim floats
    int ref_halton(std::size_t dim, const float *s_m, const float *s_a, std::size_t count, float *out)
    {
        switch (dim)
        {
            case 6: halton_run<6>(s_m, s_a, count, out); return 0;
            case 7: halton_run<7>(s_m, s_a, count, out); return 0;
            case 8: halton_run<8>(s_m, s_a, count, out); return 0;
            case 14: halton_run<14>(s_m, s_a, count, out); return 0;
            default: return -1;
        }
    }
}
