// vmv_api.hip — C ABI (include/vamp_mvt_amd.h): environment builder/upload, robot table, dispatch to the
// per-robot kernel translation units (gen/tu_<robot>.hip).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see __graft_entry__.build()).  There is no CPU
// fallback anywhere in this library: every compute entry point needs a HIP device.
#include "../../include/vamp_mvt_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "vmv_capt_build.h"
#include "vmv_mvt_build.h"
#include "vmv_grid_build.h"
#include "vmv_common.h"

// ---------------------------------------------------------------------------------------------------------
// host-side robot table
// ---------------------------------------------------------------------------------------------------------
struct vmv_link_reach  // reach certificate of one link (tools/gen_hip.py: link_samples)
{
    int group;       // index of the link's environment group = bit in EnvDev::link_skip
    int n;           // sample centres
    float slack;     // every centre the link can have lies within `slack` of a sample
    float radius;    // the link's bounding-sphere radius
    const float (*samples)[3];
};
struct vmv_robot_info
{
    const char *name;
    int dimension, n_spheres, resolution;
    float min_radius, max_radius, max_bounding_radius;
    float grid_radius[4];  // largest bounding radius of each broad-phase grid class (vmv::kGridClasses)
    float lower[16], span[16], descale[16];
    const char *end_effector;
    const char *joint_names[16];
    int n_self_pairs;
    const uint16_t (*self_pairs)[2];  // fine pairs of the self-collision groups, in group order
    int n_reach;
    const vmv_link_reach *reach;
};
#include "gen/robots_host.inc"

namespace
{
    thread_local std::string g_last_error;

    int hip_fail(hipError_t e, const char *what)
    {
        g_last_error = std::string(what) + ": " + hipGetErrorString(e);
        return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver) ?
                   VMV_ERR_NO_DEVICE :
                   VMV_ERR_HIP;
    }
#define VMV_HIP(call)                                  \
    do                                                 \
    {                                                  \
        hipError_t e_ = (call);                        \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)

    int require_device()
    {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0)
        {
            g_last_error = "no HIP device (this library has no CPU path)";
            return VMV_ERR_NO_DEVICE;
        }
        return VMV_OK;
    }

    using vmv::kMaxPrimFloats;

    const vmv::RobotLaunchers *const kLaunchers[] = {&vmv::kPandaLaunchers, &vmv::kUr5Launchers, &vmv::kFetchLaunchers,
                                                     &vmv::kBaxterLaunchers};
}  // namespace

namespace vmv
{
    int hip_status(hipError_t e, const char *what) { return hip_fail(e, what); }

    // uniform configurations inside the joint bounds (bench input generator, counter based)
    __global__ void fill_uniform_kernel(float *q, size_t total, int dim, uint64_t seed, const float *lower,
                                        const float *span)
    {
        const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= total) return;
        uint64_t z = seed + 0x9e3779b97f4a7c15ull * (uint64_t) (i + 1);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        z = z ^ (z >> 31);
        const float u = (float) (z >> 40) * (1.0f / 16777216.0f);
        const int j = (int) (i % (size_t) dim);
        q[i] = lower[j] + span[j] * u;
    }

    // Halton sample (skip + 1 + row) of the reference's sequence, element j (random/halton.hh:75-108): the
    // incremental float arithmetic there keeps exact integers n, d = b^k, so the value is n / d with n the
    // digit-reversed index; then Robot::scale_configuration (q * s_m + s_a, two roundings).
    __global__ void halton_kernel(float *q, size_t total, int dim, uint64_t skip, const float *lower, const float *span)
    {
        const size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
        if (idx >= total) return;
        const uint32_t primes[16] = {3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59};
        const int j = (int) (idx % (size_t) dim);
        uint64_t k = skip + idx / (size_t) dim + 1;
        const uint32_t b = primes[j];
        uint32_t n = 0, d = 1;
        while (k > 0)
        {
            n = n * b + (uint32_t) (k % b);
            d *= b;
            k /= b;
        }
        const float u = (float) n / (float) d;
        q[idx] = u * span[j] + lower[j];
    }

    // sphere_environment_in_collision (collision/validity.hh:47-158) for a batch of free spheres, one lane per sphere
    // (each sphere is its own replicated rake): the primitive lists with their sorted early-break, heightfields,
    // CAPT and MVT clouds, through the same device functions the robot kernels use (counted loops, no grid).
    // robot_spheres (optional, [n_robot] x y z r of one configuration, <robot>.filter_self_from_pointcloud): a sphere
    // also "hits" when it overlaps one of them (sphere_sphere_sql2 < 0, strict, robot_helper.hh:306-307)
    __global__ __launch_bounds__(kBlock) void spheres_env_kernel(const EnvDev *__restrict__ env, const float4 *__restrict__ spheres,
                                                                  const size_t n, uint8_t *__restrict__ hits,
                                                                  const float4 *__restrict__ robot_spheres, const int n_robot)
    {
        extern __shared__ __align__(16) float smem[];
        const uint32_t n_floats = env->n_floats;  // a multiple of 4
        for (uint32_t i = threadIdx.x; i < n_floats; i += blockDim.x) smem[i] = env->prims[i];
        __syncthreads();
        // [primitive block | CAPT hit-flag rows]; no radius table (free spheres bring their own radius)
        const EnvView E{(env_cptr) env, (lds_cptr) smem, 0u, (lds_cptr) smem + n_floats + kCaptFlagWords};
        const size_t rounds = (n + (size_t) gridDim.x * kBlock - 1) / ((size_t) gridDim.x * kBlock);
        for (size_t k = 0; k < rounds; ++k)  // every wave runs every round: env_hit uses wave-wide votes
        {
            const size_t i = (k * gridDim.x + blockIdx.x) * (size_t) kBlock + threadIdx.x;
            const bool active = i < n;
            const float4 s = active ? spheres[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            bool hit = env_hit<1, 0, kEnvFull>(E, s.x, s.y, s.z, s.w, active, nullptr);
            for (int k = 0; k < n_robot; ++k)  // wave-uniform records (scalar loads)
            {
                const float4 a = robot_spheres[k];
                hit |= neg(sphere_sphere_sql2(a.x, a.y, a.z, a.w, s.x, s.y, s.z, s.w));
            }
            if (active) hits[i] = hit ? 1 : 0;
        }
    }
}  // namespace vmv

// ---------------------------------------------------------------------------------------------------------
// environment (host builder + device image)
// ---------------------------------------------------------------------------------------------------------
struct vmv_env
{
    struct Sphere
    {
        float x, y, z, r, min_d;
    };
    struct Cuboid
    {
        float p[15];
        float min_d;
    };
    struct Capsule
    {
        float p[8];
        float min_d;
    };
    std::vector<Sphere> spheres;
    std::vector<Capsule> capsules, z_capsules;
    std::vector<Cuboid> cuboids, z_cuboids;
    std::vector<vmv::CaptArrays> capts;
    std::vector<vmv::MvtArrays> mvts;
    struct HeightField
    {
        float c[3], inv[3];
        size_t xd, yd;
        std::vector<float> data;
    };
    std::vector<HeightField> heightfields;
    bool attached = false;
    float attach_tf[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};  // row-major 3 x 4 (R | t), relative to the end effector
    std::vector<float> attach_spheres;                            // [n][4] = x y z r in that frame

    bool finalized = false;
    int device = -1;
    vmv::EnvLaunch launch[8]{};  // per robot: host + device copies of the kernel-side description (grids differ)
    // the robot-specific part (broad-phase grids, static links) is built on the robot's first use
    vmv::EnvDev base{};
    std::vector<vmv::GridPrim> grid_prims;
    uint32_t grid_words = 0;
    std::once_flag robot_once[8];
    std::mutex robot_mutex;  // serialises the lazy builds (they append to `allocations`)
    int robot_status[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::string robot_error[8];
    std::vector<void *> allocations;
};

namespace
{
    float dot3h(float ax, float ay, float az, float bx, float by, float bz) { return (ax * bx) + (ay * by) + (az * bz); }
    float clamp_scalar(float v, float lo, float hi) { return std::max(std::min(v, hi), lo); }  // collision/math.hh:47-51

    // collision/shapes.hh:52-67
    float cuboid_min_distance(const float *p)
    {
        const float x = p[0], y = p[1], z = p[2];
        const float d1 = dot3h(-x, -y, -z, p[3], p[4], p[5]);
        const float d2 = dot3h(-x, -y, -z, p[6], p[7], p[8]);
        const float d3 = dot3h(-x, -y, -z, p[9], p[10], p[11]);
        const float v1 = clamp_scalar(d1, -p[12], p[12]);
        const float v2 = clamp_scalar(d2, -p[13], p[13]);
        const float v3 = clamp_scalar(d3, -p[14], p[14]);
        const float xn = x + p[3] * v1 + p[6] * v2 + p[9] * v3;
        const float yn = y + p[4] * v1 + p[7] * v2 + p[10] * v3;
        const float zn = z + p[5] * v1 + p[8] * v2 + p[11] * v3;
        return std::sqrt(xn * xn + yn * yn + zn * zn);
    }

    // collision/shapes.hh:165-189
    float capsule_min_distance(const float *p)
    {
        const float x1 = p[0], y1 = p[1], z1 = p[2], xv = p[3], yv = p[4], zv = p[5], r = p[6], rdv = p[7];
        const float t = clamp_scalar(dot3h(-x1, -y1, -z1, xv, yv, zv) * rdv, 0.F, 1.F);
        const float xp = x1 + xv * t, yp = y1 + yv * t, zp = z1 + zv * t;
        float xo = -xp, yo = -yp, zo = -zp;
        const float ol = std::sqrt(dot3h(xo, yo, zo, xo, yo, zo));
        xo = xo / ol;
        yo = yo / ol;
        zo = zo / ol;
        const float ro = clamp_scalar(ol, 0.F, r);
        const float xn = xp + ro * xo, yn = yp + ro * yo, zn = zp + ro * zo;
        return std::sqrt(xn * xn + yn * yn + zn * zn);
    }

    template <typename T>
    void sort_by_min_distance(std::vector<T> &v)
    {
        std::stable_sort(v.begin(), v.end(), [](const T &a, const T &b) { return a.min_d < b.min_d; });
    }

    // the getters return the lists as the kernels will see them: sorted by min_distance (stable), finalized or not
    template <typename T>
    std::vector<T> sorted_copy(const std::vector<T> &v)
    {
        std::vector<T> c(v);
        sort_by_min_distance(c);
        return c;
    }

    template <typename T>
    int upload(vmv_env *env, const std::vector<T> &host, const T **dev_ptr)
    {
        void *d = nullptr;
        const size_t bytes = std::max<size_t>(host.size() * sizeof(T), 16);
        VMV_HIP(hipMalloc(&d, bytes));
        env->allocations.push_back(d);
        if (!host.empty()) VMV_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
        *dev_ptr = static_cast<const T *>(d);
        return VMV_OK;
    }
}  // namespace

extern "C"
{
    const char *vmv_status_string(int status)
    {
        switch (status)
        {
            case VMV_OK: return "ok";
            case VMV_ERR_INVALID_ARGUMENT: return "invalid argument";
            case VMV_ERR_NO_DEVICE: return "no HIP device";
            case VMV_ERR_HIP: return "HIP runtime error";
            case VMV_ERR_CAPACITY: return "environment exceeds on-chip staging capacity";
            case VMV_ERR_NOT_FINALIZED: return "environment not finalized";
            case VMV_ERR_UNKNOWN_ROBOT: return "unknown robot";
            case VMV_ERR_FINALIZED: return "environment already finalized";
            default: return "unknown status";
        }
    }
    int vmv_abi_version(void) { return 1; }
    const char *vmv_last_error(void) { return g_last_error.c_str(); }

    int vmv_device_count(int *count)
    {
        if (!count) return VMV_ERR_INVALID_ARGUMENT;
        *count = 0;
        hipError_t e = hipGetDeviceCount(count);
        if (e != hipSuccess)
        {
            *count = 0;
            return hip_fail(e, "hipGetDeviceCount");
        }
        return VMV_OK;
    }
    int vmv_set_device(int device)
    {
        VMV_HIP(hipSetDevice(device));
        return VMV_OK;
    }
    int vmv_get_device(int *device)
    {
        if (!device) return VMV_ERR_INVALID_ARGUMENT;
        VMV_HIP(hipGetDevice(device));
        return VMV_OK;
    }

    // ---- robots ----
    int vmv_num_robots(void) { return kNumRobots; }
    static bool robot_ok(int r) { return r >= 0 && r < kNumRobots; }
    const char *vmv_robot_name(int r) { return robot_ok(r) ? kRobots[r].name : nullptr; }
    int vmv_robot_id(const char *name)
    {
        if (!name) return -1;
        for (int i = 0; i < kNumRobots; ++i)
            if (std::strcmp(kRobots[i].name, name) == 0) return i;
        return -1;
    }
    int vmv_robot_dimension(int r) { return robot_ok(r) ? kRobots[r].dimension : -1; }
    int vmv_robot_n_spheres(int r) { return robot_ok(r) ? kRobots[r].n_spheres : -1; }
    int vmv_robot_resolution(int r) { return robot_ok(r) ? kRobots[r].resolution : -1; }
    int vmv_robot_min_max_radii(int r, float *mn, float *mx)
    {
        if (!robot_ok(r)) return VMV_ERR_UNKNOWN_ROBOT;
        if (mn) *mn = kRobots[r].min_radius;
        if (mx) *mx = kRobots[r].max_radius;
        return VMV_OK;
    }
    int vmv_robot_bounds(int r, float *lower, float *span, float *descale)
    {
        if (!robot_ok(r)) return VMV_ERR_UNKNOWN_ROBOT;
        const size_t b = sizeof(float) * (size_t) kRobots[r].dimension;
        if (lower) std::memcpy(lower, kRobots[r].lower, b);
        if (span) std::memcpy(span, kRobots[r].span, b);
        if (descale) std::memcpy(descale, kRobots[r].descale, b);
        return VMV_OK;
    }
    const char *vmv_robot_joint_name(int r, int j)
    {
        return (robot_ok(r) && j >= 0 && j < kRobots[r].dimension) ? kRobots[r].joint_names[j] : nullptr;
    }
    const char *vmv_robot_end_effector(int r) { return robot_ok(r) ? kRobots[r].end_effector : nullptr; }

    // ---- environment ----
    int vmv_env_create(vmv_env **out)
    {
        if (!out) return VMV_ERR_INVALID_ARGUMENT;
        *out = new (std::nothrow) vmv_env();
        return *out ? VMV_OK : VMV_ERR_INVALID_ARGUMENT;
    }
    int vmv_env_destroy(vmv_env *env)
    {
        if (!env) return VMV_OK;
        for (void *p : env->allocations) (void) hipFree(p);
        delete env;
        return VMV_OK;
    }
#define VMV_MUTABLE(env)                                  \
    if (!(env)) return VMV_ERR_INVALID_ARGUMENT;          \
    if ((env)->finalized) return VMV_ERR_FINALIZED;

    int vmv_env_add_sphere(vmv_env *env, float x, float y, float z, float r)
    {
        VMV_MUTABLE(env)
        // collision/shapes.hh:236-239
        env->spheres.push_back({x, y, z, r, std::sqrt(x * x + y * y + z * z) - r});
        return VMV_OK;
    }
    int vmv_env_add_cuboid(vmv_env *env, const float *p)
    {
        VMV_MUTABLE(env)
        if (!p) return VMV_ERR_INVALID_ARGUMENT;
        vmv_env::Cuboid c;
        std::memcpy(c.p, p, sizeof(c.p));
        c.min_d = cuboid_min_distance(p);
        (p[11] == 1.0f ? env->z_cuboids : env->cuboids).push_back(c);  // environment.cc:123-130
        return VMV_OK;
    }
    int vmv_env_add_capsule(vmv_env *env, const float *p)
    {
        VMV_MUTABLE(env)
        if (!p) return VMV_ERR_INVALID_ARGUMENT;
        vmv_env::Capsule c;
        std::memcpy(c.p, p, sizeof(c.p));
        c.min_d = capsule_min_distance(p);
        ((p[3] == 0.0f && p[4] == 0.0f) ? env->z_capsules : env->capsules).push_back(c);  // environment.cc:137-144
        return VMV_OK;
    }
    int vmv_env_add_heightfield(vmv_env *env, const float *center, const float *scale, size_t xd, size_t yd,
                                const float *data)
    {
        VMV_MUTABLE(env)
        if (!center || !scale || !data || xd == 0 || yd == 0 || xd * yd > (size_t) 1 << 30) return VMV_ERR_INVALID_ARGUMENT;
        if (env->heightfields.size() >= (size_t) vmv::kMaxHeightFields) return VMV_ERR_CAPACITY;
        vmv_env::HeightField h;
        for (int k = 0; k < 3; ++k)
        {
            h.c[k] = center[k];
            h.inv[k] = 1.F / scale[k];  // factory.hh:376-386
        }
        h.xd = xd;
        h.yd = yd;
        h.data.assign(data, data + xd * yd);
        env->heightfields.push_back(std::move(h));
        return VMV_OK;
    }
    int vmv_env_attach(vmv_env *env, const float *tf16, const float *spheres, size_t n)
    {
        VMV_MUTABLE(env)
        if (!tf16 || (n && !spheres)) return VMV_ERR_INVALID_ARGUMENT;
        if (n > (size_t) vmv::kMaxAttachSpheres) return VMV_ERR_CAPACITY;
        env->attached = true;
        for (int i = 0; i < 12; ++i) env->attach_tf[i] = tf16[i];  // the first three rows of the row-major 4 x 4
        env->attach_spheres.assign(spheres, spheres + 4 * n);
        return VMV_OK;
    }
    int vmv_env_detach(vmv_env *env)
    {
        VMV_MUTABLE(env)
        env->attached = false;
        env->attach_spheres.clear();
        return VMV_OK;
    }
    int vmv_env_heightfield_count(const vmv_env *env, size_t *count)
    {
        if (!env || !count) return VMV_ERR_INVALID_ARGUMENT;
        *count = env->heightfields.size();
        return VMV_OK;
    }
    int vmv_env_add_capt_pointcloud(vmv_env *env, const float *pts, size_t n, float r_min, float r_max, float r_point,
                                    uint64_t *build_ns)
    {
        VMV_MUTABLE(env)
        if (!pts || n < 2) return VMV_ERR_INVALID_ARGUMENT;
        if (env->capts.size() >= (size_t) vmv::kMaxCapt) return VMV_ERR_CAPACITY;
        const auto t0 = std::chrono::steady_clock::now();
        vmv::CaptArrays arrays;
        if (!vmv::build_capt(pts, n, r_min, r_max, r_point, arrays)) return VMV_ERR_INVALID_ARGUMENT;
        env->capts.push_back(std::move(arrays));
        if (build_ns)
            *build_ns = (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(
                            std::chrono::steady_clock::now() - t0)
                            .count();
        return VMV_OK;
    }

    int vmv_env_add_capt_pointcloud_gpu(vmv_env *env, const float *pts, size_t n, float r_min, float r_max, float r_point,
                                        uint64_t *build_ns, uint64_t *device_ns)
    {
        VMV_MUTABLE(env)
        if (!pts || n < 2) return VMV_ERR_INVALID_ARGUMENT;
        if (env->capts.size() >= (size_t) vmv::kMaxCapt) return VMV_ERR_CAPACITY;
        int rc = require_device();
        if (rc != VMV_OK) return rc;
        const auto t0 = std::chrono::steady_clock::now();
        vmv::CaptArrays arrays;
        rc = vmv::build_capt_device(pts, n, r_min, r_max, r_point, arrays, device_ns);
        if (rc != VMV_OK) return rc;
        for (void *p : {(void *) arrays.dev.tests, (void *) arrays.dev.aff_starts, (void *) arrays.dev.aabbs, (void *) arrays.dev.aff})
            if (p) env->allocations.push_back(p);  // freed with the environment
        env->capts.push_back(std::move(arrays));
        if (build_ns)
            *build_ns = (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(
                            std::chrono::steady_clock::now() - t0)
                            .count();
        return VMV_OK;
    }

    int vmv_env_add_mvt_pointcloud(vmv_env *env, const float *pts, size_t n, float r_min, float r_max,
                                   const float *ws_min, const float *ws_max, float r_point, uint64_t *build_ns,
                                   int *reason)
    {
        VMV_MUTABLE(env)
        if (reason) *reason = 0;
        if (!pts || !ws_min || !ws_max) return VMV_ERR_INVALID_ARGUMENT;
        if (env->mvts.size() >= (size_t) vmv::kMaxMvt) return VMV_ERR_CAPACITY;
        const auto t0 = std::chrono::steady_clock::now();
        vmv::MvtArrays arrays;
        const vmv::MvtStatus st = vmv::build_mvt(pts, n, r_min, r_max, ws_min, ws_max, r_point, arrays);
        if (st != vmv::MvtStatus::ok)
        {
            if (reason) *reason = (int) st;
            g_last_error = "MVT: a pool the reference sizes up front would be exhausted (see vmv_mvt_build.h)";
            return VMV_ERR_CAPACITY;
        }
        env->mvts.push_back(std::move(arrays));
        if (build_ns)
            *build_ns = (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(
                            std::chrono::steady_clock::now() - t0)
                            .count();
        return VMV_OK;
    }

    int vmv_env_mvt_count(const vmv_env *env, size_t *count)
    {
        if (!env || !count) return VMV_ERR_INVALID_ARGUMENT;
        *count = env->mvts.size();
        return VMV_OK;
    }
    int vmv_env_mvt_info(const vmv_env *env, size_t index, uint32_t *gw, uint32_t *cap, uint32_t *nv, float *isf, float *box)
    {
        if (!env || index >= env->mvts.size()) return VMV_ERR_INVALID_ARGUMENT;
        const vmv::MvtArrays &m = env->mvts[index];
        if (gw) *gw = m.grid_width;
        if (cap) *cap = m.capacity;
        if (nv) *nv = m.n_voxels();
        if (isf) *isf = m.inv_scale;
        if (box)
        {
            std::memcpy(box, m.gmin, 12);
            std::memcpy(box + 3, m.gmax, 12);
        }
        return VMV_OK;
    }

    int vmv_env_finalize(vmv_env *env)
    {
        VMV_MUTABLE(env)
        int rc = require_device();
        if (rc != VMV_OK) return rc;
        sort_by_min_distance(env->spheres);
        sort_by_min_distance(env->capsules);
        sort_by_min_distance(env->z_capsules);
        sort_by_min_distance(env->cuboids);
        sort_by_min_distance(env->z_cuboids);

        // pack the LDS primitive block (record layouts: vmv_device.h)
        std::vector<float> block;
        vmv::EnvDev D_base{};
        vmv::EnvDev &D = D_base;
        D.n_sphere = (uint32_t) env->spheres.size();
        D.off_sphere = (uint32_t) block.size();
        for (const auto &s : env->spheres)
        {
            block.insert(block.end(), {s.x, s.y, s.z, s.r, s.min_d});
            block.insert(block.end(), (size_t) vmv::kSphereRec - 5, 0.f);
        }
        D.n_capsule = (uint32_t) env->capsules.size();
        D.off_capsule = (uint32_t) block.size();
        for (const auto &c : env->capsules)
            block.insert(block.end(),
                         {c.p[0], c.p[1], c.p[2], c.p[3], c.p[4], c.p[5], c.p[6], c.p[7], c.min_d, 0.f, 0.f, 0.f});
        D.n_zcapsule = (uint32_t) env->z_capsules.size();
        D.off_zcapsule = (uint32_t) block.size();
        for (const auto &c : env->z_capsules)
            block.insert(block.end(), {c.p[0], c.p[1], c.p[2], c.p[5], c.p[6], c.p[7], c.min_d, 0.f});
        D.n_cuboid = (uint32_t) env->cuboids.size();
        D.off_cuboid = (uint32_t) block.size();
        for (const auto &c : env->cuboids)
        {
            block.insert(block.end(), c.p, c.p + 15);
            block.push_back(c.min_d);
        }
        D.n_zcuboid = (uint32_t) env->z_cuboids.size();
        D.off_zcuboid = (uint32_t) block.size();
        for (const auto &c : env->z_cuboids)
            block.insert(block.end(), {c.p[0], c.p[1], c.p[2], c.p[3], c.p[4], c.p[6], c.p[7], c.p[12], c.p[13],
                                       c.p[14], c.min_d, 0.f});
        // compact min_distance arrays (one per list) for the wave-wide live-prefix count
        auto md_array = [&](auto &list, uint32_t &off)
        {
            while (block.size() % 4) block.push_back(0.f);
            off = (uint32_t) block.size();
            for (const auto &s : list) block.push_back(s.min_d);
        };
        md_array(env->spheres, D.off_md_sphere);
        md_array(env->capsules, D.off_md_capsule);
        md_array(env->z_capsules, D.off_md_zcapsule);
        md_array(env->cuboids, D.off_md_cuboid);
        md_array(env->z_cuboids, D.off_md_zcuboid);
        uint32_t total_words = 0;
        {  // candidate words of the fine phase (vmv_device.h, kCandidateMargin): 32 primitives per word, per list
            uint32_t *const base[5] = {&D.wbase_sphere, &D.wbase_capsule, &D.wbase_zcapsule, &D.wbase_cuboid, &D.wbase_zcuboid};
            uint32_t *const shift[5] = {&D.wshift_sphere, &D.wshift_capsule, &D.wshift_zcapsule, &D.wshift_cuboid, &D.wshift_zcuboid};
            const uint32_t count[5] = {D.n_sphere, D.n_capsule, D.n_zcapsule, D.n_cuboid, D.n_zcuboid};
            uint32_t w = 0;
            for (int t = 0; t < 5; ++t)
            {
                *base[t] = w, *shift[t] = 0u;
                w += (count[t] + 31u) / 32u;
            }
            // More lists than words (all five kinds present, say): lists of at most 32 primitives share words — a list
            // starts at the next free bit when it fits into what is left of the word, else (and always when it is
            // longer than a word) at bit 0 of the next one.  Not for the environments the three-list kernel variant
            // serves (well-formed primitives without general cuboids / capsules, nothing else): it reads no shifts.
            const bool three_lists = env->capsules.empty() && env->cuboids.empty() && env->capts.empty() && env->mvts.empty() &&
                                     env->heightfields.empty();
            if (w > (uint32_t) vmv::kMaskWords && !three_lists)
            {
                uint32_t pos = 0;  // next free bit
                for (int t = 0; t < 5; ++t)
                {
                    if (count[t] == 0u) continue;
                    if (count[t] > 32u || (pos % 32u) + count[t] > 32u) pos = (pos + 31u) & ~31u;
                    *base[t] = pos / 32u, *shift[t] = pos % 32u;
                    pos += count[t];
                    if (count[t] > 32u) pos = (pos + 31u) & ~31u;
                }
                w = (pos + 31u) / 32u;
            }
            D.masked_fine = (w <= (uint32_t) vmv::kMaskWords) ? 1u : 0u;
            total_words = w;
        }
        {
            // The candidate pruning (fine-phase margin, broad-phase grid) assumes 1-Lipschitz primitive distances, which
            // holds for orthonormal cuboid axes and a capsule whose rdv is 1 / |v|^2.  The ABI accepts arbitrary
            // canonical parameters (as the reference does), so environments with an ill-formed primitive run the full
            // sorted loops instead: same answers as the reference's loop, no pruning.
            bool well_formed = true;
            const double origin[3] = {0, 0, 0};
            for (const auto *list : {&env->cuboids, &env->z_cuboids})
                for (const auto &c : *list)
                {
                    double lip = 1.0;
                    (void) vmv::grid_detail::cuboid_g(c.p, origin, list == &env->z_cuboids, lip);
                    well_formed = well_formed && std::isfinite(lip) && lip <= 1.0 + 1e-4;
                }
            for (const auto *list : {&env->capsules, &env->z_capsules})
                for (const auto &c : *list)
                {
                    const bool z = list == &env->z_capsules;
                    const double vv = (z ? 0.0 : (double) c.p[3] * c.p[3] + (double) c.p[4] * c.p[4]) + (double) c.p[5] * c.p[5];
                    const double k = vv * (double) c.p[7];
                    well_formed = well_formed && std::isfinite(k) && std::fabs(k - 1.0) <= 1e-3;
                }
            D.ill_formed = well_formed ? 0u : 1u;
            if (!well_formed) D.masked_fine = 0u;
        }
        while (block.size() % 4) block.push_back(0.f);
        if (block.size() > kMaxPrimFloats)
        {
            g_last_error = "more primitive records than the LDS staging budget (48 KiB)";
            return VMV_ERR_CAPACITY;
        }
        D.n_floats = (uint32_t) block.size();
        VMV_HIP(hipGetDevice(&env->device));
        rc = upload(env, block, &D.prims);
        if (rc != VMV_OK) return rc;

        D.n_capt = (uint32_t) env->capts.size();
        // the query copy of a cloud (blocked planes, leaf records, points sorted by distance to the leaf cell)
        auto query_copy = [&](const vmv::CaptArrays &a, vmv::CaptDev &c, uint32_t n_vectors) -> int
        {
            bool prune = true;
            if (const char *e = std::getenv("VMV_CAPT_NO_PREFIX")) prune = e[0] != '1';  // measurement / test aid
            for (int k = 0; k < 6; ++k)
                if (!(std::fabs(a.aabb_top[k]) <= 1e2f)) prune = false;  // the 1e-4 m margin is 13 fp32 ulps at 100 m; beyond: walk everything
            vmv::CaptQueryDev q;
            const int rc = vmv::build_capt_query(c.tests, c.aff_starts, c.aabbs, c.aff_x, c.aff_y, c.aff_z, a.nlog2, n_vectors,
                                                 a.r_min, a.r_max, a.r_point, prune, a.aabb_top, q);
            if (rc != VMV_OK)
            {
                if (rc == VMV_ERR_CAPACITY) g_last_error = "point cloud too large for the device query copy";
                return rc;
            }
            env->allocations.push_back(q.points);
            env->allocations.push_back(q.leaves);
            env->allocations.push_back(q.planes);
            c.q_x = q.points, c.q_y = q.points + (size_t) n_vectors * 8, c.q_z = q.points + 2 * (size_t) n_vectors * 8;
            c.q_leaves = q.leaves;
            c.q_planes = q.planes;
            c.cut_t0 = q.t0, c.cut_inv_step = q.inv_step;
            c.q_dist = q.dist;
            if (q.dist) env->allocations.push_back(q.dist);
            for (int k = 0; k < 3; ++k) c.dist_dims[k] = q.dist_dims[k], c.dist_origin[k] = q.dist_origin[k];
            c.dist_inv_cell = q.dist_inv_cell;
            return VMV_OK;
        };
        for (size_t i = 0; i < env->capts.size(); ++i)
        {
            vmv::CaptArrays &a = env->capts[i];
            vmv::CaptDev &c = D.capt[i];
            if (a.dev.tests && a.dev.device == env->device)
            {
                // built on this device: use the arrays where they are
                const size_t nv = a.dev.n_vectors;
                c.tests = a.dev.tests;
                c.aff_starts = a.dev.aff_starts;
                c.aabbs = a.dev.aabbs;
                c.aff_x = a.dev.aff;
                c.aff_y = a.dev.aff + nv * 8;
                c.aff_z = a.dev.aff + 2 * nv * 8;
                std::memcpy(c.aabb_top, a.aabb_top, sizeof(c.aabb_top));
                c.r_point = a.r_point;
                c.nlog2 = a.nlog2;
                c.n_tests = a.n_tests();
                if ((rc = query_copy(a, c, (uint32_t) nv)) != VMV_OK) return rc;
                continue;
            }
            if ((rc = vmv::download_capt(a)) != VMV_OK) return rc;  // built on another device: go through the host
            if ((rc = upload(env, a.tests, &c.tests)) != VMV_OK) return rc;
            if ((rc = upload(env, a.aff_starts, &c.aff_starts)) != VMV_OK) return rc;
            if ((rc = upload(env, a.aabbs, &c.aabbs)) != VMV_OK) return rc;
            if ((rc = upload(env, a.aff[0], &c.aff_x)) != VMV_OK) return rc;
            if ((rc = upload(env, a.aff[1], &c.aff_y)) != VMV_OK) return rc;
            if ((rc = upload(env, a.aff[2], &c.aff_z)) != VMV_OK) return rc;
            std::memcpy(c.aabb_top, a.aabb_top, sizeof(c.aabb_top));
            c.r_point = a.r_point;
            c.nlog2 = a.nlog2;
            c.n_tests = (uint32_t) a.tests.size();
            if ((rc = query_copy(a, c, (uint32_t) (a.aff[0].size() / 8))) != VMV_OK) return rc;
        }
        D.capt0_n_tests = D.n_capt ? env->capts[0].n_tests() : 0u;
        D.n_mvt = (uint32_t) env->mvts.size();
        for (size_t i = 0; i < env->mvts.size(); ++i)
        {
            const vmv::MvtArrays &m = env->mvts[i];
            vmv::MvtDev &d = D.mvt[i];
            if ((rc = upload(env, m.cells, &d.cells)) != VMV_OK) return rc;
            if ((rc = upload(env, m.vox_bbox, &d.vox_bbox)) != VMV_OK) return rc;
            if ((rc = upload(env, m.vox_offset, &d.vox_offset)) != VMV_OK) return rc;
            if ((rc = upload(env, m.px, &d.px)) != VMV_OK) return rc;
            if ((rc = upload(env, m.py, &d.py)) != VMV_OK) return rc;
            if ((rc = upload(env, m.pz, &d.pz)) != VMV_OK) return rc;
            std::memcpy(d.gmin, m.gmin, 12);
            std::memcpy(d.gmax, m.gmax, 12);
            std::memcpy(d.ws_min, m.ws_min, 12);
            d.inv_scale = m.inv_scale;
            d.r_point = m.r_point;
            d.grid_width = m.grid_width;
        }
        D.n_attach = 0;
        if (env->attached && !env->attach_spheres.empty())
        {
            // (an attachment without spheres poses nothing and tests nothing: plain fkcc gives the same answers)
            D.n_attach = (uint32_t) (env->attach_spheres.size() / 4);
            std::memcpy(D.attach_tf, env->attach_tf, sizeof(D.attach_tf));
            if ((rc = upload(env, env->attach_spheres, &D.attach_spheres)) != VMV_OK) return rc;
        }
        D.n_heightfield = (uint32_t) env->heightfields.size();
        for (size_t i = 0; i < env->heightfields.size(); ++i)
        {
            const vmv_env::HeightField &h = env->heightfields[i];
            vmv::HeightFieldDev &d = D.heightfield[i];
            if ((rc = upload(env, h.data, &d.data)) != VMV_OK) return rc;
            d.x = h.c[0], d.y = h.c[1], d.z = h.c[2];
            d.xs = h.inv[0], d.ys = h.inv[1], d.zs = h.inv[2];
            d.xd = (float) h.xd, d.yd = (float) h.yd;
            d.xd2 = (float) (h.xd / 2), d.yd2 = (float) (h.yd / 2);  // shapes.hh:289-290
            d.last = (uint32_t) (h.xd * h.yd - 1);
        }
        // broad-phase grid of the gate pass, one per robot (its reach is the robot's largest bounding radius)
        std::vector<vmv::GridPrim> gp;
        if (D.masked_fine)
        {
            auto add = [&](int type, const float *p, uint32_t wbase, uint32_t shift, size_t i)
            { gp.push_back(vmv::GridPrim{type, p, wbase + (uint32_t) ((shift + i) / 32), (uint32_t) ((shift + i) % 32)}); };
            for (size_t i = 0; i < env->spheres.size(); ++i) add(0, &env->spheres[i].x, D.wbase_sphere, D.wshift_sphere, i);
            for (size_t i = 0; i < env->capsules.size(); ++i) add(1, env->capsules[i].p, D.wbase_capsule, D.wshift_capsule, i);
            for (size_t i = 0; i < env->z_capsules.size(); ++i) add(2, env->z_capsules[i].p, D.wbase_zcapsule, D.wshift_zcapsule, i);
            for (size_t i = 0; i < env->cuboids.size(); ++i) add(3, env->cuboids[i].p, D.wbase_cuboid, D.wshift_cuboid, i);
            for (size_t i = 0; i < env->z_cuboids.size(); ++i) add(4, env->z_cuboids[i].p, D.wbase_zcuboid, D.wshift_zcuboid, i);
        }
        env->base = D_base;
        env->grid_prims = std::move(gp);
        env->grid_words = total_words;
        env->finalized = true;
        return VMV_OK;
    }

    int vmv_env_counts(const vmv_env *env, size_t *c)
    {
        if (!env || !c) return VMV_ERR_INVALID_ARGUMENT;
        c[0] = env->spheres.size();
        c[1] = env->capsules.size();
        c[2] = env->z_capsules.size();
        c[3] = env->cuboids.size();
        c[4] = env->z_cuboids.size();
        c[5] = env->capts.size();
        return VMV_OK;
    }
    int vmv_env_get_spheres(const vmv_env *env, float *out, size_t cap, size_t *n)
    {
        if (!env || !n) return VMV_ERR_INVALID_ARGUMENT;
        const auto v = sorted_copy(env->spheres);
        *n = v.size();
        if (out)
            for (size_t i = 0; i < std::min(cap, *n); ++i) std::memcpy(out + 5 * i, &v[i], 5 * sizeof(float));
        return VMV_OK;
    }
    int vmv_env_get_cuboids(const vmv_env *env, int z, float *out, size_t cap, size_t *n)
    {
        if (!env || !n) return VMV_ERR_INVALID_ARGUMENT;
        const auto v = sorted_copy(z ? env->z_cuboids : env->cuboids);
        *n = v.size();
        if (out)
            for (size_t i = 0; i < std::min(cap, *n); ++i) std::memcpy(out + 16 * i, &v[i], 16 * sizeof(float));
        return VMV_OK;
    }
    int vmv_env_get_capsules(const vmv_env *env, int z, float *out, size_t cap, size_t *n)
    {
        if (!env || !n) return VMV_ERR_INVALID_ARGUMENT;
        const auto v = sorted_copy(z ? env->z_capsules : env->capsules);
        *n = v.size();
        if (out)
            for (size_t i = 0; i < std::min(cap, *n); ++i) std::memcpy(out + 9 * i, &v[i], 9 * sizeof(float));
        return VMV_OK;
    }
    int vmv_env_capt_sizes(const vmv_env *env, size_t index, uint32_t *nlog2, uint32_t *n_aff)
    {
        if (!env || index >= env->capts.size()) return VMV_ERR_INVALID_ARGUMENT;
        if (nlog2) *nlog2 = env->capts[index].nlog2;
        if (n_aff) *n_aff = env->capts[index].n_aff_vectors();
        return VMV_OK;
    }
    int vmv_env_capt_arrays(const vmv_env *env, size_t index, float *tests, uint32_t *aff_starts, float *aabbs,
                            float *ax, float *ay, float *az, float *top)
    {
        if (!env || index >= env->capts.size()) return VMV_ERR_INVALID_ARGUMENT;
        vmv::CaptArrays &a = const_cast<vmv_env *>(env)->capts[index];  // a GPU-built cloud is downloaded on first look
        const int rc = vmv::download_capt(a);
        if (rc != VMV_OK) return rc;
        if (tests) std::memcpy(tests, a.tests.data(), a.tests.size() * 4);
        if (aff_starts) std::memcpy(aff_starts, a.aff_starts.data(), a.aff_starts.size() * 4);
        if (aabbs) std::memcpy(aabbs, a.aabbs.data(), a.aabbs.size() * 4);
        if (ax) std::memcpy(ax, a.aff[0].data(), a.aff[0].size() * 4);
        if (ay) std::memcpy(ay, a.aff[1].data(), a.aff[1].size() * 4);
        if (az) std::memcpy(az, a.aff[2].data(), a.aff[2].size() * 4);
        if (top) std::memcpy(top, a.aabb_top, sizeof(a.aabb_top));
        return VMV_OK;
    }
}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// robot-specific part of a finalized environment, built on first use (thread safe; environments stay immutable
// from the caller's point of view)
// ---------------------------------------------------------------------------------------------------------
namespace
{
    int build_robot_part(vmv_env *env, int r)
    {
        int rc;
        int dev = -1;
        VMV_HIP(hipGetDevice(&dev));
        if (dev != env->device) VMV_HIP(hipSetDevice(env->device));
        vmv::EnvDev Dr = env->base;
        const bool use_grid = std::getenv("VMV_NO_GRID") == nullptr;
        if (use_grid && !env->grid_prims.empty())
        {
            bool ok = true;
            std::vector<vmv::GridArrays> grids(vmv::kGridClasses);
            for (int c = 0; c < vmv::kGridClasses && ok; ++c)
            {
                if (c > 0 && kRobots[r].grid_radius[c] == kRobots[r].grid_radius[c - 1])
                    grids[c] = grids[c - 1];
                else
                    ok = vmv::build_grid(env->grid_prims, env->grid_words, (double) kRobots[r].grid_radius[c], grids[c]);
            }
            if (ok)
            {
                for (int c = 0; c < vmv::kGridClasses; ++c)
                {
                    vmv::GridDev &g = Dr.grid[c];
                    if ((rc = upload(env, grids[c].cells, &g.cells)) != VMV_OK) return rc;
                    for (int k = 0; k < 3; ++k)
                    {
                        g.dims[k] = grids[c].dims[k];
                        g.origin[k] = grids[c].origin[k];
                    }
                    g.inv_cell = grids[c].inv_cell;
                }
                Dr.grid_words = env->grid_words;
            }
        }
        // Reach certificates: a link whose bounding sphere cannot come within 1 mm of any primitive, whatever the
        // configuration, is skipped by the environment kernels (its gate could never fire).  Only for environments of
        // well-formed primitives (masked_fine: the distance expressions below are then exact and 1-Lipschitz) without
        // heightfields / point clouds; VMV_NO_LINK_SKIP=1 switches it off (A/B, tests).
        Dr.link_skip = 0ull;
        if (Dr.masked_fine && Dr.n_capt + Dr.n_mvt + Dr.n_heightfield == 0 && std::getenv("VMV_NO_LINK_SKIP") == nullptr)
            for (int k = 0; k < kRobots[r].n_reach; ++k)
            {
                const vmv_link_reach &lr = kRobots[r].reach[k];
                if (lr.n <= 0 || lr.group < 0 || lr.group >= 64) continue;
                const double need = (double) lr.radius + (double) lr.slack + 1e-3;
                bool free = true;
                for (int i = 0; i < lr.n && free; ++i)
                {
                    const double x[3] = {lr.samples[i][0], lr.samples[i][1], lr.samples[i][2]};
                    for (const auto &g : env->grid_prims)
                    {
                        double lip = 1.0, d;
                        if (g.type == 0)
                        {
                            const double dx = x[0] - g.p[0], dy = x[1] - g.p[1], dz = x[2] - g.p[2];
                            d = std::sqrt(dx * dx + dy * dy + dz * dz) - g.p[3];
                        }
                        else if (g.type <= 2)
                            d = vmv::grid_detail::capsule_g(g.p, x, g.type == 2, lip);
                        else
                            d = vmv::grid_detail::cuboid_g(g.p, x, g.type == 4, lip);
                        if (!(d > need * lip))
                        {
                            free = false;
                            break;
                        }
                    }
                }
                if (free) Dr.link_skip |= 1ull << lr.group;
            }
        env->launch[r].host = Dr;
        std::vector<vmv::EnvDev> one(1, Dr);
        const vmv::EnvDev *d_env = nullptr;
        if ((rc = upload(env, one, &d_env)) != VMV_OK) return rc;
        env->launch[r].d_env = d_env;
        rc = kLaunchers[r]->prepare(env->launch[r], const_cast<vmv::EnvDev *>(d_env));
        if (dev != env->device) (void) hipSetDevice(dev);
        return rc;
    }

    // A finalized environment lives on ONE device (its primitive block, point clouds, grids).  The batched entry
    // points launch on the caller's current device and stream, so that device must be the environment's.
    int check_device(const vmv_env *env)
    {
        int dev = -1;
        VMV_HIP(hipGetDevice(&dev));
        if (dev != env->device)
        {
            g_last_error = "environment was finalized on device " + std::to_string(env->device) + ", current device is " +
                           std::to_string(dev) + " (finalize one environment per device)";
            return VMV_ERR_INVALID_ARGUMENT;
        }
        return VMV_OK;
    }

    int ensure_robot(const vmv_env *cenv, int r)
    {
        vmv_env *env = const_cast<vmv_env *>(cenv);
        std::call_once(env->robot_once[r],
                       [&]()
                       {
                           std::lock_guard<std::mutex> lock(env->robot_mutex);
                           env->robot_status[r] = build_robot_part(env, r);
                           if (env->robot_status[r] != VMV_OK) env->robot_error[r] = g_last_error;
                       });
        if (env->robot_status[r] != VMV_OK) g_last_error = env->robot_error[r];
        return env->robot_status[r];
    }
}  // namespace

// ---------------------------------------------------------------------------------------------------------
// batched entry points
// ---------------------------------------------------------------------------------------------------------
extern "C"
{
    int vmv_validate_batch(int robot, const vmv_env *env, const float *d_q, size_t n, uint64_t *d_bits, void *stream)
    {
        if (!robot_ok(robot)) return VMV_ERR_UNKNOWN_ROBOT;
        if (!env || !d_q || !d_bits) return VMV_ERR_INVALID_ARGUMENT;
        if (!env->finalized) return VMV_ERR_NOT_FINALIZED;
        if (n == 0) return VMV_OK;
        if (int rc = check_device(env); rc != VMV_OK) return rc;
        if (int rc = ensure_robot(env, robot); rc != VMV_OK) return rc;
        return kLaunchers[robot]->validate(env->launch[robot], d_q, n, d_bits, static_cast<hipStream_t>(stream), 7);
    }

    int vmv_validate_batch_env(int robot, const vmv_env *env, const float *d_q, size_t n, uint64_t *d_bits, void *stream)
    {
        if (!robot_ok(robot)) return VMV_ERR_UNKNOWN_ROBOT;
        if (!env || !d_q || !d_bits) return VMV_ERR_INVALID_ARGUMENT;
        if (!env->finalized) return VMV_ERR_NOT_FINALIZED;
        if (n == 0) return VMV_OK;
        if (int rc = check_device(env); rc != VMV_OK) return rc;
        if (int rc = ensure_robot(env, robot); rc != VMV_OK) return rc;
        return kLaunchers[robot]->validate(env->launch[robot], d_q, n, d_bits, static_cast<hipStream_t>(stream), 1);
    }

    int vmv_validate_batch_self(int robot, const float *d_q, size_t n, uint64_t *d_bits, void *stream)
    {
        if (!robot_ok(robot)) return VMV_ERR_UNKNOWN_ROBOT;
        if (!d_q || !d_bits) return VMV_ERR_INVALID_ARGUMENT;
        if (n == 0) return VMV_OK;
        return kLaunchers[robot]->validate(vmv::EnvLaunch{}, d_q, n, d_bits, static_cast<hipStream_t>(stream), 2);
    }

    int vmv_validate_motion_batch(int robot, const vmv_env *env, const float *d_a, const float *d_b, size_t n,
                                  uint64_t *d_bits, void *stream)
    {
        if (!robot_ok(robot)) return VMV_ERR_UNKNOWN_ROBOT;
        if (!env || !d_a || !d_b || !d_bits) return VMV_ERR_INVALID_ARGUMENT;
        if (!env->finalized) return VMV_ERR_NOT_FINALIZED;
        if (n == 0) return VMV_OK;
        if (int rc = check_device(env); rc != VMV_OK) return rc;
        if (int rc = ensure_robot(env, robot); rc != VMV_OK) return rc;
        return kLaunchers[robot]->validate_motion(env->launch[robot], d_a, d_b, n, d_bits, static_cast<hipStream_t>(stream));
    }

    int vmv_fk_batch(int robot, const float *d_q, size_t n, float *d_out, void *stream)
    {
        if (!robot_ok(robot)) return VMV_ERR_UNKNOWN_ROBOT;
        if (!d_q || !d_out) return VMV_ERR_INVALID_ARGUMENT;
        if (n == 0) return VMV_OK;
        return kLaunchers[robot]->fk(d_q, n, d_out, static_cast<hipStream_t>(stream));
    }

    int vmv_robot_self_pairs(int robot, size_t *n_pairs, uint16_t *pairs2)
    {
        if (!robot_ok(robot)) return VMV_ERR_UNKNOWN_ROBOT;
        if (!n_pairs) return VMV_ERR_INVALID_ARGUMENT;
        *n_pairs = (size_t) kRobots[robot].n_self_pairs;
        if (pairs2) std::memcpy(pairs2, kRobots[robot].self_pairs, *n_pairs * 4);
        return VMV_OK;
    }
    int vmv_env_report_layout(const vmv_env *env, uint32_t *w)
    {
        if (!env || !w) return VMV_ERR_INVALID_ARGUMENT;
        if (!env->finalized) return VMV_ERR_NOT_FINALIZED;
        const vmv::EnvDev &D = env->base;
        uint32_t base = 0;
        const uint32_t counts[5] = {D.n_sphere, D.n_capsule, D.n_zcapsule, D.n_cuboid, D.n_zcuboid};
        for (int i = 0; i < 5; ++i)
        {
            w[i] = base;
            base += (counts[i] + 31u) / 32u;
        }
        return VMV_OK;
    }
    int vmv_contacts_batch_host(int robot, const vmv_env *env, const float *q, size_t n, uint32_t *env_words,
                                uint32_t *pair_words)
    {
        if (!robot_ok(robot)) return VMV_ERR_UNKNOWN_ROBOT;
        if (!env || !q || !env_words || !pair_words) return VMV_ERR_INVALID_ARGUMENT;
        if (!env->finalized) return VMV_ERR_NOT_FINALIZED;
        if (n == 0) return VMV_OK;
        int rc = require_device();
        if (rc != VMV_OK) return rc;
        if ((rc = check_device(env)) != VMV_OK) return rc;
        if ((rc = ensure_robot(env, robot)) != VMV_OK) return rc;
        const size_t ns = (size_t) kRobots[robot].n_spheres, npw = ((size_t) kLaunchers[robot]->n_self_pairs + 31) / 32;
        const size_t qb = n * (size_t) kRobots[robot].dimension * 4, sb = n * ns * 16, eb = n * ns * (vmv::kReportWords + 1) * 4,
                     pb = std::max<size_t>(n * npw * 4, 4);
        char *d = nullptr;
        VMV_HIP(hipMalloc((void **) &d, qb + sb + eb + pb + 1024));
        float *dq = reinterpret_cast<float *>(d), *ds = reinterpret_cast<float *>(d + ((qb + 255) & ~size_t{255}));
        uint32_t *de = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(ds) + ((sb + 255) & ~size_t{255}));
        uint32_t *dp = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(de) + ((eb + 255) & ~size_t{255}));
        rc = VMV_OK;
        if (hipMemcpy(dq, q, qb, hipMemcpyHostToDevice) != hipSuccess || hipMemset(dp, 0, pb) != hipSuccess) rc = VMV_ERR_HIP;
        if (rc == VMV_OK) rc = kLaunchers[robot]->fk(dq, n, ds, nullptr);
        if (rc == VMV_OK) rc = kLaunchers[robot]->contacts(env->launch[robot], ds, n, de, dp, nullptr);
        if (rc == VMV_OK && (hipMemcpy(env_words, de, eb, hipMemcpyDeviceToHost) != hipSuccess ||
                             (npw && hipMemcpy(pair_words, dp, n * npw * 4, hipMemcpyDeviceToHost) != hipSuccess)))
            rc = VMV_ERR_HIP;
        (void) hipFree(d);
        return rc;
    }

    int vmv_eefk_batch(int robot, const float *d_q, size_t n, float *d_out, void *stream)
    {
        if (!robot_ok(robot)) return VMV_ERR_UNKNOWN_ROBOT;
        if (!d_q || !d_out) return VMV_ERR_INVALID_ARGUMENT;
        if (n == 0) return VMV_OK;
        return kLaunchers[robot]->eefk(d_q, n, d_out, static_cast<hipStream_t>(stream));
    }
    int vmv_eefk_batch_host(int robot, const float *q, size_t n, float *out)
    {
        if (!robot_ok(robot)) return VMV_ERR_UNKNOWN_ROBOT;
        if (!q || !out) return VMV_ERR_INVALID_ARGUMENT;
        if (n == 0) return VMV_OK;
        int rc = require_device();
        if (rc != VMV_OK) return rc;
        const size_t qb = n * (size_t) kRobots[robot].dimension * 4, ob = n * 64;
        float *dq = nullptr, *dout = nullptr;
        VMV_HIP(hipMalloc((void **) &dq, qb));
        if (hipMalloc((void **) &dout, ob) != hipSuccess)
        {
            (void) hipFree(dq);
            return VMV_ERR_HIP;
        }
        rc = VMV_OK;
        if (hipMemcpy(dq, q, qb, hipMemcpyHostToDevice) != hipSuccess) rc = VMV_ERR_HIP;
        if (rc == VMV_OK) rc = vmv_eefk_batch(robot, dq, n, dout, nullptr);
        if (rc == VMV_OK && hipMemcpy(out, dout, ob, hipMemcpyDeviceToHost) != hipSuccess) rc = VMV_ERR_HIP;
        (void) hipFree(dq);
        (void) hipFree(dout);
        return rc;
    }

    // ---- host-buffer variants ----
    int vmv_fk_batch_host(int robot, const float *q, size_t n, float *out)
    {
        if (!robot_ok(robot)) return VMV_ERR_UNKNOWN_ROBOT;
        if (!q || !out) return VMV_ERR_INVALID_ARGUMENT;
        if (n == 0) return VMV_OK;
        int rc = require_device();
        if (rc != VMV_OK) return rc;
        const size_t qb = n * (size_t) kRobots[robot].dimension * 4, ob = n * (size_t) kRobots[robot].n_spheres * 16;
        float *dq = nullptr, *dout = nullptr;
        VMV_HIP(hipMalloc((void **) &dq, qb));
        if (hipMalloc((void **) &dout, ob) != hipSuccess)
        {
            (void) hipFree(dq);
            return VMV_ERR_HIP;
        }
        rc = VMV_OK;
        if (hipMemcpy(dq, q, qb, hipMemcpyHostToDevice) != hipSuccess) rc = VMV_ERR_HIP;
        if (rc == VMV_OK) rc = vmv_fk_batch(robot, dq, n, dout, nullptr);
        if (rc == VMV_OK && hipMemcpy(out, dout, ob, hipMemcpyDeviceToHost) != hipSuccess) rc = VMV_ERR_HIP;
        (void) hipFree(dq);
        (void) hipFree(dout);
        return rc;
    }

    // Device staging of the host-buffer entry points: one grow-only arena per (thread, device), so that a planner's many
    // small calls (one configuration, a handful of edges) do not pay three hipMalloc / hipFree pairs each.
    namespace
    {
        // Device staging of the *_host entry points, per thread.  Never freed from a destructor: a thread_local of the
        // main thread is destroyed during static destruction, possibly after the HIP runtime is gone (the memory goes
        // back with the process).  Long-lived callers release it with vmv_release_staging(); a request above
        // kStagingKeepBytes is not kept beyond its call, so one huge batch does not pin its memory for the thread's life.
        constexpr size_t kStagingKeepBytes = size_t{64} << 20;
        struct StagingArena
        {
            void *base = nullptr;
            size_t capacity = 0;
            int device = -1;
            void release()
            {
                if (base) (void) hipFree(base);
                base = nullptr;
                capacity = 0;
            }
            void *get(size_t bytes)
            {
                int dev = -1;
                if (hipGetDevice(&dev) != hipSuccess) return nullptr;
                if (base && (dev != device || bytes > capacity)) release();
                if (!base)
                {
                    const size_t want = std::max<size_t>(bytes, 1u << 16);
                    if (hipMalloc(&base, want) != hipSuccess)
                    {
                        base = nullptr;
                        return nullptr;
                    }
                    capacity = want;
                    device = dev;
                }
                return base;
            }
            void trim()
            {
                if (capacity > kStagingKeepBytes) release();
            }
        };
        thread_local StagingArena g_staging;
    }  // namespace

    static int validate_host_common(int robot, const vmv_env *env, const float *a, const float *b, size_t n,
                                    uint64_t *bits)
    {
        if (!robot_ok(robot)) return VMV_ERR_UNKNOWN_ROBOT;
        if (!a || !bits) return VMV_ERR_INVALID_ARGUMENT;
        if (n == 0) return VMV_OK;
        int rc = require_device();
        if (rc != VMV_OK) return rc;
        const size_t qb = (n * (size_t) kRobots[robot].dimension * 4 + 255) & ~size_t{255}, wb = ((n + 63) / 64) * 8;
        char *arena = static_cast<char *>(g_staging.get(2 * qb + wb));
        if (!arena) return hip_fail(hipErrorOutOfMemory, "staging arena");
        float *da = reinterpret_cast<float *>(arena), *db = reinterpret_cast<float *>(arena + qb);
        uint64_t *dbits = reinterpret_cast<uint64_t *>(arena + 2 * qb);
        const size_t bytes = n * (size_t) kRobots[robot].dimension * 4;
        if (hipMemcpy(da, a, bytes, hipMemcpyHostToDevice) != hipSuccess) return VMV_ERR_HIP;
        if (b && hipMemcpy(db, b, bytes, hipMemcpyHostToDevice) != hipSuccess) return VMV_ERR_HIP;
        rc = b ? vmv_validate_motion_batch(robot, env, da, db, n, dbits, nullptr) :
                 vmv_validate_batch(robot, env, da, n, dbits, nullptr);
        if (rc == VMV_OK && hipMemcpy(bits, dbits, wb, hipMemcpyDeviceToHost) != hipSuccess) rc = VMV_ERR_HIP;
        g_staging.trim();
        return rc;
    }
    int vmv_release_staging(void)
    {
        g_staging.release();
        vmv::release_edge_scratch();  // (waits for the edge batches still in flight: hipFree synchronizes)
        return VMV_OK;
    }
    int vmv_validate_batch_host(int robot, const vmv_env *env, const float *q, size_t n, uint64_t *bits)
    {
        return validate_host_common(robot, env, q, nullptr, n, bits);
    }
    int vmv_validate_motion_batch_host(int robot, const vmv_env *env, const float *a, const float *b, size_t n,
                                       uint64_t *bits)
    {
        if (!b) return VMV_ERR_INVALID_ARGUMENT;
        return validate_host_common(robot, env, a, b, n, bits);
    }

    // ---- multi-GPU sharding arithmetic (vamp_mvt_amd/sharding.py: shard_range) ----
    size_t vmv_shard_words(size_t n, int world)
    {
        if (world < 1) return 0;
        const size_t words = (n + 63) / 64;
        return (words + (size_t) world - 1) / (size_t) world;
    }
    int vmv_shard_range(size_t n, int rank, int world, size_t *lo, size_t *hi)
    {
        if (!lo || !hi || world < 1 || rank < 0 || rank >= world) return VMV_ERR_INVALID_ARGUMENT;
        const size_t per = vmv_shard_words(n, world) * 64;
        *lo = std::min((size_t) rank * per, n);
        *hi = std::min(((size_t) rank + 1) * per, n);
        return VMV_OK;
    }

    // ---- free spheres against the environment ----
    int vmv_spheres_in_collision_batch(const vmv_env *env, const float *d_spheres, size_t n, uint8_t *d_hits, void *stream)
    {
        if (!env || !d_spheres || !d_hits) return VMV_ERR_INVALID_ARGUMENT;
        if (!env->finalized) return VMV_ERR_NOT_FINALIZED;
        if (n == 0) return VMV_OK;
        if (int rc = check_device(env); rc != VMV_OK) return rc;
        if (int rc = ensure_robot(env, 0); rc != VMV_OK) return rc;  // any robot's image: the counted loops use no grid
        const size_t blocks = (n + vmv::kBlock - 1) / vmv::kBlock;
        hipLaunchKernelGGL(vmv::spheres_env_kernel, dim3((unsigned) std::min<size_t>(blocks, 4096)), dim3(vmv::kBlock),
                           (env->base.n_floats + vmv::kCaptFlagWords) * sizeof(float), static_cast<hipStream_t>(stream), env->launch[0].d_env,
                           reinterpret_cast<const float4 *>(d_spheres), n, d_hits, (const float4 *) nullptr, 0);
        VMV_HIP(hipGetLastError());
        return VMV_OK;
    }
    int vmv_filter_self_from_pointcloud(int robot, const vmv_env *env, const float *q, const float *points_xyz, size_t n,
                                        float point_radius, float *out_xyz, size_t capacity, size_t *n_out)
    {
        if (!robot_ok(robot)) return VMV_ERR_UNKNOWN_ROBOT;
        if (!env || !q || (n && !points_xyz) || !n_out) return VMV_ERR_INVALID_ARGUMENT;
        if (!env->finalized) return VMV_ERR_NOT_FINALIZED;
        *n_out = 0;
        if (n == 0) return VMV_OK;
        int rc = require_device();
        if (rc != VMV_OK) return rc;
        if ((rc = check_device(env)) != VMV_OK) return rc;
        if ((rc = ensure_robot(env, 0)) != VMV_OK) return rc;
        const size_t dim = (size_t) kRobots[robot].dimension, ns = (size_t) kRobots[robot].n_spheres;
        std::vector<float> spheres(4 * n);
        for (size_t i = 0; i < n; ++i)
        {
            std::memcpy(&spheres[4 * i], points_xyz + 3 * i, 12);
            spheres[4 * i + 3] = point_radius;
        }
        float *dq = nullptr, *drs = nullptr, *ds = nullptr;
        uint8_t *dh = nullptr;
        std::vector<uint8_t> hits(n);
        auto release = [&]()
        {
            (void) hipFree(dq);
            (void) hipFree(drs);
            (void) hipFree(ds);
            (void) hipFree(dh);
        };
        if (hipMalloc((void **) &dq, dim * 4) != hipSuccess || hipMalloc((void **) &drs, ns * 16) != hipSuccess ||
            hipMalloc((void **) &ds, n * 16) != hipSuccess || hipMalloc((void **) &dh, n) != hipSuccess ||
            hipMemcpy(dq, q, dim * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(ds, spheres.data(), n * 16, hipMemcpyHostToDevice) != hipSuccess)
        {
            release();
            return VMV_ERR_HIP;
        }
        rc = kLaunchers[robot]->fk(dq, 1, drs, nullptr);  // Robot::sphere_fk of the one configuration
        if (rc == VMV_OK)
        {
            const size_t blocks = (n + vmv::kBlock - 1) / vmv::kBlock;
            hipLaunchKernelGGL(vmv::spheres_env_kernel, dim3((unsigned) std::min<size_t>(blocks, 4096)), dim3(vmv::kBlock),
                               (env->base.n_floats + vmv::kCaptFlagWords) * sizeof(float), nullptr, env->launch[0].d_env,
                               reinterpret_cast<const float4 *>(ds), n, dh, reinterpret_cast<const float4 *>(drs), (int) ns);
            if (hipGetLastError() != hipSuccess || hipMemcpy(hits.data(), dh, n, hipMemcpyDeviceToHost) != hipSuccess)
                rc = VMV_ERR_HIP;
        }
        release();
        if (rc != VMV_OK) return rc;
        size_t m = 0;
        for (size_t i = 0; i < n; ++i)
            if (!hits[i])
            {
                if (out_xyz && m < capacity) std::memcpy(out_xyz + 3 * m, points_xyz + 3 * i, 12);
                ++m;
            }
        *n_out = m;
        return (out_xyz && m > capacity) ? VMV_ERR_CAPACITY : VMV_OK;
    }
    int vmv_spheres_in_collision_batch_host(const vmv_env *env, const float *spheres, size_t n, uint8_t *hits)
    {
        if (!env || !spheres || !hits) return VMV_ERR_INVALID_ARGUMENT;
        if (n == 0) return VMV_OK;
        int rc = require_device();
        if (rc != VMV_OK) return rc;
        float *ds = nullptr;
        uint8_t *dh = nullptr;
        VMV_HIP(hipMalloc((void **) &ds, n * 16));
        if (hipMalloc((void **) &dh, n) != hipSuccess)
        {
            (void) hipFree(ds);
            return VMV_ERR_HIP;
        }
        if (hipMemcpy(ds, spheres, n * 16, hipMemcpyHostToDevice) != hipSuccess) rc = VMV_ERR_HIP;
        if (rc == VMV_OK) rc = vmv_spheres_in_collision_batch(env, ds, n, dh, nullptr);
        if (rc == VMV_OK && hipMemcpy(hits, dh, n, hipMemcpyDeviceToHost) != hipSuccess) rc = VMV_ERR_HIP;
        (void) hipFree(ds);
        (void) hipFree(dh);
        return rc;
    }

    // ---- measurement support ----
    int vmv_time_validate_batch(int robot, const vmv_env *env, const float *d_q, size_t n, uint64_t *d_bits, int iters,
                                void *stream, float *avg_ms)
    {
        if (!avg_ms || iters < 1) return VMV_ERR_INVALID_ARGUMENT;
        hipStream_t s = static_cast<hipStream_t>(stream);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        VMV_HIP(hipEventCreate(&e0));
        hipError_t he = hipEventCreate(&e1);
        int rc = VMV_OK;
        float ms = 0.f;
        if (he == hipSuccess) he = hipEventRecord(e0, s);
        for (int i = 0; he == hipSuccess && i < iters && rc == VMV_OK; ++i)
            rc = vmv_validate_batch(robot, env, d_q, n, d_bits, stream);
        if (he == hipSuccess) he = hipEventRecord(e1, s);
        if (he == hipSuccess) he = hipEventSynchronize(e1);
        if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
        (void) hipEventDestroy(e0);  // on every exit path
        if (e1) (void) hipEventDestroy(e1);
        if (he != hipSuccess) return hip_fail(he, "vmv_time_validate_batch");
        *avg_ms = ms / (float) iters;
        return rc;
    }

    int vmv_fill_uniform_configs(int robot, float *d_q, size_t n, uint64_t seed, void *stream)
    {
        if (!robot_ok(robot)) return VMV_ERR_UNKNOWN_ROBOT;
        if (!d_q) return VMV_ERR_INVALID_ARGUMENT;
        const int dim = kRobots[robot].dimension;
        float *d_bounds = nullptr;
        VMV_HIP(hipMalloc((void **) &d_bounds, 32 * sizeof(float)));
        VMV_HIP(hipMemcpy(d_bounds, kRobots[robot].lower, 16 * sizeof(float), hipMemcpyHostToDevice));
        VMV_HIP(hipMemcpy(d_bounds + 16, kRobots[robot].span, 16 * sizeof(float), hipMemcpyHostToDevice));
        const size_t total = n * (size_t) dim;
        hipLaunchKernelGGL(vmv::fill_uniform_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), d_q, total, dim, seed, d_bounds, d_bounds + 16);
        VMV_HIP(hipGetLastError());
        VMV_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
        (void) hipFree(d_bounds);
        return VMV_OK;
    }

    int vmv_halton_configs(int robot, uint64_t skip, float *d_q, size_t n, void *stream)
    {
        if (!robot_ok(robot)) return VMV_ERR_UNKNOWN_ROBOT;
        if (!d_q || skip + n > 1000000ull) return VMV_ERR_INVALID_ARGUMENT;
        if (n == 0) return VMV_OK;
        const int dim = kRobots[robot].dimension;
        float *d_bounds = nullptr;
        VMV_HIP(hipMalloc((void **) &d_bounds, 32 * sizeof(float)));
        VMV_HIP(hipMemcpy(d_bounds, kRobots[robot].lower, 16 * sizeof(float), hipMemcpyHostToDevice));
        VMV_HIP(hipMemcpy(d_bounds + 16, kRobots[robot].span, 16 * sizeof(float), hipMemcpyHostToDevice));
        const size_t total = n * (size_t) dim;
        hipLaunchKernelGGL(vmv::halton_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), d_q, total, dim, skip, d_bounds, d_bounds + 16);
        VMV_HIP(hipGetLastError());
        VMV_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
        (void) hipFree(d_bounds);
        return VMV_OK;
    }

    const char *vmv_kernel_name(int robot, const char *entry_point)
    {
        static thread_local std::string name;
        if (!robot_ok(robot) || !entry_point) return nullptr;
        // validate_batch / validate_motion_batch run as two kernels (environment half, self-collision half);
        // the environment kernel dominates
        std::string ep(entry_point);
        if (ep == "validate_batch") ep = "validate_env";
        if (ep == "validate_motion_batch") ep = "validate_motion_env";
        name = std::string("vmv::") + kRobots[robot].name + "::" + ep + "_kernel";
        return name.c_str();
    }
}  // extern "C"
