// vmv_common.h — declarations shared by the API translation unit and the per-robot kernel translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "vmv_device.h"

namespace vmv
{
    constexpr int kBlock = 256;  // 4 waves per workgroup; each wave owns one LDS slab
    constexpr int kWavesPerBlock = kBlock / kWave;
    constexpr uint32_t kMaxLdsBytes = 160u * 1024u;   // gfx950 LDS per CU / max per workgroup
    constexpr uint32_t kMaxPrimFloats = 12u * 1024u;  // 48 KiB of primitive records

    // what a launcher needs to know about a finalized environment
    struct EnvLaunch
    {
        const EnvDev *d_env;  // device copy
        EnvDev host;          // host copy (sizes)
    };

    // status codes are the VMV_* values of include/vamp_mvt_amd.h
    struct RobotLaunchers
    {
        // stage bit 1 = environment kernel (writes the words), bit 2 = self-collision kernel, bit 4 = attachment kernel
        // (both AND into them; the attachment kernel only runs for environments with an attachment)
        int (*validate)(const EnvLaunch &, const float *d_q, size_t n, uint64_t *d_bits, hipStream_t, int stages);
        int (*validate_motion)(const EnvLaunch &, const float *d_a, const float *d_b, size_t n, uint64_t *d_bits,
                               hipStream_t);
        int (*fk)(const float *d_q, size_t n, float *d_out, hipStream_t);
        // once per (environment, robot), after the robot's EnvDev is on the device: evaluates the robot's static links
        // against the environment and stores the answer in d_env->static_hit (synchronous)
        int (*prepare)(const EnvLaunch &, EnvDev *d_env);
        int (*eefk)(const float *d_q, size_t n, float *d_out16, hipStream_t);  // 4 x 4 row-major frames
        // contact report (Robot::fkcc_debug) from the fine spheres of launch fk
        int (*contacts)(const EnvLaunch &, const float *d_spheres, size_t n, uint32_t *d_env_words, uint32_t *d_pair_words,
                        hipStream_t);
        int n_self_pairs;
    };

    extern const RobotLaunchers kPandaLaunchers, kUr5Launchers, kFetchLaunchers, kBaxterLaunchers;

    int hip_status(hipError_t e, const char *what);  // records vmv_last_error(), maps to VMV_ERR_*
}  // namespace vmv
