// vmv_common.h — declarations shared by the API translation unit and the per-robot kernel translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>

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

    // ---- (edge, rake) task scheduling of vmv_validate_motion_batch: the robot-independent half (vmv_edge_tasks.hip) ----
    constexpr uint32_t kEdgeScanBlock = 2048;        // edges per workgroup of the scan kernels
    constexpr uint32_t kEdgeSliceEdges = 1u << 20;   // a larger batch is validated slice by slice (32-bit task arithmetic)
    constexpr uint32_t kEdgeMaxPasses = 8;
    struct EdgeScratch
    {
        uint32_t *steps;       // [n] rakes of edge e (planning/validate.hh:41), written by pass 0
        uint32_t *excl;        // [n] exclusive scan of the current pass's task counts
        uint32_t *block_sums;  // [ceil(n / kEdgeScanBlock)]
        uint32_t *total;       // [kEdgeMaxPasses] tasks of each pass
    };
    class EdgeScratchLease
    {
        std::unique_lock<std::mutex> lock_;

    public:
        EdgeScratch s{};
        EdgeScratchLease();
        int acquire(hipStream_t stream, size_t n_edges);  // holds the pool until the lease goes out of scope
    };
    void release_edge_scratch();
    // task counts of the pass that covers rakes [lo, hi) of the edges still valid in d_bits -> s.excl, s.total[slot]
    int launch_edge_pass_scan(const EdgeScratch &s, const uint64_t *d_bits, uint32_t n, uint32_t lo, uint32_t hi, uint32_t slot,
                              hipStream_t stream);
}  // namespace vmv
