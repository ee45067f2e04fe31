// vmv_edge_tasks.hip — robot-independent half of the (edge, rake) task scheduling behind vmv_validate_motion_batch.
//
// validate_motion (planning/validate.hh:24-67) walks an edge rake by rake and stops at the first colliding one; its
// answer is the AND over the rakes, the early-out only skips work.  The task kernels (vmv_robot_tu.inc:
// rake_tasks_*_kernel) therefore evaluate the rakes of a batch as independent, equally sized tasks: pass 0 is rake 0 of
// every edge (the coarse rake that spans the whole edge), each later pass covers the rakes [lo, hi) of the edges that
// are still valid.  Between two passes this file turns "edge e still has c_e rakes in [lo, hi)" into the exclusive scan
// a task kernel searches to find its (edge, rake):
//     count_e = alive(e) ? clamp(steps_e, lo, hi) - lo : 0,   excl[e] = sum of count_0 .. count_{e-1},   total = sum.
// Also here: the per-(device, stream) scratch those kernels work in.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <utility>

#include "../../include/vamp_mvt_amd.h"
#include "vmv_common.h"

namespace vmv
{
    namespace
    {
        constexpr uint32_t kScanThreads = 256, kScanPerThread = 8;
        static_assert(kScanThreads * kScanPerThread == kEdgeScanBlock, "one scan block = 256 threads x 8 edges");

        // the 8 task counts of edges [8 * i, 8 * i + 8): one validity byte, two 16-byte loads of rake counts
        __device__ __forceinline__ void task_counts(const uint32_t *__restrict__ steps, const uint8_t *__restrict__ bits8,
                                                    const uint32_t n, const uint32_t first, const uint32_t lo,
                                                    const uint32_t hi, uint32_t (&c)[kScanPerThread])
        {
#pragma unroll
            for (uint32_t j = 0; j < kScanPerThread; ++j) c[j] = 0u;
            if (first >= n) return;
            const uint32_t byte = bits8[first >> 3];
            uint32_t s[kScanPerThread];
            if (first + kScanPerThread <= n)
            {
                const uint4 a = *reinterpret_cast<const uint4 *>(steps + first);
                const uint4 b = *reinterpret_cast<const uint4 *>(steps + first + 4);
                s[0] = a.x, s[1] = a.y, s[2] = a.z, s[3] = a.w, s[4] = b.x, s[5] = b.y, s[6] = b.z, s[7] = b.w;
            }
            else
            {
#pragma unroll
                for (uint32_t j = 0; j < kScanPerThread; ++j) s[j] = (first + j < n) ? steps[first + j] : 0u;
            }
#pragma unroll
            for (uint32_t j = 0; j < kScanPerThread; ++j)
            {
                const uint32_t top = s[j] < hi ? s[j] : hi;
                c[j] = (((byte >> j) & 1u) != 0u && top > lo) ? top - lo : 0u;
            }
        }

        __device__ __forceinline__ uint32_t wave_sum(uint32_t v)
        {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += (uint32_t) __shfl_xor((int) v, d);
            return v;
        }

        // block_sums[b] = tasks of edges [b * kEdgeScanBlock, (b + 1) * kEdgeScanBlock)
        __global__ __launch_bounds__(kScanThreads) void edge_block_sums_kernel(const uint32_t *__restrict__ steps,
                                                                               const uint8_t *__restrict__ bits8, const uint32_t n,
                                                                               const uint32_t lo, const uint32_t hi,
                                                                               uint32_t *__restrict__ block_sums)
        {
            __shared__ uint32_t part[kScanThreads / kWave];
            uint32_t c[kScanPerThread];
            task_counts(steps, bits8, n, blockIdx.x * kEdgeScanBlock + threadIdx.x * kScanPerThread, lo, hi, c);
            uint32_t mine = 0u;
#pragma unroll
            for (uint32_t j = 0; j < kScanPerThread; ++j) mine += c[j];
            const uint32_t w = wave_sum(mine);
            if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = w;
            __syncthreads();
            if (threadIdx.x == 0)
            {
                uint32_t s = 0u;
                for (uint32_t i = 0; i < kScanThreads / kWave; ++i) s += part[i];
                block_sums[blockIdx.x] = s;
            }
        }

        // excl[e] for the block's edges: the sum of the block sums before it (every block adds them up itself: there are
        // at most a few hundred) + the exclusive scan inside the block.  The last block also writes the pass's total.
        __global__ __launch_bounds__(kScanThreads) void edge_scan_write_kernel(const uint32_t *__restrict__ steps,
                                                                               const uint8_t *__restrict__ bits8, const uint32_t n,
                                                                               const uint32_t lo, const uint32_t hi,
                                                                               const uint32_t *__restrict__ block_sums,
                                                                               uint32_t *__restrict__ excl, uint32_t *__restrict__ total)
        {
            __shared__ uint32_t part[kScanThreads / kWave], before[kScanThreads / kWave];
            const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
            uint32_t prior = 0u;
            for (uint32_t i = threadIdx.x; i < blockIdx.x; i += kScanThreads) prior += block_sums[i];
            prior = wave_sum(prior);
            if (lane == 0) before[wave] = prior;
            const uint32_t first = blockIdx.x * kEdgeScanBlock + threadIdx.x * kScanPerThread;
            uint32_t c[kScanPerThread];
            task_counts(steps, bits8, n, first, lo, hi, c);
            uint32_t mine = 0u;
#pragma unroll
            for (uint32_t j = 0; j < kScanPerThread; ++j) mine += c[j];
            uint32_t incl = mine;  // inclusive scan over the wave
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1)
            {
                const uint32_t up = (uint32_t) __shfl_up((int) incl, d);
                incl += (lane >= (uint32_t) d) ? up : 0u;
            }
            if (lane == kWave - 1) part[wave] = incl;
            __syncthreads();
            uint32_t base = 0u, all = 0u;
            for (uint32_t i = 0; i < kScanThreads / kWave; ++i)
            {
                base += before[i];
                all += part[i];
                base += (i < wave) ? part[i] : 0u;
            }
            uint32_t run = base + incl - mine;
            if (first < n)
            {
                uint32_t o[kScanPerThread];
#pragma unroll
                for (uint32_t j = 0; j < kScanPerThread; ++j) o[j] = run, run += c[j];
                if (first + kScanPerThread <= n)
                {
                    *reinterpret_cast<uint4 *>(excl + first) = make_uint4(o[0], o[1], o[2], o[3]);
                    *reinterpret_cast<uint4 *>(excl + first + 4) = make_uint4(o[4], o[5], o[6], o[7]);
                }
                else
                {
#pragma unroll
                    for (uint32_t j = 0; j < kScanPerThread; ++j)
                        if (first + j < n) excl[first + j] = o[j];
                }
            }
            if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
            {
                uint32_t b = 0u;
                for (uint32_t i = 0; i < kScanThreads / kWave; ++i) b += before[i];
                *total = b + all;
            }
        }

        struct ScratchBuffer
        {
            void *base = nullptr;
            size_t bytes = 0;
        };
        std::mutex g_scratch_mutex;
        std::map<std::pair<int, hipStream_t>, ScratchBuffer> g_scratch;  // never freed at exit (see vmv_release_staging)
    }  // namespace

    EdgeScratchLease::EdgeScratchLease() : lock_(g_scratch_mutex, std::defer_lock) {}

    // Scratch of one launch sequence on `stream`: the sequences of one stream run in order, so they share one buffer; the
    // lease holds the pool's lock until the sequence is enqueued, so two host threads never interleave their sequences
    // on one stream.  The buffer grows on demand (hipFree of the old one waits for the work that still uses it).
    int EdgeScratchLease::acquire(hipStream_t stream, size_t n_edges)
    {
        int dev = -1;
        if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return hip_status(e, "hipGetDevice");
        const size_t n_pad = (n_edges + 63u) & ~size_t{63};
        const size_t blocks = (n_pad + kEdgeScanBlock - 1) / kEdgeScanBlock;
        const size_t want = n_pad * 2 * sizeof(uint32_t) + ((blocks + 63u) & ~size_t{63}) * sizeof(uint32_t) + 256;
        lock_.lock();
        ScratchBuffer &b = g_scratch[{dev, stream}];
        if (b.bytes < want)
        {
            if (b.base) (void) hipFree(b.base);
            b.base = nullptr, b.bytes = 0;
            const size_t grow = want < (size_t{1} << 20) ? (size_t{1} << 20) : want;
            if (hipError_t e = hipMalloc(&b.base, grow); e != hipSuccess)
            {
                b.base = nullptr;
                lock_.unlock();
                return hip_status(e, "hipMalloc(edge task scratch)");
            }
            b.bytes = grow;
        }
        uint32_t *p = static_cast<uint32_t *>(b.base);
        s.steps = p;
        s.excl = p + n_pad;
        s.block_sums = p + 2 * n_pad;
        s.total = s.block_sums + ((blocks + 63u) & ~size_t{63});
        return VMV_OK;
    }

    void release_edge_scratch()
    {
        std::lock_guard<std::mutex> g(g_scratch_mutex);
        for (auto &kv : g_scratch)
            if (kv.second.base) (void) hipFree(kv.second.base);
        g_scratch.clear();
    }

    int launch_edge_pass_scan(const EdgeScratch &s, const uint64_t *d_bits, uint32_t n, uint32_t lo, uint32_t hi, uint32_t slot,
                              hipStream_t stream)
    {
        const uint32_t blocks = (n + kEdgeScanBlock - 1) / kEdgeScanBlock;
        const uint8_t *bits8 = reinterpret_cast<const uint8_t *>(d_bits);
        if (blocks > 1)  // (a single block has nothing before it: one launch less for planner-sized batches)
            hipLaunchKernelGGL(edge_block_sums_kernel, dim3(blocks), dim3(kScanThreads), 0, stream, s.steps, bits8, n, lo, hi,
                               s.block_sums);
        hipLaunchKernelGGL(edge_scan_write_kernel, dim3(blocks), dim3(kScanThreads), 0, stream, s.steps, bits8, n, lo, hi,
                           s.block_sums, s.excl, s.total + slot);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return hip_status(e, "edge pass scan");
        return VMV_OK;
    }
}  // namespace vmv
