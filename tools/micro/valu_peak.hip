// Micro-benchmark: sustained wave64 fp32 VALU issue rate on gfx950 for the instruction mix this path uses
// (v_add/v_sub/v_mul, no FMA contraction), at 1..8 waves per SIMD.  Prints cycles per VALU instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ void valu_loop(float *out, int iters, float a, float b)
{
    float v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) v[i] = a + (float) (threadIdx.x + i);
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int i = 0; i < ILP; ++i)
        {
            v[i] = v[i] * a;      // v_mul_f32
            v[i] = v[i] + b;      // v_add_f32
            v[i] = v[i] - a;      // v_sub_f32
            v[i] = v[i] * b;      // v_mul_f32
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *out;
    hipMalloc(&out, sizeof(float) * 64 * 32 * cus * 8);
    const int iters = 20000;
    constexpr int ILP = 4;
    for (int waves_per_simd : {1, 2, 3, 4, 6, 8})
    {
        const int threads = 64 * waves_per_simd;  // block = waves_per_simd waves; 4 blocks per CU -> one block per SIMD..
        const int blocks = cus * 4;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(valu_loop<ILP>, dim3(blocks), dim3(threads), 0, 0, out, 100, 1.0001f, 0.5f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(valu_loop<ILP>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = (double) iters * ILP * 4;
        const double waves_per_cu = 4.0 * waves_per_simd;
        const double instr_per_simd = instr_per_wave * waves_per_cu / 4.0;
        const double cycles = ms * 1e-3 * (p.clockRate * 1e3);
        printf("waves/SIMD %d: %.3f ms, %.2f cycles per VALU instr per SIMD (at %d MHz nominal), %.1f TFLOP/s-equivalent(1 flop/instr)\n",
               waves_per_simd, ms, cycles / instr_per_simd, p.clockRate / 1000,
               instr_per_simd * 4 * cus * 64 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
