// Microbenchmark: sustained issue rate of v_mfma_f32_16x16x32_bf16 streams on gfx950 (cycles per MFMA per SIMD, wall-derived
// clock, TFLOP/s) -- the building block of an fp32-emulating (3 x bf16 split) conv trunk, next to mfma_rate.hip (f32).
// hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_rate mfma_bf16_rate.hip && ./mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8 a, b[NACC];
    for (int e = 0; e < 8; e++) a[e] = (__bf16)(threadIdx.x * 1e-3f + e);
    for (int i = 0; i < NACC; i++)
        for (int e = 0; e < 8; e++) b[i][e] = (__bf16)((threadIdx.x + i) * 1e-4f + e);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[i], acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(const char *name, int threads)
{
    int iters = 4000, blocks = 256;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= blocks;
    double mfma_per_simd = (double)iters * NACC * (threads / 64) / 4.0;
    // s_memtime ticks at a fixed 100 MHz on this part: report time per MFMA from the wall clock
    double ns_per_mfma = ms * 1e6 / mfma_per_simd;
    printf("%-24s waves/WG %2d  ns/MFMA/SIMD %.2f  TFLOP/s %.1f\n", name, threads / 64, ns_per_mfma,
           16384.0 * iters * NACC * (threads / 64) * blocks / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<15>("bf16 16x16x32, 15 acc", 512);
    run<15>("bf16 16x16x32, 15 acc", 256);
    run<4>("bf16 16x16x32, 4 acc", 256);
    run<30>("bf16 16x16x32, 30 acc", 256);
    return 0;
}
