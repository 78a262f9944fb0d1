// Microbenchmark: sustained issue rate of v_mfma_f32_16x16x4_f32 streams (cycles per MFMA per SIMD).
// hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// the same FLOP per wave on v_mfma_f32_32x32x2_f32 (64 cycles per instruction, 4096 FLOP): does the chip hold a different
// clock on this shape (MI355X_MICROARCH.md, DVFS give-back item 7)?
template <int NACC>
__global__ __launch_bounds__(512) void k32(float *out, unsigned long long *cyc, unsigned long long *rtc, int iters)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.0f;
    float a = (threadIdx.x % 97) * 1.37e-3f - 0.06f, b[NACC];
    for (int i = 0; i < NACC; i++) b[i] = ((threadIdx.x * 31 + i * 17) % 89) * 2.1e-3f - 0.09f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[i], acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rtc[blockIdx.x] = r1 - r0; }
}

template <int NACC>
__global__ __launch_bounds__(512) void k16clk(float *out, unsigned long long *cyc, unsigned long long *rtc, int iters)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0, 0, 0, 0};
    float a = (threadIdx.x % 97) * 1.37e-3f - 0.06f, b[NACC];
    for (int i = 0; i < NACC; i++) b[i] = ((threadIdx.x * 31 + i * 17) % 89) * 2.1e-3f - 0.09f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[i], acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rtc[blockIdx.x] = r1 - r0; }
}

template <class K>
void run_clk(const char *name, K kern, int nacc, double flop_per_mfma, int threads)
{
    // >= 2 s of back-to-back launches on non-trivial operands before the reading, then the median over workgroups
    int iters = 20000, blocks = 256;
    float *out; unsigned long long *cyc, *rtc;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8); hipMalloc(&rtc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0, total = 0;
    while (total < 2500.0f) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, rtc, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        total += ms;
    }
    std::vector<unsigned long long> h(blocks), hr(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), rtc, blocks * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz(blocks);
    double mean = 0;
    for (int i = 0; i < blocks; i++) { ghz[i] = (double)h[i] / (double)hr[i] * 0.1; mean += h[i]; }
    mean /= blocks;
    std::sort(ghz.begin(), ghz.end());
    const double mfma_per_simd = (double)iters * nacc * (threads / 64) / 4.0;
    printf("%-34s waves/WG %2d  cycles/MFMA/SIMD %.2f   in-kernel clock %.3f GHz (median; %.3f .. %.3f)  TF %.1f\n", name, threads / 64,
           mean / mfma_per_simd, ghz[blocks / 2], ghz[0], ghz[blocks - 1], flop_per_mfma * iters * nacc * (threads / 64) * blocks / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc); hipFree(rtc);
}

template <int NACC, int MODE>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters, int lds_bytes_dummy)
{
    extern __shared__ float lds[];
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b[NACC];
    for (int i = 0; i < NACC; i++) b[i] = (threadIdx.x + i) * 1e-4f;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 1e-5f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (MODE == 1) b[i] = lds[(threadIdx.x + i * 64 + it) & 4095];
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[i], acc[i], 0, 0, 0);
        }
        if (MODE == 2) {   // pinned 1:1 like the conv loop, loads for the NEXT iteration
#pragma unroll
            for (int i = 0; i < NACC; i++) b[i] = lds[(threadIdx.x + i * 64 + it) & 4095];
#pragma unroll
            for (int i = 0; i < NACC; i++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int MODE>
void run(const char *name, int threads, int lds)
{
    int iters = 2000, blocks = 256;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    hipFuncSetAttribute((const void *)k<NACC, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(threads), lds, 0, out, cyc, iters, lds);
        hipEventRecord(e1); hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= blocks;
    double mfma_per_simd = (double)iters * NACC * (threads / 64) / 4.0;
    printf("%-34s waves/WG %2d  cycles/MFMA/SIMD %.2f   wall-derived GHz %.3f  TF %.1f\n", name, threads / 64, mean / mfma_per_simd,
           mean / (ms * 1e-3) / 1e9, 2048.0 * iters * NACC * (threads / 64) * blocks / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run_clk("16x16x4, 15 acc, random operands", k16clk<15>, 15, 2048.0, 512);
    run_clk("32x32x2, 4 acc, random operands", k32<4>, 4, 4096.0, 512);
    run_clk("16x16x4, 15 acc, random operands", k16clk<15>, 15, 2048.0, 256);
    run_clk("32x32x2, 4 acc, random operands", k32<4>, 4, 4096.0, 256);
    const int L = 120 * 1024;
    run<15, 0>("15 acc, regs only", 512, L);
    run<15, 0>("15 acc, regs only", 256, L);
    run<4, 0>("4 acc, regs only", 256, L);
    run<30, 0>("30 acc, regs only", 256, L);
    run<15, 1>("15 acc, ds_read before each", 512, L);
    run<15, 2>("15 acc, pinned 1:1 next-iter reads", 512, L);
    run<15, 2>("15 acc, pinned 1:1 next-iter reads", 256, L);
    return 0;
}
