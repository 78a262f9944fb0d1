// Microbenchmark: sustained issue rate of v_mfma_f32_16x16x4_f32 streams (cycles per MFMA per SIMD).
// hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
