// Microbenchmark: cycles per float32 MFMA on one SIMD (gfx950), by waves per SIMD and by shape.  hipcc --offload-arch=gfx950 -O3
// scripts/micro/mfma_f32_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int SHAPE, int NACC>      // SHAPE 16: 16x16x4, 32: 32x32x2; NACC independent accumulators walked round robin
__global__ void __launch_bounds__(1024) rate_kernel(float *out, unsigned long long *cyc, int iters, float a0, float b0)
{
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    unsigned long long t0, t1;
    if constexpr (SHAPE == 16) {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        float s = 0;
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i)
            for (int e = 0; e < 16; ++e) acc[i][e] = 0;
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        float s = 0;
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int SHAPE, int NACC>
void run(int threads, int blocks, const char *what)
{
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipMalloc(&cyc, (size_t)blocks * 16 * 8);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((rate_kernel<SHAPE, NACC>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0f, 0.5f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 16);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    double worst = 0;
    for (int w = 0; w < waves; ++w) worst = h[w] > worst ? h[w] : worst;
    const double per_simd_mfmas = (double)iters * NACC * (waves / 4.0);
    printf("%-44s waves/SIMD %d  NACC %2d  blocks %4d: %.2f cycles per MFMA per SIMD (slowest wave of block 0: %.0f cycles)\n", what, waves / 4, NACC,
           blocks, worst / per_simd_mfmas, worst);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    run<16, 8>(256, 1, "16x16x4 f32, one CU");
    run<16, 24>(256, 1, "16x16x4 f32, one CU");
    run<16, 24>(768, 1, "16x16x4 f32, one CU");
    run<16, 24>(768, 256, "16x16x4 f32, every CU");
    run<16, 2>(256, 1, "16x16x4 f32, one CU");
    run<32, 6>(256, 1, "32x32x2 f32, one CU");
    run<32, 6>(768, 1, "32x32x2 f32, one CU");
    run<32, 6>(768, 256, "32x32x2 f32, every CU");
    return 0;
}
