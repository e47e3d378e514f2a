// store_rate.hip -- how fast can a kernel that ONLY writes fill a buffer?  (round 6: block1_conv1 of the float32 graphs writes 512 MB and
// two unrelated kernels both take 0.152-0.155 ms = 3.4 TB/s.)  Shapes: (0) 1 KB contiguous per wave instruction (dwordx4),
// (1) 16 x 64-byte segments 256 B apart (a 16x16 accumulator's four channels per lane), (2) = (0) with nontemporal stores,
// (3) = (0) with one dword per lane (256 B per instruction).  Build: hipcc --offload-arch=gfx950 -O3 -o ab/store_rate scripts/micro/store_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int SHAPE>
__global__ void __launch_bounds__(256) fill(unsigned *out, size_t bytes_per_wave, unsigned v)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    char *base = reinterpret_cast<char *>(out) + wave * bytes_per_wave;
    const u32x4 val = {v, v + 1, v + 2, v + (unsigned)lane};
    if constexpr (SHAPE == 0 || SHAPE == 2) {
        for (size_t o = 0; o < bytes_per_wave; o += 1024) {
            u32x4 *p = reinterpret_cast<u32x4 *>(base + o + lane * 16);
            if constexpr (SHAPE == 2) __builtin_nontemporal_store(val, p);
            else *p = val;
        }
    } else if constexpr (SHAPE == 1) {
        for (size_t o = 0; o < bytes_per_wave; o += 4096)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                *reinterpret_cast<u32x4 *>(base + o + (lane & 15) * 256 + nb * 64 + (lane >> 4) * 16) = val;
    } else {
        for (size_t o = 0; o < bytes_per_wave; o += 256) *reinterpret_cast<unsigned *>(base + o + lane * 4) = v + lane;
    }
}

int main()
{
    const size_t sizes[3] = {(size_t)512 << 20, (size_t)128 << 20, (size_t)32 << 20};
    unsigned *buf;
    if (hipMalloc(&buf, sizes[0]) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (size_t bytes : sizes)
        for (size_t per_wave : {(size_t)4096, (size_t)16384, (size_t)65536})
            for (int shape = 0; shape < 4; ++shape) {
                const unsigned grid = (unsigned)(bytes / per_wave / 4);
                float best = 1e9f, sum = 0;
                for (int rep = 0; rep < 12; ++rep) {
                    hipEventRecord(e0);
                    switch (shape) {
                    case 0: fill<0><<<grid, 256>>>(buf, per_wave, rep); break;
                    case 1: fill<1><<<grid, 256>>>(buf, per_wave, rep); break;
                    case 2: fill<2><<<grid, 256>>>(buf, per_wave, rep); break;
                    default: fill<3><<<grid, 256>>>(buf, per_wave, rep); break;
                    }
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
                }
                printf("bytes %4zu MB  per-wave %6zu  shape %d  avg %.4f ms  best %.4f ms  -> %.2f TB/s (avg)\n", bytes >> 20, per_wave, shape,
                       sum / 10, best, bytes / (sum / 10 * 1e-3) / 1e12);
            }
    return 0;
}
