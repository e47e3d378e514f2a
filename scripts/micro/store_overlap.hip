// store_overlap.hip -- do a wave's float32 MFMAs and its global stores overlap on gfx950?  (round 6: block1_conv1 of the float32 graphs
// takes 0.145 ms whatever its instruction stream looks like; alone its MFMAs are 0.053 ms and its stores 0.092 ms.)
// Every wave: ROWS x { MF x v_mfma_f32_16x16x4_f32 (four independent chains), then 16 KB of stores (16 x dwordx4, the 16 x 64-byte shape) }.
// MODE 0: the 16 stores in one burst behind the MFMAs; 1: four stores behind every quarter of the MFMAs; 2: burst, nontemporal;
// 3: the stores of row r issued in the middle of row r + 1's MFMAs.
// Build: hipcc --offload-arch=gfx950 -O3 -o ab/store_overlap scripts/micro/store_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int MF, int MODE>
__global__ void __launch_bounds__(256, 2) k(float *out, int rows, float a, float b)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, (short)0, 0x7ffffff0, 0x00020000);
    const unsigned vo = (unsigned)((lane & 15) * 256 + (lane >> 4) * 16);
    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{a, b, a, b};
    f32x4 old[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) old[m][n] = acc[m][n];
    for (int r = 0; r < rows; ++r) {
        const unsigned so = (unsigned)(((wave * rows + r) * 16384) & 0x7fffffff);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int i = 0; i < MF / 16; ++i)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + i, b + n, acc[m][n], 0, 0, 0);
            if (MODE == 1)
#pragma unroll
                for (int n = 0; n < 4; ++n) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[m][n]), rs, vo, so + m * 4096 + n * 64, 0);
            if (MODE == 3 && m == 1 && r > 0)
#pragma unroll
                for (int mm = 0; mm < 4; ++mm)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, old[mm][n]), rs, vo, so - 16384 + mm * 4096 + n * 64, 0);
        }
        if (MODE == 0 || MODE == 2)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[m][n]), rs, vo, so + m * 4096 + n * 64, MODE == 2 ? 2 : 0);
        if (MODE == 3)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) old[m][n] = acc[m][n];
    }
}

template <int MF, int MODE>
static void run(float *buf, int grid, int rows)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float sum = 0;
    for (int rep = 0; rep < 12; ++rep) {
        (void)hipEventRecord(e0);
        k<MF, MODE><<<grid, 256>>>(buf, rows, 1.0f, 0.5f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) sum += ms;
    }
    printf("grid %4d rows/wave %3d  MFMAs/row %3d  mode %d: %.4f ms  (%.0f MB stored)\n", grid, rows, MF, MODE, sum / 10, grid * 4.0 * rows * 16384 / 1e6);
}

int main()
{
    float *buf;
    if (hipMalloc(&buf, (size_t)1 << 30) != hipSuccess) return 1;
    for (int grid : {512, 1024}) {
        const int rows = 32768 / (grid * 4);
        run<0, 0>(buf, grid, rows);
        run<32, 0>(buf, grid, rows);
        run<64, 0>(buf, grid, rows);
        run<112, 0>(buf, grid, rows);
        run<224, 0>(buf, grid, rows);
        run<112, 1>(buf, grid, rows);
        run<112, 2>(buf, grid, rows);
        run<112, 3>(buf, grid, rows);
        run<224, 3>(buf, grid, rows);
    }
    return 0;
}
