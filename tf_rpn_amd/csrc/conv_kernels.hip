// conv_kernels.hip -- the conv stack of the RPN forward path for gfx950 (MI355X).
//
// Keras Conv2D semantics (models/rpn_vgg16.py:16-20, models/rpn_mobilenet_v2.py:16-20):
// NHWC x HWIO cross-correlation, bias add, then activation.  Dense convolutions run as an
// implicit GEMM on the matrix cores:
//     M = output pixels (an 8 x 16 spatial patch of one image per workgroup = 128 rows)
//     N = output channels (32 / 64 / 128 per workgroup)
//     K = taps x input channels, walked tap by tap in 16-channel slices
// conv_igemm_f32 uses v_mfma_f32_32x32x2_f32: float32 in, float32 accumulate, bit-for-bit an
// ordered fmaf chain, so this path is the parity-clean one.  The im2col tile is never
// materialised in HBM: each K-slice of the A operand is gathered from the NHWC input
// (zero-filled at the borders) into LDS, k-major, so that MFMA fragment reads are
// conflict-free ds_read_b32; global loads for slice s+1 are issued before the MFMAs of
// slice s and written to the other LDS buffer afterwards (register-staged double buffer,
// one barrier per slice).  Bias, residual add, activation and the reg/cls split of the
// RPN head are fused into the epilogue.
//
// Roofline: MFMA-bound; 2*M*N*K flops per launch against 157.3 TFLOP/s (f32 MFMA peak).
#include "conv_kernels.h"

#include <cstdlib>
#include <cstring>

namespace rpn {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kConvThreads = 256;
constexpr int BK = 16;                 // K-slice (input channels per step)
constexpr int TH = 8, TW = 16;         // spatial patch of output pixels per workgroup
constexpr int BM = TH * TW;            // 128 GEMM rows
constexpr int LDA = BM + 2;            // +2 floats: conflict-free ds_write_b32 of the gathered tile

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

PackedShape packed_shape(int R, int S, int Cin, int Cout)
{
    PackedShape ps{};
    ps.R = R; ps.S = S; ps.Cin = Cin; ps.Cout = Cout;
    ps.generic = (Cin % 4 != 0) ? 1 : 0;
    ps.cout_pad = round_up(Cout, 32);
    if (ps.generic) {
        ps.cin_pad = Cin;
        ps.rows = round_up(R * S * Cin, BK);
    } else {
        ps.cin_pad = round_up(Cin, BK);
        ps.rows = R * S * ps.cin_pad;
    }
    return ps;
}

void pack_weights_host(const PackedShape &ps, const float *hwio, const float *scale, float *dst)
{
    memset(dst, 0, ps.floats() * sizeof(float));
    for (int t = 0; t < ps.R * ps.S; ++t)
        for (int c = 0; c < ps.Cin; ++c) {
            const int row = ps.generic ? (t * ps.Cin + c) : (t * ps.cin_pad + c);
            const float *src = hwio + ((size_t)t * ps.Cin + c) * ps.Cout;
            float *d = dst + (size_t)row * ps.cout_pad;
            for (int n = 0; n < ps.Cout; ++n) d[n] = scale ? src[n] * scale[n] : src[n];
        }
}

__global__ void pack_weights_kernel(PackedShape ps, const float *__restrict__ hwio, float *__restrict__ dst)
{
    const size_t total = ps.floats();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % ps.cout_pad);
        const int row = (int)(i / ps.cout_pad);
        int t, c;
        if (ps.generic) {
            t = row / ps.Cin;
            c = row - t * ps.Cin;
        } else {
            t = row / ps.cin_pad;
            c = row - t * ps.cin_pad;
        }
        float v = 0.0f;
        if (n < ps.Cout && c < ps.Cin && t < ps.R * ps.S) v = hwio[((size_t)t * ps.Cin + c) * ps.Cout + n];
        dst[i] = v;
    }
}

void pack_weights_device(const PackedShape &ps, const float *d_hwio, float *d_dst, hipStream_t stream)
{
    const size_t total = ps.floats();
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, stream, ps, d_hwio, d_dst);
}

__device__ __forceinline__ float apply_act(float v, int act)
{
    switch (act) {
        case ACT_RELU: return v > 0.0f ? v : 0.0f;
        case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        case ACT_RELU6: return v < 0.0f ? 0.0f : (v > 6.0f ? 6.0f : v);
        default: return v;
    }
}

// XCD-aware, bijective workgroup remap: hardware deals consecutive workgroup ids round-robin
// over the 8 XCDs, so give each XCD a contiguous run of logical tiles (the N-tiles of one
// M-tile, and neighbouring M-tiles, then share that XCD's L2).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// The second launch bound (a minimum of 2 waves per SIMD) is what keeps the register allocation lean: without it hipcc
// unrolls and prefetches every fragment read of a slice into 164 VGPRs + 64 AGPRs -- two 4-wave workgroups per CU, the
// matrix pipe busy 78 % of the time with the waves parked at the slice barrier for 18 % of theirs (profiles/r03_f32_pmc.txt).
// With it the 128 x 128 tile takes 104 VGPRs, accumulators included: up to four workgroups per CU (33 KB of LDS each) whose
// barrier waits cover one another.  Same instruction sequence per accumulator, same bits: 705 -> 722 images/s.
template <int WM, int WN, int MI, int NI, bool GENERIC, bool POOL, bool LEAN = false>      // (LEAN: see conv_igemm_f32_dma)
__global__ void __launch_bounds__(kConvThreads, 2)
conv_igemm_f32(ConvArgs a, int tiles_x, int tiles_y, int n_tiles)
{
    static_assert(WM * WN == 4 && WM * MI * 32 == BM, "tile shape");
    constexpr int BN = WN * NI * 32;
    constexpr int B4 = BK * BN / 4;                                   // float4s in a B slice
    constexpr int NB4 = (B4 + kConvThreads - 1) / kConvThreads;       // per thread

    __shared__ float As[2][BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = wg % n_tiles;
    int mt = wg / n_tiles;
    const int tx = mt % tiles_x;
    mt /= tiles_x;
    const int ty = mt % tiles_y;
    const int img = mt / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = nt * BN;

    const float *__restrict__ xin = a.x + (size_t)img * a.H * a.W * a.Cin;
    const int taps = a.R * a.S;

    // ---- per-thread gather coordinates of the A tile -------------------------------------
    // fast path: 2 pixels x one float4 (4 channels); generic path: 8 pixels x one scalar
    constexpr int NA = GENERIC ? 8 : 2;
    int a_iy0[NA], a_ix0[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int p = GENERIC ? ((tid >> 4) + 16 * j) : ((tid >> 2) + 64 * j);
        a_iy0[j] = (oy0 + (p >> 4)) * a.stride - a.pad_t;
        a_ix0[j] = (ox0 + (p & 15)) * a.stride - a.pad_l;
    }
    const int a_q = tid & 3;          // fast: channel quad
    const int a_kk = tid & 15;        // generic: k within the slice

    const int cpt = GENERIC ? 1 : (a.ps.cin_pad / BK);                 // slices per tap
    const int nsteps = GENERIC ? (a.ps.rows / BK) : (taps * cpt);

    float4 a_reg[GENERIC ? 1 : 2];
    float a_sreg[GENERIC ? 8 : 1];
    float4 b_reg[NB4];

    // Fast path: the gather walks (tap, 16-channel slice) incrementally and loads through raw buffer descriptors -- an
    // out-of-image pixel, a channel quad beyond Cin or a weight column beyond cout_pad is an out-of-range offset for
    // which the hardware returns zeros: no division, no predicated branch around a load in the loop (the first
    // version spent ~75 vector and ~50 scalar instructions and six exec-masked branches per 32 MFMAs here, on an
    // issue port the float32 MFMA shares with the vector ALU).
    constexpr unsigned kOobF = 0x80000000u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(xin), (short)0, (int)((size_t)a.H * a.W * a.Cin * 4 > 0x7fffffffull ? 0x7fffffff : (size_t)a.H * a.W * a.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.w), (short)0, (int)(a.ps.floats() * 4 > 0x7fffffffull ? 0x7fffffff : a.ps.floats() * 4), 0x00020000);
    int a_off0[2];                      // byte offset of (pixel j, tap (0,0), channel 4 a_q); may be negative (padding)
    unsigned a_taps[2];                 // bit t: tap t of pixel j lies inside the image
    unsigned b_off[NB4];                // byte offset of this thread's float4 in weight rows 0..15 (or kOobF)
    if constexpr (!GENERIC) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            a_off0[j] = ((a_iy0[j] * a.W + a_ix0[j]) * a.Cin + 4 * a_q) * 4;
            unsigned m = 0;
            int t = 0;
            for (int r = 0; r < a.R; ++r)
                for (int q = 0; q < a.S; ++q, ++t) {
                    const int iy = a_iy0[j] + r, ix = a_ix0[j] + q;
                    if (t < 32 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) m |= 1u << t;
                }
            a_taps[j] = m;
        }
#pragma unroll
        for (int i = 0; i < NB4; ++i) {
            const int e = tid + i * kConvThreads;
            const int kk = e / (BN / 4);
            const int col = n0 + 4 * (e % (BN / 4));
            const bool v = (B4 % kConvThreads == 0 || e < B4) && col < a.ps.cout_pad;
            b_off[i] = v ? (unsigned)((kk * a.ps.cout_pad + col) * 4) : kOobF;
        }
    }
    // walk state of the NEXT slice to load: tap index, its (r, s), slice within the tap, byte offset of the tap
    int w_tap = 0, w_r = 0, w_s = 0, w_cs = 0, w_tapoff = 0;
    const int b_step = 16 * a.ps.cout_pad * 4;          // bytes per slice of the packed weight matrix
    int w_brow = 0;

    auto load_global = [&](int step) {
        if constexpr (GENERIC) {
            const int k = step * BK + a_kk;
            const bool kv = k < taps * a.Cin;
            const int tap = k / a.Cin;
            const int c = k - tap * a.Cin;
            const int r = tap / a.S, s = tap - r * a.S;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int iy = a_iy0[j] + r, ix = a_ix0[j] + s;
                const bool v = kv && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                a_sreg[j] = v ? xin[((size_t)iy * a.W + ix) * a.Cin + c] : 0.0f;
            }
            const int row0 = step * BK;
#pragma unroll
            for (int i = 0; i < NB4; ++i) {
                const int e = tid + i * kConvThreads;
                const int kk = e / (BN / 4);
                const int col = n0 + 4 * (e % (BN / 4));
                const bool v = (B4 % kConvThreads == 0 || e < B4) && col < a.ps.cout_pad;
                b_reg[i] = v ? *reinterpret_cast<const float4 *>(a.w + (size_t)(row0 + kk) * a.ps.cout_pad + col)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            (void)step;                                  // (the walk state is the step)
            const int c0 = w_cs * BK;
            const bool cv = c0 + 4 * a_q < a.Cin;        // cin_pad > Cin: the quads past Cin are zero rows of the weights too
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool v = cv && ((a_taps[j] >> w_tap) & 1u);
                const unsigned off = v ? (unsigned)(a_off0[j] + w_tapoff + c0 * 4) : kOobF;
                a_reg[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
            }
#pragma unroll
            for (int i = 0; i < NB4; ++i)
                b_reg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrs, b_off[i], w_brow, 0));
            w_brow += b_step;
            if (++w_cs == cpt) {                         // next tap (scalar bookkeeping)
                w_cs = 0;
                ++w_tap;
                if (++w_s == a.S) { w_s = 0; ++w_r; }
                w_tapoff = (w_r * a.W + w_s) * a.Cin * 4;
            }
        }
    };

    auto store_lds = [&](int buf) {
        if constexpr (GENERIC) {
#pragma unroll
            for (int j = 0; j < 8; ++j) As[buf][a_kk][(tid >> 4) + 16 * j] = a_sreg[j];
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = (tid >> 2) + 64 * j;
                As[buf][4 * a_q + 0][p] = a_reg[j].x;
                As[buf][4 * a_q + 1][p] = a_reg[j].y;
                As[buf][4 * a_q + 2][p] = a_reg[j].z;
                As[buf][4 * a_q + 3][p] = a_reg[j].w;
            }
        }
#pragma unroll
        for (int i = 0; i < NB4; ++i) {
            const int e = tid + i * kConvThreads;
            if (B4 % kConvThreads == 0 || e < B4) {
                const int kk = e / (BN / 4);
                const int nn = 4 * (e % (BN / 4));
                *reinterpret_cast<float4 *>(&Bs[buf][kk][nn]) = b_reg[i];
            }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int am = wm * MI * 32 + (lane & 31);
    const int bn = wn * NI * 32 + (lane & 31);
    const int kh = lane >> 5;

    load_global(0);
    store_lds(0);
    __syncthreads();

    int cur = 0;
    for (int step = 0; step < nsteps; ++step) {
        const bool more = step + 1 < nsteps;
#ifndef RPN_EXP_F32_NOGATHER
        if (more) load_global(step + 1);
#endif
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            // MFMA step kk multiplies channels (4 q + e, 4 q + 2 + e), q = kk / 2, e = kk % 2: a quad's products enter an
            // accumulator in the order 0, 2, 1, 3 -- the order of conv_igemm_f32_dma (8-byte fragment reads), so that a layer
            // gives the same bits whichever of the two kernels and whichever tile width its grid size selects
            const int krow = GENERIC ? 2 * kk + kh : 4 * (kk >> 1) + 2 * kh + (kk & 1);
            float av[MI], bv[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) av[i] = As[cur][krow][am + 32 * i];
#pragma unroll
            for (int j = 0; j < NI; ++j) bv[j] = Bs[cur][krow][bn + 32 * j];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
#ifdef RPN_EXP_F32_NOGATHER      /* timing experiment only (results are wrong): no global loads, no LDS writes, no barrier in the loop */
        (void)more;
#else
        if (more) store_lds(cur ^ 1);
#ifdef RPN_EXP_F32_HALFBAR       /* timing experiment only (results are wrong): a barrier every second slice */
        if (step & 1)
#endif
        __syncthreads();
        cur ^= 1;
#endif
    }
#ifdef RPN_EXP_F32_NOEPI         /* timing experiment only: the accumulators are kept alive, nothing is written */
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
#endif

#define RPN_F32_EPI_LEAN LEAN
#include "conv_f32_epilogue.inc"
#undef RPN_F32_EPI_LEAN
}

// ---- the same GEMM with LDS-DMA staging (`buffer_load ... lds`) --------------------------------------------------------
// conv_igemm_f32<.., false> stages a slice through registers: 4 global loads, 10 ds_write, a wait and the registers in between
// per thread and slice, on the issue port the float32 MFMA shares with the vector ALU -- with the gather compiled out
// (RPN_EXP_F32_NOGATHER) the 128 x 128 tile ran at 0.86-0.88 of the MFMA peak against 0.76 with it.  Here every wave issues
// FOUR instructions per slice instead: two 1 KB pieces of the A tile and up to two of the B tile go global -> LDS directly
// (no registers, no ds_write), into the buffer the MFMAs are not reading; one `s_waitcnt vmcnt(0)` + barrier per slice as
// before, covered by the other workgroups of the CU.
//   A tile in LDS: pixel-major 16-byte pieces, piece e = 4 m + (q ^ ((m >> 2) & 3)) holds channels 4q .. 4q+3 of pixel m
//     of the slice.  The LDS side of a DMA is lane-linear, so the permutation is applied on the SOURCE address: instruction
//     i of a slice covers pixels 16 i .. 16 i + 15, lane l fetches pixel 16 i + (l >> 2), quad (l & 3) ^ ((m >> 2) & 3) --
//     16 whole 64-byte lines per instruction.  An out-of-image pixel / a quad beyond Cin is an out-of-range offset: zeros.
//   A fragments: lane (m, kh = lane >> 5) reads channels 2 kh, 2 kh + 1 of quad q as ONE ds_read_b64 (half the fragment
//     reads of the k-major tile; the XOR keeps the 32 lanes of a half-wave on 16 different piece columns, two lanes each:
//     the floor for 8-byte reads at a 64-byte pixel pitch) and feeds two MFMAs: k-pairs (4q, 4q + 2) then (4q + 1, 4q + 3).
//     A quad's four products therefore enter an accumulator in the order 0, 2, 1, 3 -- every non-generic layer runs on
//     this kernel at every tile width, so an image's bits do not depend on the batch it is in (tests: batch invariance).
//   B tile: the packed weight rows of the slice, [16][BN] floats, a linear copy (BN / 64 instructions per wave; at BN = 32
//     two waves issue one each).
// LEAN: the epilogue is the fast path alone (conv_f32_epilogue.inc: one float32 output, no residual, ReLU / linear, the image's output
// below 2 GiB -- conv_f32_lean_ok, checked by the launcher); with both epilogues in one kernel the general one's registers (the
// sigmoid's divide, the SPLIT16 output) set the allocation: 102 instead of 92, four workgroups per CU instead of five.
template <int WM, int WN, int MI, int NI, bool POOL, bool LEAN = false>
__global__ void __launch_bounds__(kConvThreads, 2)
conv_igemm_f32_dma(ConvArgs a, int tiles_x, int tiles_y, int n_tiles)
{
    static_assert(WM * WN == 4 && WM * MI * 32 == BM, "tile shape");
    constexpr int BN = WN * NI * 32;
    constexpr int B_INSTR = BK * BN / 256;              // 1 KB instructions per B slice: 8, 4 or 2
    constexpr int B_ROWS = 256 / BN;                    // weight rows per instruction: 2, 4 or 8
    constexpr int B_PER_WAVE = B_INSTR >= 4 ? B_INSTR / 4 : 1;

    __shared__ __attribute__((aligned(16))) float4 As4[2][BM * 4];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = wg % n_tiles;
    int mt = wg / n_tiles;
    const int tx = mt % tiles_x;
    mt /= tiles_x;
    const int ty = mt % tiles_y;
    const int img = mt / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = nt * BN;

    const float *__restrict__ xin = a.x + (size_t)img * a.H * a.W * a.Cin;
    const int cpt = a.ps.cin_pad / BK;                  // slices per tap
    const int nsteps = a.R * a.S * cpt;

    constexpr unsigned kOobF = 0x80000000u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(xin), (short)0, (int)((size_t)a.H * a.W * a.Cin * 4 > 0x7fffffffull ? 0x7fffffff : (size_t)a.H * a.W * a.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.w), (short)0, (int)(a.ps.floats() * 4 > 0x7fffffffull ? 0x7fffffff : a.ps.floats() * 4), 0x00020000);

    // this lane's two pieces of an A slice: instruction 2 wave + j, pixel m_j = 32 wave + 16 j + (lane >> 2)
    int a_off0[2];                      // byte offset of (pixel, tap (0,0), its channel quad) -- may be negative
    unsigned a_taps[2];                 // bit t: tap t of the pixel lies inside the image
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = 32 * wave + 16 * j + (lane >> 2);
        const int q = (lane & 3) ^ ((m >> 2) & 3);
        const int iy0 = (oy0 + (m >> 4)) * a.stride - a.pad_t, ix0 = (ox0 + (m & 15)) * a.stride - a.pad_l;
        a_off0[j] = ((iy0 * a.W + ix0) * a.Cin + 4 * q) * 4;
        unsigned msk = 0;
        int t = 0;
        for (int r = 0; r < a.R; ++r)
            for (int s = 0; s < a.S; ++s, ++t) {
                const int iy = iy0 + r, ix = ix0 + s;
                if (t < 32 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) msk |= 1u << t;
            }
        a_taps[j] = msk;
    }
    // this lane's pieces of a B slice: instruction ib = B_PER_WAVE * wave + t covers rows ib * B_ROWS ..., 4 columns per lane
    unsigned b_off0 = kOobF, b_off1 = kOobF;
    {
        const int col = n0 + 4 * (lane % (BN / 4));
        const int row0 = (B_PER_WAVE * wave) * B_ROWS + lane / (BN / 4);
        if (col < a.ps.cout_pad) {
            b_off0 = (unsigned)((row0 * a.ps.cout_pad + col) * 4);
            b_off1 = (unsigned)(((row0 + B_ROWS) * a.ps.cout_pad + col) * 4);
        }
    }
    const bool b_wave = B_PER_WAVE * wave < B_INSTR;     // (BN = 32: waves 0 and 1 only)

    // walk state of the NEXT slice to fetch: tap index, its (r, s), slice within the tap, byte offsets
    int w_tap = 0, w_r = 0, w_s = 0, w_cs = 0, w_tapoff = 0, w_brow = 0;
    const int b_step = 16 * a.ps.cout_pad * 4;          // bytes per slice of the packed weight matrix
#define RPN_F32_LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
    // (round 6) the lane's source offset of the tap being fetched is computed once per TAP (validity of the tap folded in) and the
    // slice's channel step rides in the scalar offset: a slice's fetch is four DMA issues and scalar bookkeeping -- it was ~18 vector
    // instructions and four exec branches per slice and wave, and vector-ALU time is float32-MFMA time on this chip
    const bool whole_slices = a.Cin % BK == 0;           // (else the last slice of a tap is partial: per-slice channel test)
    unsigned a_tapoff[2];
    auto set_tap = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) a_tapoff[j] = ((a_taps[j] >> w_tap) & 1u) ? (unsigned)(a_off0[j] + w_tapoff) : kOobF;
    };
    set_tap();
    auto fetch = [&](int buf) {
        const int c0 = w_cs * BK;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned off = a_tapoff[j];
            if (!whole_slices) {                         // (the lane's channel quad, recomputed: not worth a register across the loop)
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const int mq = 32 * wave + 16 * j + (ln >> 2);
                if (c0 + 4 * ((ln & 3) ^ ((mq >> 2) & 3)) >= a.Cin) off = kOobF;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, RPN_F32_LDS_PTR(&As4[buf][(2 * wave + j) * 64]), 16, off, c0 * 4, 0, 0);
        }
        if (b_wave) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, RPN_F32_LDS_PTR(&Bs[buf][0][0] + (B_PER_WAVE * wave) * 256), 16, b_off0,
                                                     w_brow, 0, 0);
            if constexpr (B_PER_WAVE == 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, RPN_F32_LDS_PTR(&Bs[buf][0][0] + (2 * wave + 1) * 256), 16, b_off1,
                                                         w_brow, 0, 0);
        }
        w_brow += b_step;
        if (++w_cs == cpt) {                             // next tap (scalar bookkeeping)
            w_cs = 0;
            ++w_tap;
            if (++w_s == a.S) { w_s = 0; ++w_r; }
            w_tapoff = (w_r * a.W + w_s) * a.Cin * 4;
            set_tap();
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int kh = lane >> 5;
    // fragment addresses: float index of (pixel am + 32 i, quad 0, channel 2 kh); quad q is at (that ^ (q << 2)) -- the XOR term
    // (m >> 2) & 3 sits in the same two bits
    int a_idx[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wm * MI * 32 + (lane & 31) + 32 * i;
        a_idx[i] = 16 * m + 4 * ((m >> 2) & 3);
    }
    const int bn = wn * NI * 32 + (lane & 31);

    fetch(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int cur = 0;
    for (int step = 0; step < nsteps; ++step) {
        if (step + 1 < nsteps) fetch(cur ^ 1);
        const float *Af = reinterpret_cast<const float *>(&As4[cur][0]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float2 av[MI];
            float b0[NI], b1[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) av[i] = *reinterpret_cast<const float2 *>(Af + ((a_idx[i] ^ (q << 2)) + 2 * kh));
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                b0[j] = Bs[cur][4 * q + 2 * kh][bn + 32 * j];
                b1[j] = Bs[cur][4 * q + 2 * kh + 1][bn + 32 * j];
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, b1[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the next slice has landed; this one's fragment reads are done
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }
#undef RPN_F32_LDS_PTR

#define RPN_F32_EPI_LEAN LEAN
#include "conv_f32_epilogue.inc"
#undef RPN_F32_EPI_LEAN
}

bool conv_f32_uses_dma(int tile_n, int generic);

template <int WM, int WN, int MI, int NI, bool POOL>
static hipError_t launch_variant_p(const ConvArgs &a, hipStream_t stream)
{
    constexpr int BN = WN * NI * 32;
    const int tiles_x = (a.OW + TW - 1) / TW, tiles_y = (a.OH + TH - 1) / TH;
    const int n_tiles = (a.Cout + BN - 1) / BN;
    const long long nblocks = (long long)tiles_x * tiles_y * a.B * n_tiles;
    if (nblocks <= 0 || nblocks > 0x7fffffffll) return hipErrorInvalidValue;
    const int OHo = POOL ? a.OH >> 1 : a.OH, OWo = POOL ? a.OW >> 1 : a.OW;
    const bool lean = !a.residual && a.split >= a.Cout && !a.out_split && (a.act == ACT_RELU || a.act == ACT_LINEAR) &&
                      (long long)OHo * OWo * a.ld1 * 4 <= 0x7fffffffll;
    if (a.ps.generic)
        hipLaunchKernelGGL((conv_igemm_f32<WM, WN, MI, NI, true, POOL>), dim3((unsigned)nblocks), dim3(kConvThreads), 0,
                           stream, a, tiles_x, tiles_y, n_tiles);
    else {
        // (conv_f32_uses_dma: the DMA kernel on every tile width since round 6; same bits as the register-staged kernel)
        if (conv_f32_uses_dma(BN, 0)) {
            if (lean)
                hipLaunchKernelGGL((conv_igemm_f32_dma<WM, WN, MI, NI, POOL, true>), dim3((unsigned)nblocks), dim3(kConvThreads), 0, stream,
                                   a, tiles_x, tiles_y, n_tiles);
            else
                hipLaunchKernelGGL((conv_igemm_f32_dma<WM, WN, MI, NI, POOL, false>), dim3((unsigned)nblocks), dim3(kConvThreads), 0, stream,
                                   a, tiles_x, tiles_y, n_tiles);
        }
        else if (lean)
            hipLaunchKernelGGL((conv_igemm_f32<WM, WN, MI, NI, false, POOL, true>), dim3((unsigned)nblocks), dim3(kConvThreads), 0,
                               stream, a, tiles_x, tiles_y, n_tiles);
        else
            hipLaunchKernelGGL((conv_igemm_f32<WM, WN, MI, NI, false, POOL, false>), dim3((unsigned)nblocks), dim3(kConvThreads), 0,
                               stream, a, tiles_x, tiles_y, n_tiles);
    }
    return hipGetLastError();
}

// The fused 2 x 2 pool (ConvArgs::pool) is a TEMPLATE parameter of the kernels: as a run-time branch of the epilogue it took
// the un-pooled 128 x 128 instantiations from 92 / 106 to 227 registers (round 4; tests/test_host.py now holds the budgets).
template <int WM, int WN, int MI, int NI>
static hipError_t launch_variant(const ConvArgs &a, hipStream_t stream)
{
    return a.pool ? launch_variant_p<WM, WN, MI, NI, true>(a, stream) : launch_variant_p<WM, WN, MI, NI, false>(a, stream);
}

// Output channels per workgroup tile (128 | 64 | 32) of the f32 implicit-GEMM conv for a layer: the widest tile that
// fits Cout -- unless that leaves CUs without a workgroup (single images, small feature maps: MobileNetV2's 1x1 layers
// at batch 1 give 32 workgroups of 128 x 128), then the narrowest tile that fills them, or failing that the one with
// the most workgroups.  "Fills" = two workgroups per CU (the 128 x 128 tile has one barrier per 16-channel slice: with one
// 4-wave workgroup per CU nothing covers its waits -- 31 x 31 x 512 at batch 8: 0.379 ms on 256 tiles of 128 x 128,
// 0.344 ms on 512 of 128 x 64).  Every variant handles every Cout (cout_pad is a multiple of 32).
int conv_f32_tile_n(int B, int OH, int OW, int Cout)
{
    static const int n_cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
            n = 256;
        return n;
    }();
    const long long m_tiles = (long long)((OW + TW - 1) / TW) * ((OH + TH - 1) / TH) * B;
    static const int occ = RPN_LAB_KNOB("RPN_F32_OCC", 2);     // workgroups per CU wanted
    int bn = Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);
    while (bn > 32 && m_tiles * ((Cout + bn - 1) / bn) < (long long)occ * n_cus) bn >>= 1;
    return bn;
}

bool conv_f32_uses_dma(int tile_n, int generic)
{
    // (round 6: on every tile width -- with the fetch's source offsets computed once per tap the DMA kernel is 3 % faster on the 64-wide
    // tiles too (block1_conv2 1.186 -> 1.146 ms, the 31 x 31 layers 0.336 -> 0.325 at batch 8); rounds 3-5 used it for 128-wide tiles
    // only.  Laboratory knob: 0 = the register-staged kernel everywhere, 3 = the old rule.  Same bits either way.)
    static const int dma = RPN_LAB_KNOB("RPN_F32_DMA", 1);
    (void)tile_n;
    return !generic && (dma == 1 || dma == 2 || (dma == 3 && tile_n == 128));
}

hipError_t launch_conv_f32(const ConvArgs &a, hipStream_t stream)
{
    const int bn = conv_f32_tile_n(a.B, a.OH, a.OW, a.Cout);
    if (bn == 128) return launch_variant<2, 2, 2, 2>(a, stream);       // 128 x 128
    if (bn == 64) return launch_variant<4, 1, 1, 2>(a, stream);        // 128 x 64
    return launch_variant<4, 1, 1, 1>(a, stream);                      // 128 x 32
}

// ---- RPN head: rpn_reg (linear) | rpn_cls (sigmoid), two 1x1 convs on the 512-channel rpn_conv output ------------------
// (models/rpn_vgg16.py:19-20 = models/rpn_mobilenet_v2.py:19-20).  As an implicit GEMM this is (B F F) x 512 x 5K with
// 5K = 45 (75 at 15 anchors): through conv_igemm_f32 it is one 128 x 32 tile walking K = 512 in 32 barrier-separated
// 16-channel steps, each a global -> LDS round trip -- 32-38 us whatever the batch, for 0.35 GFLOP.  Here a workgroup takes
// 32 pixels x all columns and its four waves split K (128 channels each): every lane loads its A operand -- 32
// consecutive channels of one pixel per 16-pixel block, k permuted accordingly: the MFMA does not care which k sits in
// which slot as long as A and B agree -- straight from global memory with eight 16-byte loads, all in flight at once, and
// the B operand from a fragment-ordered copy of the weight matrix (pack_head_weights_host).  The four partial sums meet
// in LDS and are added in wave order (deterministic), + bias, sigmoid on the objectness columns.  Exact f32 MFMA.
using f32x4_h = __attribute__((ext_vector_type(4))) float;

void pack_head_weights_host(const float *w /* [512][ld] */, int ld, int ncols, float *dst /* [4][NB][64][32] */)
{
    const int NB = (ncols + 15) / 16;
    for (int wv = 0; wv < 4; ++wv)
        for (int nb = 0; nb < NB; ++nb)
            for (int lane = 0; lane < 64; ++lane)
                for (int kk = 0; kk < 32; ++kk) {
                    const int k = 128 * wv + 32 * (lane >> 4) + kk, n = nb * 16 + (lane & 15);
                    dst[(((size_t)wv * NB + nb) * 64 + lane) * 32 + kk] = n < ncols ? w[(size_t)k * ld + n] : 0.0f;
                }
}

// PXV (16 | 4): pixels of its one M-block that a workgroup owns.  4 (one image at 500 x 500: 1024 pixels, four slabs of 2 KB
// each) makes 256 workgroups of 32 KB instead of 64 of 128 KB: a workgroup reads its operand at ONE CU's bandwidth (~10 B / clk),
// so the 64-workgroup form spent 5 us on loads alone; rows 4 .. 15 of the M-block are zeros, neither loaded nor stored (same bits:
// an output's sum order does not depend on the other rows).
template <int NB, int MB, int PXV = 16>          // MB: 16-pixel blocks per workgroup (2; 1 on small grids -- twice the workgroups, same bits)
__global__ void __launch_bounds__(256, 1)
rpn_head_kernel(const float *__restrict__ x, long long P, const float *__restrict__ wp, const float *__restrict__ bias,
                int n_reg, int n_cls, float *__restrict__ reg, float *__restrict__ cls, int n_slabs, long long slab_floats,
                const float *__restrict__ conv_bias)
{
    __shared__ __attribute__((aligned(16))) float part[4 * MB * NB * 64 * 4];        // [wave][tile][lane][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    static_assert(PXV == 16 || (PXV == 4 && MB == 1), "PXV");
    const long long p0 = (long long)blockIdx.x * (PXV * MB);
    // A: pixel p0 + 16 mb + lr, channels 128 wave + 32 lk .. + 32 (rows beyond P repeat the last pixel; never stored)
    f32x4_h av[MB][8];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        long long px = p0 + 16 * mb + lr;
        if (px >= P) px = P - 1;
        if (PXV < 16 && lr >= PXV) {            // rows this workgroup does not own: zero operand, no loads
#pragma unroll
            for (int i = 0; i < 8; ++i) av[mb][i] = f32x4_h{0.f, 0.f, 0.f, 0.f};
            continue;
        }
        const f32x4_h *src = reinterpret_cast<const f32x4_h *>(x + px * 512 + 128 * wave + 32 * lk);
#pragma unroll
        for (int i = 0; i < 8; ++i) av[mb][i] = src[i];
        if (n_slabs > 1) {              // K-tree producer: add the partial sums in tree order, + its bias, ReLU
            const f32x4_h *s1 = reinterpret_cast<const f32x4_h *>(x + (size_t)1 * slab_floats + px * 512 + 128 * wave + 32 * lk);
#pragma unroll
            for (int i = 0; i < 8; ++i) av[mb][i] = av[mb][i] + s1[i];                          // s0 + s1
            if (n_slabs == 4) {
                const f32x4_h *s2 = reinterpret_cast<const f32x4_h *>(x + (size_t)2 * slab_floats + px * 512 + 128 * wave + 32 * lk);
                const f32x4_h *s3 = reinterpret_cast<const f32x4_h *>(x + (size_t)3 * slab_floats + px * 512 + 128 * wave + 32 * lk);
#pragma unroll
                for (int i = 0; i < 8; ++i) av[mb][i] = av[mb][i] + (s2[i] + s3[i]);            // (s0 + s1) + (s2 + s3)
            }
            const f32x4_h *cb = reinterpret_cast<const f32x4_h *>(conv_bias + 128 * wave + 32 * lk);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4_h v = av[mb][i] + cb[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) av[mb][i][e] = v[e] > 0.0f ? v[e] : 0.0f;
            }
        }
    }
    f32x4_h acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4_h{0.f, 0.f, 0.f, 0.f};
    const f32x4_h *wsrc = reinterpret_cast<const f32x4_h *>(wp) + ((size_t)wave * NB * 64 + lane) * 8;
    f32x4_h bv[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[0][i] = wsrc[i];
    __builtin_amdgcn_sched_barrier(0);                 // all 24 loads above are issued before the first MFMA waits for one
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if (nb + 1 < NB) {
#pragma unroll
            for (int i = 0; i < 8; ++i) bv[(nb + 1) & 1][i] = wsrc[(size_t)(nb + 1) * 64 * 8 + i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb][i][e], bv[nb & 1][i][e], acc[mb][nb], 0, 0, 0);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            *reinterpret_cast<f32x4_h *>(&part[(((wave * MB + mb) * NB + nb) * 64 + lane) * 4]) = acc[mb][nb];
    __syncthreads();
    // tile t = mb * NB + nb is finished by wave t % 4: partial sums added in wave order
    for (int t = wave; t < MB * NB; t += 4) {
        const int mb = t / NB, nb = t - mb * NB;
        f32x4_h sum = *reinterpret_cast<const f32x4_h *>(&part[(((0 * MB + mb) * NB + nb) * 64 + lane) * 4]);
#pragma unroll
        for (int wv = 1; wv < 4; ++wv) {
            const f32x4_h v = *reinterpret_cast<const f32x4_h *>(&part[(((wv * MB + mb) * NB + nb) * 64 + lane) * 4]);
            sum += v;
        }
        const int n = nb * 16 + lr;
        if (n >= n_reg + n_cls) continue;
        const float b = bias[n];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long px = p0 + 16 * mb + 4 * lk + i;
            if (px >= P || 4 * lk + i >= PXV) continue;
            const float v = sum[i] + b;
            if (n < n_reg) reg[px * n_reg + n] = v;
            else cls[px * n_cls + (n - n_reg)] = 1.0f / (1.0f + expf(-v));
        }
    }
}

bool rpn_head_supported(int Cin, int ncols) { return Cin == 512 && ncols >= 1 && ncols <= 96; }

hipError_t launch_rpn_head(const float *x, long long P, const float *w_packed, const float *bias, int n_reg, int n_cls,
                           float *reg, float *cls, hipStream_t stream, int n_slabs, long long slab_floats, const float *conv_bias)
{
    if ((n_slabs != 1 && n_slabs != 2 && n_slabs != 4) || (n_slabs > 1 && !conv_bias)) return hipErrorInvalidValue;   // the kernel adds 1, 2 or 4 slabs, nothing else
    const int ncols = n_reg + n_cls;
    if (!rpn_head_supported(512, ncols) || P <= 0 || (P + 31) / 32 > 0x7fffffffll) return hipErrorInvalidValue;
    // a grid of fewer than 256 workgroups (one image): 16 pixels per workgroup instead of 32 -- its operand, up to four slabs of
    // 2 KB per pixel, is read at one CU's bandwidth per workgroup
    static const int mb1_max = RPN_LAB_KNOB("RPN_HEAD_MB1", 256);      // 32-pixel workgroups from this many on (16-pixel ones below)
    const int MB = (P + 31) / 32 < mb1_max ? 1 : 2;
    const bool px4 = MB == 1 && (P + 15) / 16 <= 64 && n_slabs > 1;     // <= 64 workgroups of up to four slabs: 4 pixels each instead
    const dim3 grid((unsigned)(px4 ? (P + 3) / 4 : (P + 16 * MB - 1) / (16 * MB)));
    const int NB = (ncols + 15) / 16;
#define RPN_HEAD(NB_)                                                                                                 \
    if (px4)                                                                                                          \
        hipLaunchKernelGGL((rpn_head_kernel<NB_, 1, 4>), grid, dim3(256), 0, stream, x, P, w_packed, bias, n_reg, n_cls, reg, cls,  \
                           n_slabs, slab_floats, conv_bias);                                                          \
    else if (MB == 1)                                                                                                 \
        hipLaunchKernelGGL((rpn_head_kernel<NB_, 1>), grid, dim3(256), 0, stream, x, P, w_packed, bias, n_reg, n_cls, reg, cls,     \
                           n_slabs, slab_floats, conv_bias);                                                          \
    else                                                                                                              \
        hipLaunchKernelGGL((rpn_head_kernel<NB_, 2>), grid, dim3(256), 0, stream, x, P, w_packed, bias, n_reg, n_cls, reg, cls,     \
                           n_slabs, slab_floats, conv_bias)
    switch (NB) {
        case 1: RPN_HEAD(1); break;
        case 2: RPN_HEAD(2); break;
        case 3: RPN_HEAD(3); break;
        case 4: RPN_HEAD(4); break;
        case 5: RPN_HEAD(5); break;
        default: RPN_HEAD(6); break;
    }
#undef RPN_HEAD
    return hipGetLastError();
}

// ---- MaxPooling2D(2,2) 'valid' -----------------------------------------------------------
__global__ void __launch_bounds__(256)
maxpool2x2_kernel(const float *__restrict__ x, int H, int W, int C4, int OH, int OW, long long total,
                  float *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long long t = i / C4;
        const int ox = (int)(t % OW);
        t /= OW;
        const int oy = (int)(t % OH);
        const long long b = t / OH;
        const float4 *p = reinterpret_cast<const float4 *>(x) + ((b * H + 2 * oy) * W + 2 * ox) * C4 + c;
        const float4 v00 = p[0], v01 = p[C4], v10 = p[(size_t)W * C4], v11 = p[(size_t)W * C4 + C4];
        float4 m;
        m.x = fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x));
        m.y = fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y));
        m.z = fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z));
        m.w = fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w));
        reinterpret_cast<float4 *>(out)[i] = m;
    }
}

hipError_t launch_maxpool2x2(const float *x, int B, int H, int W, int C, float *out, hipStream_t stream)
{
    if (C % 4 != 0) return hipErrorInvalidValue;
    const int OH = H / 2, OW = W / 2, C4 = C / 4;
    const long long total = (long long)B * OH * OW * C4;
    if (total == 0) return hipSuccess;
    long long grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3((unsigned)grid), dim3(256), 0, stream, x, H, W, C4, OH, OW, total,
                       out);
    return hipGetLastError();
}

// ---- depthwise 3x3 + bias + activation (MobileNetV2) ---------------------------------------
// HBM-bound: 4 B read + 4 B written per element (+ 9 taps served from L1/L2); one float4 of
// channels per lane, lanes run along channels so every access is a contiguous 16-byte vector.
__global__ void __launch_bounds__(256)
dwconv3x3_kernel(const float *__restrict__ x, int H, int W, int C4, const float *__restrict__ w,
                 const float *__restrict__ bias, int stride, int pad_t, int pad_l, int OH, int OW, int act,
                 long long total, float *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long long t = i / C4;
        const int ox = (int)(t % OW);
        t /= OW;
        const int oy = (int)(t % OH);
        const long long b = t / OH;
        float4 acc = bias ? reinterpret_cast<const float4 *>(bias)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = oy * stride + r - pad_t;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int ix = ox * stride + s - pad_l;
                if (ix < 0 || ix >= W) continue;
                const float4 v = reinterpret_cast<const float4 *>(x)[((b * H + iy) * W + ix) * C4 + c];
                const float4 k = reinterpret_cast<const float4 *>(w)[(r * 3 + s) * C4 + c];
                acc.x = fmaf(v.x, k.x, acc.x);
                acc.y = fmaf(v.y, k.y, acc.y);
                acc.z = fmaf(v.z, k.z, acc.z);
                acc.w = fmaf(v.w, k.w, acc.w);
            }
        }
        acc.x = apply_act(acc.x, act);
        acc.y = apply_act(acc.y, act);
        acc.z = apply_act(acc.z, act);
        acc.w = apply_act(acc.w, act);
        reinterpret_cast<float4 *>(out)[i] = acc;
    }
}

hipError_t launch_dwconv3x3(const float *x, int B, int H, int W, int C, const float *w, const float *bias,
                            int stride, int pad_t, int pad_l, int OH, int OW, int act, float *out,
                            hipStream_t stream)
{
    if (C % 4 != 0) return hipErrorInvalidValue;
    const int C4 = C / 4;
    const long long total = (long long)B * OH * OW * C4;
    if (total == 0) return hipSuccess;
    long long grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3((unsigned)grid), dim3(256), 0, stream, x, H, W, C4, w, bias, stride,
                       pad_t, pad_l, OH, OW, act, total, out);
    return hipGetLastError();
}

}  // namespace rpn
