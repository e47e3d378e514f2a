// conv_wino_kernels.hip -- 3x3 stride-1 'same' convolution as Winograd F(2x2, 3x3) on the float32 MFMA (precision "f32w").
//
// The exact-float32 implicit GEMM (conv_kernels.hip) is bound by the float32 matrix pipe: 157.3 TFLOP/s / 156.55 GFLOP per VGG16
// image = 1005 images/s at 100 %.  Winograd's minimal filtering computes each 2 x 2 output tile of a 3 x 3 conv from
// a 4 x 4 input patch with 16 multiplications per (input channel, output channel) instead of 36: 2.25 x fewer MFMA flops, all still
// float32 multiply-adds with float32 accumulation -- the transforms only add and subtract (the filter transform's halves are exact
// in binary).  The sums are taken in another order than the direct conv's, so the results differ from it by rounding (measured
// against float64: 2-3e-6 at |y| ~ 4, the direct float32 kernel 1-2e-6; the contract of this path is 1e-4): it is a precision of
// its own ("f32w"), never a silent replacement of "f32".
//   y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        (Lavin & Gray; g: 3 x 3 filter, d: 4 x 4 input patch)
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
// One launch, everything fused: a workgroup (768 threads, one per CU: 155 KB of LDS) owns 8 x 8 Winograd tiles = 16 x 16 output
// pixels x 64 output channels and walks the input channels in slices of 8, ONE workgroup barrier per slice, two wave roles:
//   STAGING waves 8 .. 11 (one per SIMD), one slice ahead of the MFMAs:
//   * the slice's 18 x 18 x 8 input patch: buffer loads (out-of-image = out-of-range offset = zeros) -> registers -> LDS `raws`,
//     channel-major planes, double-buffered;
//   * the slice's 16 transformed filter matrices U[xi][8][64] (32 KB, packed that way by pack_weights_wino_host): LDS-DMA straight
//     into the buffer the MFMAs are not reading;
//   * B^T d B per (tile, channel): 8 LDS reads, 32 adds, 16 LDS writes -> V[xi][8][64 tiles], double-buffered;
//   MFMA waves 0 .. 7 (two per SIMD):
//   * 16 independent GEMMs M[xi] += V[xi]^T U[xi]: wave w owns xi = 2 w, 2 w + 1 for all 64 tiles x 64 channels (8 accumulator
//     blocks of 32 x 32 = 128 registers), 32 x mfma_f32_32x32x2 per slice and wave, one LDS read per MFMA, the operands of the
//     next group of four MFMAs requested before the current group is issued;
//   * epilogue in four quarters (32 tiles x 32 channels): accumulators -> LDS [xi][tile][channel], A^T M A + bias + activation by
//     (tile, channel) pairs, 128-byte NHWC stores (or, `pool`, the max of the tile's 2 x 2 outputs = MaxPooling2D(2, 2)).
// Measured (VGG16, batch 8, round 5; timing experiments -DRPN_EXP_WINO_*): with the staging compiled out the slice loop runs at
// 0.87 of the f32 MFMA rate (block3_conv2 0.556 ms -- the same ceiling the direct kernel shows without its gather); with it 0.685 ms:
// on this chip a SIMD's vector-ALU / LDS-issue work is ADDED to its float32-MFMA time, whichever wave it comes from (one instruction
// stream for everything: 0.694; roles: 0.685; the transform alone ~11 %, the patch + filter staging ~9 %).
// Every output's sum order is fixed by (slice, channel pair) alone: an image's bits do not depend on the batch it is in.
#include "conv_kernels.h"
#include "rpn_common.h"

#include <vector>

// Debug build only (-DRPN_STAMP, scripts/wn_stamp_probe.py): per-WAVE cycle stamps of the F(4x4, 3x3) kernels' slice loop -- slot 2 s =
// arrival at slice s's barrier, 2 s + 1 = release from it (s < 28), 60 = kernel entry, 61 = loop entry, 62 = loop exit, 63 = end.
#ifdef RPN_STAMP
__device__ unsigned long long g_wn_stamps[64 * 16 * 64];
#define RPN_WN_STAMP(k)                                                                                    \
    do {                                                                                                   \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 64 && (k) < 64)                                        \
            g_wn_stamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 64 + (k)] = __builtin_readcyclecounter(); \
    } while (0)
extern "C" int rpn_debug_read_wn_stamps(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wn_stamps), (size_t)n * 8);
}
#define RPN_WN_STAMP_RT(k)                                                                                 \
    do {                                                                                                   \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 64)                                                    \
            g_wn_stamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 64 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define RPN_WN_STAMP(k) ((void)0)
#define RPN_WN_STAMP_RT(k) ((void)0)
#endif

namespace rpn {

using f32x16_w = __attribute__((ext_vector_type(16))) float;
using f32x4_w = __attribute__((ext_vector_type(4))) float;
using f32x2_w = __attribute__((ext_vector_type(2))) float;
using u32x4_w = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kWinoThreads = 768;              // 8 MFMA waves + 4 staging waves
constexpr int kWinoKS = 8;                  // input channels per slice
constexpr int kWinoBN = 64;                 // output channels per workgroup
constexpr int kWinoTT = 8;                  // Winograd tiles per workgroup side (8 x 8 tiles = 16 x 16 output pixels)
constexpr int kRawPitch = 24, kRawPlane = 18 * kRawPitch;       // raw[channel][18][24] (18 columns used; two patch rows = 48 floats = 16 banks apart: the transform's 8-byte reads of a 16-lane group -- two tile rows -- are conflict-free; at pitch 20: 20 % of LDS cycles were conflicts)
constexpr int kUFloats = 16 * kWinoKS * kWinoBN;                // 8192 floats = 32 KB per slice and N tile
constexpr int kVFloats = 16 * kWinoKS * 64;                     // V[xi][k][tile]

size_t wino4_weight_floats(int Cin, int Cout);
size_t wino4n_weight_floats(int Cin, int Cout);
size_t wino_weight_floats(int Cin, int Cout, int variant)
{
    if (variant == 16) return wino4n_weight_floats(Cin, Cout);
    return variant == 4 || variant == 8 ? wino4_weight_floats(Cin, Cout) : (size_t)((Cout + kWinoBN - 1) / kWinoBN) * (Cin / kWinoKS) * kUFloats;
}
bool wino_supported(int Cin, int Cout) { return Cin >= kWinoKS && Cin % kWinoKS == 0 && Cout >= 32 && Cout % 32 == 0; }

// HWIO (3,3,Cin,Cout) -> U = G g G^T per (c, n), laid out [n_tile][slice][xi = 4 i + j][k][64]; scale[n] (BatchNorm fold) or null.
// Computed in double and rounded once (the halves are exact, the sums of three weights are not).
void pack_weights_wino4_host(const float *hwio, const float *scale, int Cin, int Cout, float *dst);
void pack_weights_wino4n_host(const float *hwio, const float *scale, int Cin, int Cout, float *dst);
void pack_weights_wino_host(const float *hwio, const float *scale, int Cin, int Cout, float *dst, int variant)
{
    if (variant == 16) {
        pack_weights_wino4n_host(hwio, scale, Cin, Cout, dst);
        return;
    }
    if (variant == 4 || variant == 8) {
        pack_weights_wino4_host(hwio, scale, Cin, Cout, dst);
        return;
    }
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int n_tiles = (Cout + kWinoBN - 1) / kWinoBN, n_slices = Cin / kWinoKS;
    for (int nt = 0; nt < n_tiles; ++nt)
        for (int s = 0; s < n_slices; ++s)
            for (int k = 0; k < kWinoKS; ++k)
                for (int nn = 0; nn < kWinoBN; ++nn) {
                    const int c = s * kWinoKS + k, n = nt * kWinoBN + nn;
                    double g[3][3];
                    for (int r = 0; r < 3; ++r)
                        for (int q = 0; q < 3; ++q)
                            g[r][q] = n < Cout ? (double)hwio[((size_t)(r * 3 + q) * Cin + c) * Cout + n] * (scale ? (double)scale[n] : 1.0) : 0.0;
                    double t[4][3];
                    for (int i = 0; i < 4; ++i)
                        for (int q = 0; q < 3; ++q) t[i][q] = G[i][0] * g[0][q] + G[i][1] * g[1][q] + G[i][2] * g[2][q];
                    for (int i = 0; i < 4; ++i)
                        for (int j = 0; j < 4; ++j) {
                            const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                            dst[(((size_t)nt * n_slices + s) * 16 + (4 * i + j)) * (kWinoKS * kWinoBN) + k * kWinoBN + nn] = (float)u;
                        }
                }
}

struct WinoArgs {
    const float *x;          // (B,H,W,Cin) float32 NHWC
    const float *u;          // pack_weights_wino_host
    const float *bias;       // (Cout) or null
    float *out;              // (B,H,W,Cout), or pooled (B,H/2,W/2,Cout)
    int B, H, W, Cin, Cout;
    int act, pool;
    // F(4x4, 3x3) with the input channels split over TWO workgroups per tile (variant 8: layers whose 16 x 32-pixel tiles fill only
    // half of the chip): each writes its raw partial outputs [512 pixels][64 channels] to part[(2 * tile + half)], the LAST arriver of
    // a tile (ticket) adds the two in fixed order (half 0 + half 1), + bias, activation, and writes the layer's output.
    int ksplit;              // 1 | 2
    float *part;             // ksplit 2: 2 x tiles x 32768 floats
    unsigned *tickets;       // ksplit 2: one per tile, zero between launches (the last arriver resets its ticket)
};

__device__ __forceinline__ float wino_act(float v, int act)
{
    if (act == ACT_RELU) return v > 0.0f ? v : 0.0f;
    if (act == ACT_RELU6) return v < 0.0f ? 0.0f : (v > 6.0f ? 6.0f : v);
    return v;
}

__device__ __forceinline__ int wino_xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

#define RPN_WINO_LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

__global__ void __launch_bounds__(kWinoThreads, 1)
conv3x3_wino_f32_kernel(WinoArgs a, int tiles_x, int tiles_y, int n_tiles)
{
    __shared__ __attribute__((aligned(16))) float Us[2][kUFloats];       // 64 KB: the slice's transformed filters, double-buffered
    __shared__ __attribute__((aligned(16))) float Vs[2][kVFloats];       // 64 KB: the slice's transformed input, double-buffered
    __shared__ __attribute__((aligned(16))) float raws[2][kWinoKS * kRawPlane];   // 22.5 KB: the slice's input patch, double-buffered

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int wg = wino_xcd_remap(blockIdx.x, gridDim.x);
    const int nt = wg % n_tiles;
    int mt = wg / n_tiles;
    const int tx = mt % tiles_x;
    mt /= tiles_x;
    const int ty = mt % tiles_y;
    const int img = mt / tiles_y;
    const int oy0 = ty * 2 * kWinoTT, ox0 = tx * 2 * kWinoTT, n0 = nt * kWinoBN;

    const float *__restrict__ xin = a.x + (size_t)img * a.H * a.W * a.Cin;
    const int n_slices = a.Cin / kWinoKS;
    constexpr unsigned kOob = 0x80000000u;
    const long long xbytes = (long long)a.H * a.W * a.Cin * 4;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xin), (short)0, (int)(xbytes > 0x7fffffffll ? 0x7fffffff : xbytes), 0x00020000);
    const float *ubase = a.u + (size_t)nt * n_slices * kUFloats;
    const long long ubytes = (long long)n_slices * kUFloats * 4;
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ubase), (short)0, (int)(ubytes > 0x7fffffffll ? 0x7fffffff : ubytes), 0x00020000);

    // Wave roles: waves 0 .. 7 are the MFMA waves (two per SIMD), waves 8 .. 11 (one per SIMD) the STAGING waves -- they fetch the
    // raw patches, run the input transform and issue the filter DMAs, so that the MFMA waves' instruction streams hold nothing but
    // fragment reads and MFMAs (with everything in one stream the transform's ~700 cycles and the staging's ~570 per slice were
    // simply added to the MFMAs' 4096: a wave waiting for the matrix pipe cannot issue its vector instructions either).
    const bool stager = wave >= 8;
    const int hid = tid - 512;             // staging thread 0 .. 255
    // a staging thread's pieces of a slice's raw patch: piece = (pixel of the 18 x 18 patch, half = 4 channels); 648 pieces
    unsigned r_off[3];                     // byte offset of (pixel, channel 4 * half) in slice 0, or kOob
    int r_lds[3];                          // float index of raw[4 * half][py][px]
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int piece = hid + j * 256;
        const int half = piece & 1, pix = piece >> 1;
        const int py = pix / 18, px = pix - py * 18;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        const bool v = stager && piece < 648 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        r_off[j] = v ? (unsigned)(((iy * a.W + ix) * a.Cin + 4 * half) * 4) : kOob;
        r_lds[j] = (4 * half) * kRawPlane + py * kRawPitch + px;
    }
    const bool third = hid + 512 < 648;

    const int t_tile = hid & 63, t_c0 = hid >> 6;                   // transform role: tile, channels t_c0 and t_c0 + 4 of the slice
    const int t_ty = t_tile >> 3, t_tx = t_tile & 7;
    const int kh = lane >> 5, l31 = lane & 31;
    const int sw = wave - 8;               // staging wave 0 .. 3

    // the pieces of the pipeline (all indices wave-uniform; no DMA builtin inside a lambda: the host pass drops the kernel's stub)
#define RPN_WINO_DMA_U(SLICE, BUF)                                                                                          \
    _Pragma("unroll") for (int q = 0; q < 8; ++q)                                                                           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(urs, RPN_WINO_LDS_PTR(&Us[BUF][(sw * 8 + q) * 256]), 16,                   \
                                                 (unsigned)(((sw * 8 + q) * 256 + lane * 4) * 4), (SLICE) * kUFloats * 4, 0, 0)
#define RPN_WINO_LOAD_RAW(CB)                                                                                               \
    do {                                                                                                                    \
        rr[0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, r_off[0], (CB), 0);                                              \
        rr[1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, r_off[1], (CB), 0);                                              \
        rr[2] = __builtin_amdgcn_raw_buffer_load_b128(xrs, r_off[2], (CB), 0);   /* (no piece: kOob, zeros, no traffic) */   \
    } while (0)
    u32x4_w rr[3] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    auto raw_to_lds = [&](int buf) {                                 // this thread's pieces -> channel-major planes
        float *raw = raws[buf];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < 2 || third) {
                const f32x4_w v = __builtin_bit_cast(f32x4_w, rr[j]);
#pragma unroll
                for (int i = 0; i < 4; ++i) raw[r_lds[j] + i * kRawPlane] = v[i];
            }
    };
    auto transform = [&](int buf) {                                  // V[buf] = B^T d B of (tile, channel) from raws[buf]
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int t_c = t_c0 + 4 * r;
            // ONE opaque index per patch, constants behind it: hipcc's waitcnt pass lets an LDS read overtake the filter DMAs in
            // flight only when the read carries the alias scope of its __shared__ array, and the pass that attaches the scopes
            // follows at most five address computations from the array -- with the index split into hoisted pieces the reads lost
            // theirs, and the transform waited for the whole memory latency of the slice's loads (s_waitcnt vmcnt(0) in front of it)
            int ro = buf * (kWinoKS * kRawPlane) + t_c * kRawPlane + (2 * t_ty) * kRawPitch + 2 * t_tx;
            asm volatile("" : "+v"(ro));
            const float *rp = &raws[0][0] + ro;
            float d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) d[i][j] = rp[i * kRawPitch + j];
            }
            float t[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[0][j] = d[0][j] - d[2][j];
                t[1][j] = d[1][j] + d[2][j];
                t[2][j] = d[2][j] - d[1][j];
                t[3][j] = d[1][j] - d[3][j];
            }
            float *vp = Vs[buf] + t_c * 64 + t_tile;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                vp[(4 * i + 0) * (kWinoKS * 64)] = t[i][0] - t[i][2];
                vp[(4 * i + 1) * (kWinoKS * 64)] = t[i][1] + t[i][2];
                vp[(4 * i + 2) * (kWinoKS * 64)] = t[i][2] - t[i][1];
                vp[(4 * i + 3) * (kWinoKS * 64)] = t[i][1] - t[i][3];
            }
        }
    };

    // The two roles are two separate instruction streams with the same NUMBER of workgroup barriers (2 + n_slices + 8): the
    // accumulators are live in the MFMA waves' stream only (one stream for both roles: 392 spilled registers at the 168 a
    // twelve-wave workgroup may have).
    if (stager) {
        // ---- prologue: slice 0 complete (raw -> V[0], U[0]), slice 1's raw patch in LDS
        RPN_WINO_LOAD_RAW(0);
        RPN_WINO_DMA_U(0, 0);
        raw_to_lds(0);
        if (n_slices > 1) RPN_WINO_LOAD_RAW(kWinoKS * 4);
        __syncthreads();
        transform(0);
        if (n_slices > 1) raw_to_lds(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- steady state, ONE barrier per slice.  Between the barriers of iteration s the MFMA waves run the MFMAs of slice s
        // (V[s & 1], U[s & 1]); the staging waves transform slice s + 1 (raws[(s + 1) & 1] -> V[(s + 1) & 1], last read by the MFMAs
        // of slice s - 1), fetch slice s + 2's patch into registers and, at the end, into raws[s & 1] (last read by the transform
        // of slice s), and DMA slice s + 1's filters into U[(s + 1) & 1].
        for (int s = 0; s < n_slices; ++s) {
            const int cur = s & 1;
#ifdef RPN_EXP_WINO_NOSTAGE      /* timing experiment (wrong results): the staging waves only keep the barriers */
            (void)cur;
            __syncthreads();
            continue;
#endif
            // (the full iterations and the last two are separate straight-line bodies: with the requests under conditions hipcc's
            // counted waits assume the DMAs may not have been issued and make raw_to_lds wait for all of them)
            if (s + 2 < n_slices) {
                RPN_WINO_LOAD_RAW((s + 2) * kWinoKS * 4);
                RPN_WINO_DMA_U(s + 1, cur ^ 1);
                transform(cur ^ 1);
                raw_to_lds(cur);
            } else if (s + 1 < n_slices) {
                RPN_WINO_DMA_U(s + 1, cur ^ 1);
                transform(cur ^ 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // slice s + 1's filters have landed
            __syncthreads();
        }
#pragma unroll 1
        for (int q = 0; q < 8; ++q) __syncthreads();                 // the epilogue's barriers (four quarters x 2)
        return;
    }

    f32x16_w acc[2][2][2];                 // [xi - 2 wave][M block][N block]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][mb][nb][e] = 0.0f;
    __syncthreads();
    __syncthreads();
    for (int s = 0; s < n_slices; ++s) {
        const int cur = s & 1;
        {
            // ---- 16 GEMMs: this wave's two xi, 64 tiles x 64 channels, k pairs (2 kp, 2 kp + 1); the operands of group g + 1 (4
            // MFMAs: one xi, one k pair) are requested before the MFMAs of group g are issued, into the other register set
            const float *U = Us[cur];
            const float *V = Vs[cur];
            float av[2][2], bv[2][2];
            auto ld = [&](int g, int buf) {
                const int xi = 2 * wave + (g >> 2), k = 2 * (g & 3) + kh;
                const float *vb = V + xi * (kWinoKS * 64) + l31 + k * 64;
                const float *ub = U + xi * (kWinoKS * kWinoBN) + l31 + k * kWinoBN;
                av[buf][0] = vb[0]; av[buf][1] = vb[32];
                bv[buf][0] = ub[0]; bv[buf][1] = ub[32];
            };
            constexpr int kCross = 0x2 | 0x4 | 0x10 | 0x200;        // what may cross: VALU, SALU, VMEM, DS writes
            ld(0, 0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g + 1 < 8) ld(g + 1, (g + 1) & 1);
                __builtin_amdgcn_sched_barrier(kCross);
                const int j = g >> 2, bf = g & 1;
                acc[j][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[bf][0], bv[bf][0], acc[j][0][0], 0, 0, 0);
                acc[j][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[bf][0], bv[bf][1], acc[j][0][1], 0, 0, 0);
                acc[j][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[bf][1], bv[bf][0], acc[j][1][0], 0, 0, 0);
                acc[j][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[bf][1], bv[bf][1], acc[j][1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(kCross);
            }
        }
        __syncthreads();
    }
#undef RPN_WINO_DMA_U
#undef RPN_WINO_LOAD_RAW

    // ---- epilogue: four quarters (M block, N block); M staging [xi][32 tiles][32 channels] = 64 KB over Us
    float *Ms = &Us[0][0];
    const int e_n = tid & 31;               // output role (the 512 threads of the MFMA waves): channel of the quarter, tiles e_t0 and e_t0 + 16
    const int e_t0 = tid >> 5;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int xi = 2 * wave + j;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * kh;            // tile of the block
                    Ms[(xi * 32 + row) * 32 + l31] = acc[j][mb][nb][e];
                }
            }
            __syncthreads();
            const int n = n0 + nb * 32 + e_n;
            const float bias = (a.bias && n < a.Cout) ? a.bias[n] : 0.0f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tl = e_t0 + 16 * h;                                   // tile inside the block
                const int tile = mb * 32 + tl;
                const int tyy = tile >> 3, txx = tile & 7;
                float m[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) m[i][jj] = Ms[((4 * i + jj) * 32 + tl) * 32 + e_n];
                float r[2][4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    r[0][jj] = m[0][jj] + m[1][jj] + m[2][jj];
                    r[1][jj] = m[1][jj] - m[2][jj] - m[3][jj];
                }
                float y[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    y[i][0] = r[i][0] + r[i][1] + r[i][2];
                    y[i][1] = r[i][1] - r[i][2] - r[i][3];
                }
                const int oy = oy0 + 2 * tyy, ox = ox0 + 2 * txx;
                if (n < a.Cout) {
                    if (a.pool) {
                        // MaxPooling2D(2, 2) 'valid': the tile's four outputs are one window; max commutes with + bias and
                        // the monotone activation; a window that reaches past an odd edge is dropped
                        const int PH = a.H >> 1, PW = a.W >> 1, py = oy >> 1, px = ox >> 1;
                        if (py < PH && px < PW) {
                            const float v = fmaxf(fmaxf(y[0][0], y[0][1]), fmaxf(y[1][0], y[1][1]));
                            a.out[(((size_t)img * PH + py) * PW + px) * a.Cout + n] = wino_act(v + bias, a.act);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj)
                                if (oy + i < a.H && ox + jj < a.W)
                                    a.out[(((size_t)img * a.H + oy + i) * a.W + ox + jj) * a.Cout + n] = wino_act(y[i][jj] + bias, a.act);
                    }
                }
            }
            __syncthreads();
        }
    }
}

// =================================================================================================================================
// F(4x4, 3x3): each 4 x 4 output tile from a 6 x 6 input patch with 36 multiplications per channel pair instead of 144 -- 4 x fewer
// MFMA flops than the direct conv, 1.78 x fewer than F(2x2, 3x3); its transforms multiply by small constants (4, 5, 2, 8 and, in
// the filter transform, 1/4, 1/6, 1/12, 1/24: U = G g G^T is computed in double and rounded once).  Measured on the whole VGG16
// graph against float64 (numpy restatement, 96 x 96 image): head outputs 6.4e-6 / 1.8e-6 where the direct float32 graph gives
// 1.8e-6 / 0.5e-6 and F(2x2, 3x3) 2.4e-6 / 0.6e-6 -- fifteen times inside the path's 1e-4 contract.
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// Same structure as the F(2x2, 3x3) kernel above: a 1024-thread workgroup per CU owns 4 x 8 tiles = 16 x 32 output pixels x 64 output
// channels and walks the input channels in slices of 4, one barrier per slice; four STAGING waves (18 x 34 x 4 patch -> LDS, the 6 x 6
// transforms -- a (tile, channel) pair by two threads, three output rows each), twelve MFMA waves (xi = 3 w .. 3 w + 2 each: 96
// accumulator registers; 12 MFMAs per slice and wave; the wave's filter fragments straight from L2 into its registers, a slice
// ahead).  Used where its 16 x 32-pixel tiles fill the chip; where they fill half of it (the 31 x 31 layers at batch 8) the input
// channels are split over two workgroups per tile (wino_variant 8, w4_ksplit_finish); smaller grids stay on F(2x2, 3x3).
constexpr int kW4Threads = 1024;            // 12 MFMA waves + 4 staging waves
constexpr int kW4KS = 4;                    // input channels per slice
constexpr int kW4Xi = 36;
constexpr int kW4Tickets = 16384;            // K split (variant 8): tickets at the front of the workspace
constexpr int kW4TY = 4, kW4TX = 8;         // Winograd tiles per workgroup: 4 x 8 (16 x 32 output pixels)
constexpr int kW4PatchH = 4 * kW4TY + 2, kW4PatchW = 4 * kW4TX + 2;      // 18 x 34
constexpr int kW4Pitch = 36, kW4Plane = kW4PatchH * kW4Pitch + 26;       // raw[channel][18][36] + 26: the plane stride is 2 (mod 32) floats, a transform wave's two channels fall on different banks
constexpr int kW4UFloats = kW4Xi * kW4KS * kWinoBN;                     // 9216 floats = 36 KB per slice and N tile
constexpr int kW4VFloats = kW4Xi * kW4KS * 32;                          // V[xi][k][32 tiles]: 18 KB

size_t wino4_weight_floats(int Cin, int Cout) { return (size_t)((Cout + kWinoBN - 1) / kWinoBN) * (Cin / kW4KS) * kW4UFloats; }

void pack_weights_wino4_host(const float *hwio, const float *scale, int Cin, int Cout, float *dst)
{
    static const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int n_tiles = (Cout + kWinoBN - 1) / kWinoBN, n_slices = Cin / kW4KS;
    for (int nt = 0; nt < n_tiles; ++nt)
        for (int s = 0; s < n_slices; ++s)
            for (int k = 0; k < kW4KS; ++k)
                for (int nn = 0; nn < kWinoBN; ++nn) {
                    const int c = s * kW4KS + k, n = nt * kWinoBN + nn;
                    double g[3][3];
                    for (int r = 0; r < 3; ++r)
                        for (int q = 0; q < 3; ++q)
                            g[r][q] = n < Cout ? (double)hwio[((size_t)(r * 3 + q) * Cin + c) * Cout + n] * (scale ? (double)scale[n] : 1.0) : 0.0;
                    double t[6][3];
                    for (int i = 0; i < 6; ++i)
                        for (int q = 0; q < 3; ++q) t[i][q] = G[i][0] * g[0][q] + G[i][1] * g[1][q] + G[i][2] * g[2][q];
                    for (int i = 0; i < 6; ++i)
                        for (int j = 0; j < 6; ++j) {
                            const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                            dst[(((size_t)nt * n_slices + s) * kW4Xi + (6 * i + j)) * (kW4KS * kWinoBN) + k * kWinoBN + (nn & 31) * 2 + (nn >> 5)] = (float)u;
                        }
                }
}

// one 1-D input transform B^T d: all six outputs, or (HALF 0 / 1) outputs 0..2 / 3..5 only
__device__ __forceinline__ void w4_bt6(const float (&d)[6], float (&t)[6])
{
    const float a = __builtin_fmaf(-4.0f, d[2], d[4]), b = __builtin_fmaf(-4.0f, d[1], d[3]);
    const float c = d[4] - d[2], e = d[3] - d[1];
    t[0] = __builtin_fmaf(4.0f, d[0], __builtin_fmaf(-5.0f, d[2], d[4]));
    t[1] = a + b;
    t[2] = a - b;
    t[3] = __builtin_fmaf(2.0f, e, c);
    t[4] = __builtin_fmaf(-2.0f, e, c);
    t[5] = __builtin_fmaf(4.0f, d[1], __builtin_fmaf(-5.0f, d[3], d[5]));
}
template <int HALF>
__device__ __forceinline__ void w4_bt3(const float (&d)[6], float (&t)[3])
{
    if (HALF == 0) {
        const float a = __builtin_fmaf(-4.0f, d[2], d[4]), b = __builtin_fmaf(-4.0f, d[1], d[3]);
        t[0] = __builtin_fmaf(4.0f, d[0], __builtin_fmaf(-5.0f, d[2], d[4]));
        t[1] = a + b;
        t[2] = a - b;
    } else {
        const float c = d[4] - d[2], e = d[3] - d[1];
        t[0] = __builtin_fmaf(2.0f, e, c);
        t[1] = __builtin_fmaf(-2.0f, e, c);
        t[2] = __builtin_fmaf(4.0f, d[1], __builtin_fmaf(-5.0f, d[3], d[5]));
    }
}
// one 1-D output transform A^T m
__device__ __forceinline__ void w4_at(const float (&m)[6], float (&y)[4])
{
    const float p = m[1] + m[2], q = m[1] - m[2], r = m[3] + m[4], s = m[3] - m[4];
    y[0] = m[0] + p + r;
    y[1] = __builtin_fmaf(2.0f, s, q);
    y[2] = __builtin_fmaf(4.0f, r, p);
    y[3] = __builtin_fmaf(8.0f, s, q) + m[5];
}

// a.pool as an opaque scalar: derived values (output sizes, edge tests) are invariant in the persistent kernels' tile loop and were
// hoisted to the top of the kernel -- one of them as a 0 / 1 VECTOR register held (and spilled to scratch) across the slice loop
__device__ __forceinline__ int w4_pool(const WinoArgs &a)
{
    int p = a.pool;
    asm volatile("" : "+s"(p));
    return p;
}
// The tail of an epilogue phase, shared by the F(4x4, 3x3) forms: a thread holds the 4 x 4 outputs y of one (tile, channel) pair and
// writes them (+ bias, activation; `pool`: the 2 x 2 maxima).  Round 6: one epilogue phase was ~490 vector instructions per thread
// (stamps: 20 k cycles of epilogue per workgroup, a quarter of a 128-channel layer) -- sixteen 64-bit address computations, a
// scalar branch on `act` and an exec-mask branch per output.  Now: buffer stores whose per-(row, column) step rides in the SCALAR
// offset (one 32-bit vector offset per thread), one branch on `act` for all sixteen values (ReLU = v_max), and the edge tests only
// for tiles that cross the image border (`interior`: wave-uniform in the wide form, a ballot in the 64-channel form).
// out_rs: the image's output tensor (H x W x Cout, or pooled); voff: byte offset of (oy, ox, n) (pooled: of (oy / 2, ox / 2, n)).
__device__ __forceinline__ void w4_store_outputs(const WinoArgs &a, float (&y)[4][4], __amdgpu_buffer_rsrc_t out_rs, unsigned voff, int oy, int ox,
                                                 float bias, bool interior)
{
    if (w4_pool(a)) {
        const int PH = a.H >> 1, PW = a.W >> 1;
        float v[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                v[i][j] = fmaxf(fmaxf(y[2 * i][2 * j], y[2 * i][2 * j + 1]), fmaxf(y[2 * i + 1][2 * j], y[2 * i + 1][2 * j + 1])) + bias;
        if (a.act == ACT_RELU) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) v[i][j] = fmaxf(v[i][j], 0.0f);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) v[i][j] = wino_act(v[i][j], a.act);
        }
        int row = PW * a.Cout * 4, col = a.Cout * 4;
        asm volatile("" : "+s"(row), "+s"(col));          // (opaque: see below)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (interior || ((oy >> 1) + i < PH && (ox >> 1) + j < PW))
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[i][j]), out_rs, voff, i * row + j * col, 0);
        return;
    }
    if (a.act == ACT_RELU) {                              // (v_max: a NaN becomes 0, as with `v > 0 ? v : 0`)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) y[p][q] = fmaxf(y[p][q] + bias, 0.0f);
    } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) y[p][q] = wino_act(y[p][q] + bias, a.act);
    }
    // (opaque scalars: in the persistent kernels the sixteen (row, column) steps are invariant in the tile loop and were hoisted to
    // the top of the kernel -- twenty scalar registers held for its whole life, which pushed others into vector-register lanes)
    int row = a.W * a.Cout * 4, col = a.Cout * 4;
    asm volatile("" : "+s"(row), "+s"(col));
    if (interior) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[p][q]), out_rs, voff, p * row + q * col, 0);
    } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (oy + p < a.H && ox + q < a.W)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[p][q]), out_rs, voff, p * row + q * col, 0);
    }
}

// the image's output tensor as a buffer (launch_conv3x3_wino checks that it is below 2 GiB)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t w4_out_rsrc(const WinoArgs &a, int img)
{
    const int pool = w4_pool(a);
    const int OH = pool ? a.H >> 1 : a.H, OW = pool ? a.W >> 1 : a.W;
    return __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)img * OH * OW * a.Cout, (short)0, OH * OW * a.Cout * 4, 0x00020000);
}

// Output transform of one (tile, channel) pair of an epilogue phase: thread -> (channel = tid & 31, tile = tid >> 5); MsA / MsB = the parked
// accumulators [xi][tile][32] of xi 0 .. 17 / 18 .. 35; Y = A^T M A in ONE pass over the 36 parked values: column j's six values ->
// R[0..3][j] = A^T m, then the four column transforms.  (Round 5 made two passes of two output rows each under a run-time loop,
// whose branch inside the unrolled column loop cut the reads into twelve groups of six, each waited for at once.)
__device__ __forceinline__ void w4_output_pair(const WinoArgs &a, const float *MsA, const float *MsB, int tid, int ph, int img, int oy0, int ox0, int n0,
                                               bool split, __amdgpu_buffer_rsrc_t part_rs, __amdgpu_buffer_rsrc_t out_rs)
{
    const int e_n = tid & 31, e_tile = tid >> 5;
    const int e_ty = e_tile >> 3, e_tx = e_tile & 7;
    const int n = n0 + ph * 32 + e_n;
    const float bias = (a.bias && n < a.Cout) ? a.bias[n] : 0.0f;
    const int oy = oy0 + 4 * e_ty, ox = ox0 + 4 * e_tx;
    if (n >= a.Cout) return;
    constexpr int kSc1 = 16;                                             // cache-policy bit 4 = sc1 (device scope, write-through)
    float R[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float m[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) m[i] = (i < 3 ? MsA : MsB)[((6 * (i % 3) + j) * 32 + e_tile) * 32 + e_n];     // xi = 6 i + j
        const float pp = m[1] + m[2], qq = m[1] - m[2], rr = m[3] + m[4], ss = m[3] - m[4];
        R[0][j] = m[0] + pp + rr;
        R[1][j] = __builtin_fmaf(2.0f, ss, qq);
        R[2][j] = __builtin_fmaf(4.0f, rr, pp);
        R[3][j] = __builtin_fmaf(8.0f, ss, qq) + m[5];
    }
    float y[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p) w4_at(R[p], y[p]);
    if (split) {                                                         // K split: the raw partial tile [16 x 32 pixels][64 channels]
        const unsigned pv = (unsigned)((((4 * e_ty) * 32 + 4 * e_tx) * kWinoBN + ph * 32 + e_n) * 4);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[p][q]), part_rs, pv, ((p * 32 + q) * kWinoBN) * 4, kSc1);
        return;
    }
    const int pool = w4_pool(a);
    const bool inside = pool ? ((oy >> 1) + 2 <= (a.H >> 1) && (ox >> 1) + 2 <= (a.W >> 1)) : (oy + 4 <= a.H && ox + 4 <= a.W);
    const bool interior = __builtin_amdgcn_ballot_w64(!inside) == 0ull;    // (a wave holds two tiles here)
    const unsigned voff = pool ? (unsigned)((((oy >> 1) * (a.W >> 1) + (ox >> 1)) * a.Cout + n) * 4) : (unsigned)(((oy * a.W + ox) * a.Cout + n) * 4);
    w4_store_outputs(a, y, out_rs, voff, oy, ox, bias, interior);
}

// K split, after the workgroup's partial tile is written: take the tile's ticket; the LAST of the two arrivers adds the halves in
// fixed order, + bias, activation, and writes the output tile.  Nobody waits for anybody.  Called by all 1024 threads.
// Visibility (as in ir_block_x3_kernel, mnv2_block_kernels.hip) without a cache-wide release / acquire pair -- this is, cell by cell,
// the first row of MI355X_MICROARCH.md's table "Hand-offs measured with sc1 loads in place of the acquire" (gfx950, ROCm 7.2):
//   stores   EVERY handed-off word leaves by a buffer store with sc1 (w4_output_pair: 4-byte stores; write-through to memory, the
//            line is dropped from the XCD's L2);
//   drain    every storing wave runs `s_waitcnt vmcnt(0)` after its stores, THEN the workgroup barrier, THEN one lane's agent-scope
//            atomic add on the tile's ticket -- the signal comes after the wait of every wave it signals for;
//   consumer the workgroup whose add returned 1 came last: the adding lane uses the returned value, publishes it through LDS, and the
//            other waves load only after the workgroup barrier that lane then joins;
//   loads    EVERY load of the handed-off words is a buffer load with sc1 to registers (never flat, never plain): served by L2 /
//            memory, the CU's L1 -- which no other CU's store ever refreshes -- is bypassed, so no `buffer_inv` is needed;
//   memory   hipMalloc, one workgroup per CU (1024 threads, 144 KB of LDS).
// A relaxed atomic is enough under those conditions: ordering comes from the drained write-through stores in front of it, not from
// the atomic's own semantics; an agent-scope release + acquire instead writes back / invalidates whole caches per workgroup (measured
// in round 3 on the same seam shape: configs[4] 0.384 -> 0.455 ms).  The last arriver resets the ticket for the next launch (the
// kernel boundary publishes it).  tests/test_gpu_wino.py::test_f32w_soak_two_handles_two_streams_and_a_busy_chip runs the seam
// under uneven load (another stream's persistent kernels own the CUs) and checks every output word.
__device__ __forceinline__ void w4_ksplit_finish(const WinoArgs &a, float *lds, int tid, int tile_id, int img, int oy0, int ox0, int n0)
{
    constexpr int kSc1 = 16;
    constexpr int BN = kWinoBN, TW = 32;
    constexpr int kPartF = 512 * kWinoBN;                // floats per partial tile: 512 pixels x 64 channels
    constexpr int kC4 = BN / 4;                          // 16-byte pieces per pixel
    unsigned *flag = reinterpret_cast<unsigned *>(lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(a.tickets + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == 1u) __hip_atomic_store(a.tickets + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
        *flag = ticket;
    }
    __syncthreads();
    if (*flag != 1u) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.part + (size_t)tile_id * 2 * kPartF, (short)0, 2 * kPartF * 4, 0x00020000);
    if (a.pool) {                                        // + block*_pool: 8 x TW / 2 pooled pixels x kC4 pieces, two per thread
        const int PH = a.H >> 1, PW = a.W >> 1;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = k * kW4Threads + tid;
            const int ppix = e / kC4, c4 = (e % kC4) * 4;
            const int py = (oy0 >> 1) + ppix / (TW / 2), px = (ox0 >> 1) + ppix % (TW / 2), n = n0 + c4;
            f32x4_w m = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int pix = (2 * (ppix / (TW / 2)) + (w >> 1)) * TW + 2 * (ppix % (TW / 2)) + (w & 1);
                const f32x4_w p0 = __builtin_bit_cast(f32x4_w, __builtin_amdgcn_raw_buffer_load_b128(rs, (pix * BN + c4) * 4, 0, kSc1));
                const f32x4_w p1 = __builtin_bit_cast(f32x4_w, __builtin_amdgcn_raw_buffer_load_b128(rs, kPartF * 4 + (pix * BN + c4) * 4, 0, kSc1));
#pragma unroll
                for (int i = 0; i < 4; ++i) m[i] = w == 0 ? p0[i] + p1[i] : fmaxf(m[i], p0[i] + p1[i]);
            }
            if (py < PH && px < PW && n < a.Cout) {
                f32x4_w b = {0.f, 0.f, 0.f, 0.f};
                if (a.bias) b = *reinterpret_cast<const f32x4_w *>(a.bias + n);
                f32x4_w v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = wino_act(m[i] + b[i], a.act);
                *reinterpret_cast<f32x4_w *>(a.out + (((size_t)img * PH + py) * PW + px) * a.Cout + n) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {                        // 8192 pieces of four channels, eight per thread
        const int e = k * kW4Threads + tid;
        const int pix = e / kC4, c4 = (e % kC4) * 4;
        const f32x4_w p0 = __builtin_bit_cast(f32x4_w, __builtin_amdgcn_raw_buffer_load_b128(rs, e * 16, 0, kSc1));
        const f32x4_w p1 = __builtin_bit_cast(f32x4_w, __builtin_amdgcn_raw_buffer_load_b128(rs, kPartF * 4 + e * 16, 0, kSc1));
        const int oy = oy0 + pix / TW, ox = ox0 + pix % TW, n = n0 + c4;
        if (oy < a.H && ox < a.W && n < a.Cout) {
            f32x4_w b = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) b = *reinterpret_cast<const f32x4_w *>(a.bias + n);
            f32x4_w v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = wino_act((p0[i] + p1[i]) + b[i], a.act);
            *reinterpret_cast<f32x4_w *>(a.out + (((size_t)img * a.H + oy) * a.W + ox) * a.Cout + n) = v;
        }
    }
}

// Persistent workgroups of the wide form (round 6): the grid is one workgroup per CU (or fewer), workgroup w walks the tiles
// L, L + G, L + 2 G, ... (L = wino_xcd_remap(w, G): consecutive tiles -- the N tiles of a pixel block, neighbouring blocks -- on one
// XCD, as before).  What it buys: the staging waves request the NEXT tile's first four slices in front of the epilogue of the current
// one, so a tile's prologue no longer waits for a first-touch memory latency (~2.5 k of its ~6.5 k cycles).
struct WnTile {
    int img, oy0, ox0, n0, nt;
};
// (tw: the tile's pixel width -- 16 in the wide form, 32 in the 64-channel form --, bn: its channel width)
__device__ __forceinline__ WnTile wn_tile(int t, int tiles_x, int tiles_y, int n_tiles, int tw = 16, int bn = 128)
{
    WnTile r;
    r.nt = t % n_tiles;
    int mt = t / n_tiles;
    const int tx = mt % tiles_x;
    mt /= tiles_x;
    const int ty = mt % tiles_y;
    r.img = mt / tiles_y;
    r.oy0 = ty * 16;
    r.ox0 = tx * tw;
    r.n0 = r.nt * bn;
    return r;
}
// tile + step, digit by digit with carries (st = wn_tile(step): its oy0 / ox0 carry the scaled digits): no division per tile
__device__ __forceinline__ void wn_advance(WnTile &tl, const WnTile &st, int tiles_x, int tiles_y, int n_tiles, int tw = 16, int bn = 128)
{
    int nt = tl.nt + st.nt, c = nt >= n_tiles;
    nt -= c ? n_tiles : 0;
    int ox = tl.ox0 + st.ox0 + tw * c;
    c = ox >= tw * tiles_x;
    ox -= c ? tw * tiles_x : 0;
    int oy = tl.oy0 + st.oy0 + 16 * c;
    c = oy >= 16 * tiles_y;
    oy -= c ? 16 * tiles_y : 0;
    tl.img += st.img + c;
    tl.nt = nt; tl.n0 = nt * bn; tl.ox0 = ox; tl.oy0 = oy;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wn_image_rsrc(const WinoArgs &a, int img)
{
    const long long xbytes = (long long)a.H * a.W * a.Cin * 4;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x) + (size_t)img * a.H * a.W * a.Cin, (short)0,
                                             (int)(xbytes > 0x7fffffffll ? 0x7fffffff : xbytes), 0x00020000);
}

// The staging role of the 64-channel F(4x4, 3x3) form (waves 12 .. 15), one instantiation per PART (output rows 3 PART .. 3 PART + 2
// of every pair's transform; pair_half: which 64 of the slice's 128 (tile, channel) pairs): straight-line code, see w4n_staging_role.
// Round 6, last pass: PERSISTENT like the wide form (one workgroup per CU walks tiles t_first, t_first + t_step, ...; the next tile's
// first two slices are requested in front of the current tile's epilogue).  With the input channels split over two workgroups
// (variant 8) the grid is two workgroups per tile as before and t_step = t_total: one turn of the tile loop.
template <int PART>
__device__ __forceinline__ void w4_staging_role(const WinoArgs &a, float *ldsU, float *ldsVR, float (*raws)[kW4KS * kW4Plane],
                                                int pair_half, int tid, int lane, int t_first, int t_step, int t_total, int tiles_x,
                                                int tiles_y, int n_tiles, int khalf, int s_begin, int n_slices)
{
    constexpr unsigned kOob = 0x80000000u;
    constexpr int kW4Half = 18 * 32 * 32;
    (void)kW4Half;
    const int hid = tid - 768;             // staging thread 0 .. 255
    // issue priority over the MFMA waves: the three MFMA waves of a SIMD always have a matrix instruction pending, and at equal
    // priority the staging wave's vector instructions were starved behind them -- the MFMA waves then waited for it at the
    // slice barrier (wide form 1 873 -> 1 976 images/s; no change for the 32-tile form, kept for symmetry)
    __builtin_amdgcn_s_setprio(3);
    // a staging thread's pixels of a slice's raw patch (18 x 34 = 612 pixels, one 16-byte load = 4 channels each)
    unsigned r_off[3];
    int r_lds[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int pix = hid + j * 256;
        const int py = pix / kW4PatchW, px = pix - py * kW4PatchW;
        r_lds[j] = py * kW4Pitch + px + ((py >> 2) & 1) * 2;      // rows of odd tile rows skewed by two floats (see transform)
    }
    auto set_offsets = [&](const WnTile &tl) {                    // this thread's source offsets inside the tile's image
        int h = hid;                                              // (opaque: the pixel's row / column are recomputed per tile, not kept in
        asm volatile("" : "+v"(h));                               // registers across the tile loop)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int pix = h + j * 256;
            const int py = pix / kW4PatchW, px = pix - py * kW4PatchW;
            const int iy = tl.oy0 - 1 + py, ix = tl.ox0 - 1 + px;
            const bool v = pix < kW4PatchH * kW4PatchW && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            r_off[j] = v ? (unsigned)(((iy * a.W + ix) * a.Cin) * 4) : kOob;
        }
    };
    auto image_rsrc = [&](int img) {                              // the image's input from this workgroup's first slice on
        const long long xbytes = (long long)a.H * a.W * a.Cin * 4 - (long long)s_begin * kW4KS * 4;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x) + (size_t)img * a.H * a.W * a.Cin + s_begin * kW4KS, (short)0,
                                                 (int)(xbytes > 0x7fffffffll ? 0x7fffffff : xbytes), 0x00020000);
    };
    const bool third = hid + 512 < kW4PatchH * kW4PatchW;
    // transform role: a lane = one (tile, channel) pair, two WAVES per pair (output rows 3 part .. 3 part + 2)
    // (the part is wave-uniform: with a lane-dependent part every wave runs both halves of the row transform under exec masks.
    // Before the staging waves had issue priority this measured SLOWER -- two waves read a pair's patch instead of two lanes
    // sharing one broadcast read --, with it +0.7 %.)
    constexpr int t_part = PART;                     // (template parameter: straight-line code, see w4n_staging_role)
    const int t_pair = pair_half * 64 + lane;
    const int t_tile = t_pair & 31, t_c = t_pair >> 5;
    const int t_ty = t_tile >> 3, t_tx = t_tile & 7;
    // TWO register sets for the raw patch: slice t travels in set t & 1, requested TWO iterations before the transform that reads
    // it and written to LDS one iteration after the request -- a request consumed in the iteration that issues it makes every
    // iteration at least one memory latency long, and the MFMA waves wait for the staging waves at the slice barrier (the matrix
    // pipe was 48 % busy).  (The same idea measured slower while the filter DMAs shared the stream: their conservative
    // vmcnt(0) waited for everything in flight.)
    u32x4_w rr[2][3] = {{{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}}, {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}}};
#define RPN_W4_LOAD_RAW(SLICE, SET)                                                                                         \
do {                                                                                                                    \
    rr[SET][0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, r_off[0], (SLICE) * kW4KS * 4, 0);                          \
    rr[SET][1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, r_off[1], (SLICE) * kW4KS * 4, 0);                          \
    rr[SET][2] = __builtin_amdgcn_raw_buffer_load_b128(xrs, r_off[2], (SLICE) * kW4KS * 4, 0);   /* (no piece: kOob, zeros, no traffic) */ \
} while (0)
#define RPN_W4_RAW_TO_LDS(BUF, SET)                                                                                         \
{                                                                                                                       \
    float *raw_ = raws[BUF];                                                                                            \
    _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_)                                                                    \
        if (j_ < 2 || third) {                                                                                          \
            const f32x4_w v_ = __builtin_bit_cast(f32x4_w, rr[SET][j_]);                                                \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) raw_[r_lds[j_] + i_ * kW4Plane] = v_[i_];                  \
        }                                                                                                               \
}
    auto transform = [&](int buf, int vofs) {               // raws[buf] (in ldsU) -> V at float offset vofs of ldsVR (one of three buffers)
        // T = B^T d over the patch rows for the three output rows of this part, then V = T B along the columns
        // Bank skew: tile columns are 4 floats apart and four tile rows 144 = 16 (mod 32) floats, so the 32 tiles of a wave's
        // 8-byte read would fall on 16 of the 32 banks; patch rows 4 .. 7, 12 .. 15 are stored two floats to the right
        // (raw_to_lds), which puts odd tile rows on the other 16.  A tile's rows 4, 5 belong to the next tile row: second base.
        // (indices in float PAIRS, so that a column pair is ONE ds_read_b64: see w4n_staging_role)
        static_assert((kW4KS * kW4Plane) % 2 == 0 && kW4Plane % 2 == 0 && kW4Pitch % 2 == 0, "pair index");
        int ro2 = (buf * (kW4KS * kW4Plane) + t_c * kW4Plane + (4 * t_ty) * kW4Pitch + 4 * t_tx + (t_ty & 1) * 2) / 2;
        int ro45 = ro2 + 1 - (t_ty & 1) * 2;
        asm volatile("" : "+v"(ro2), "+v"(ro45));
        const f32x2_w *rp = reinterpret_cast<const f32x2_w *>(ldsU) + ro2;
        const f32x2_w *rp45 = reinterpret_cast<const f32x2_w *>(ldsU) + ro45;
        float T[3][6];
#pragma unroll
        for (int jp = 0; jp < 3; ++jp) {              // column pairs (2 jp, 2 jp + 1)
            float d0[6], d1[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const f32x2_w dd = (r < 4 ? rp : rp45)[(r * kW4Pitch) / 2 + jp];
                d0[r] = dd[0];
                d1[r] = dd[1];
            }
            float t0[3], t1[3];
            if (t_part == 0) { w4_bt3<0>(d0, t0); w4_bt3<0>(d1, t1); }
            else { w4_bt3<1>(d0, t0); w4_bt3<1>(d1, t1); }
#pragma unroll
            for (int i = 0; i < 3; ++i) { T[i][2 * jp] = t0[i]; T[i][2 * jp + 1] = t1[i]; }
        }
        int vo = vofs + (18 * t_part) * (kW4KS * 32) + t_c * 32 + t_tile;
        asm volatile("" : "+v"(vo));
        float *vp = ldsVR + vo;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float v[6];
            w4_bt6(T[i], v);
#pragma unroll
            for (int j = 0; j < 6; ++j) vp[(6 * i + j) * (kW4KS * 32)] = v[j];
        }
    };
    // (Touching the filter slice three ahead -- one 128-byte line per staging thread, results unused, left in flight by a counted
    // wait -- to turn the LDS-DMA's Infinity-Cache / HBM misses into L2 hits measured SLOWER: 1 717 -> 1 639 images/s.)
    // Pipeline as in w4n_staging_role: slice t is requested in iteration t - 4 (register set t & 1), written to raws[t & 1] in
    // iteration t - 3, transformed into V[t % 3] in iteration t - 2, read by the MFMA waves during iteration t - 1, multiplied in
    // iteration t; no conditions in the loop body (requests clamped to the last slice, the tail works on stale data nobody reads).
    const int last = n_slices - 1;
    const bool split = a.ksplit == 2;
    WnTile tl = wn_tile(t_first, tiles_x, tiles_y, n_tiles, 4 * kW4TX, kWinoBN);
    __amdgpu_buffer_rsrc_t xrs = image_rsrc(tl.img);
    set_offsets(tl);
    RPN_W4_LOAD_RAW(0, 0);
    RPN_W4_LOAD_RAW(1 < last ? 1 : last, 1);
    for (int t = t_first; t < t_total; t += t_step) {
    const int tile_id = t, img = tl.img, oy0 = tl.oy0, ox0 = tl.ox0, n0 = tl.n0;
    u32x4_w p2[3], p3[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        p2[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, r_off[j], (2 < last ? 2 : last) * kW4KS * 4, 0);
        p3[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, r_off[j], (3 < last ? 3 : last) * kW4KS * 4, 0);
    }
    RPN_W4_RAW_TO_LDS(0, 0);
    RPN_W4_RAW_TO_LDS(1, 1);
    __syncthreads();                                                       // (A)
    transform(0, 0);
    transform(1, kW4VFloats);
    __syncthreads();                                                       // (B)
#pragma unroll
    for (int j = 0; j < 3; ++j) rr[0][j] = p2[j];
    RPN_W4_RAW_TO_LDS(0, 0);
#pragma unroll
    for (int j = 0; j < 3; ++j) rr[1][j] = p3[j];
    __syncthreads();                                                       // (C): V[0], V[1] complete, slice 2 in raws[0], slice 3 in register set 1
#define RPN_W4_STAGE(S, SET)                                                                                                \
{                                                                                                                       \
    RPN_W4_LOAD_RAW((S) + 4 < last ? (S) + 4 : last, SET);                                                              \
    transform(SET, vnext);                                           /* slice S + 2: raws[S & 1] -> V[(S + 2) % 3] */   \
    RPN_W4_RAW_TO_LDS((SET) ^ 1, (SET) ^ 1);                         /* slice S + 3 */                                  \
    vnext = vnext + kW4VFloats == 3 * kW4VFloats ? 0 : vnext + kW4VFloats;                                              \
    __syncthreads();                                                                                                    \
}
    {
        int vnext = 2 * kW4VFloats;                                  // V buffer of slice s + 2
        int s = 0;
        for (; s + 1 < n_slices; s += 2) {
            RPN_W4_STAGE(s, 0);
            RPN_W4_STAGE(s + 1, 1);
        }
        if (s < n_slices) RPN_W4_STAGE(s, 0);
    }
#undef RPN_W4_STAGE
    // the next tile's first two slices: in flight across this tile's epilogue
    if (t + t_step < t_total) {
        tl = wn_tile(t + t_step, tiles_x, tiles_y, n_tiles, 4 * kW4TX, kWinoBN);     // (by division: see the MFMA role)
        xrs = image_rsrc(tl.img);
        set_offsets(tl);
        RPN_W4_LOAD_RAW(0, 0);
        RPN_W4_LOAD_RAW(1 < last ? 1 : last, 1);
    }
    // ---- epilogue: two phases of 32 output channels; every thread of the workgroup transforms one (tile, channel) pair per phase
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        split ? a.part + ((size_t)tile_id * 2 + khalf) * (512 * kWinoBN) : nullptr, (short)0, split ? 512 * kWinoBN * 4 : 0, 0x00020000);
    int tid_o = tid;                                             // (opaque: what the epilogue derives from the thread index is invariant in
    asm volatile("" : "+v"(tid_o));                              //  the tile loop and was hoisted in front of it -- into scratch)
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        __syncthreads();                                         // the phase's accumulators are in LDS
        w4_output_pair(a, ldsU, ldsVR, tid_o, ph, img, oy0, ox0, n0, split, prs, w4_out_rsrc(a, img));
        __syncthreads();                                         // the staging area may be overwritten
    }
    if (a.ksplit == 2) w4_ksplit_finish(a, ldsU, tid_o, tile_id, img, oy0, ox0, n0);
    }   // tile loop
#undef RPN_W4_LOAD_RAW
#undef RPN_W4_RAW_TO_LDS
}

__global__ void __launch_bounds__(kW4Threads, 1)
conv3x3_wino4_f32_kernel(WinoArgs a, int tiles_x, int tiles_y, int n_tiles)
{
    // two arrays of 72 KB.  The slice loop uses ldsVR only: V x 2 (36 KB), raw x 2 (21 KB); the epilogue parks the 36 x 32 x 32
    // accumulators of a phase (144 KB) over both: xi 0 .. 17 in ldsU, xi 18 .. 35 in ldsVR.  (Two arrays since the filters still came
    // by LDS-DMA into ldsU: hipcc's waitcnt pass lets an LDS access overtake an LDS-DMA in flight only when the two carry the alias
    // scopes of DIFFERENT __shared__ arrays.)
    constexpr int kW4Half = 18 * 32 * 32;                                  // floats per array
    static_assert(3 * kW4VFloats <= kW4Half && 2 * kW4KS * kW4Plane <= kW4Half, "LDS layout");
    __shared__ __attribute__((aligned(16))) float ldsU[kW4Half];
    __shared__ __attribute__((aligned(16))) float ldsVR[kW4Half];
    // slice loop (round 6): V x 3 in ldsVR (54 KB), the raw patch x 2 in ldsU (21 KB; the filters no longer pass through LDS)
    float (*raws)[kW4KS * kW4Plane] = reinterpret_cast<float (*)[kW4KS * kW4Plane]>(ldsU);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // persistent workgroups (see w4_staging_role): tiles t_first, t_first + t_step, ...; both roles walk the same tiles with the same
    // number of barriers per tile.  K split (variant 8): two workgroups per tile -- neighbours: same XCD, same time --, one turn.
    const int wg = wino_xcd_remap(blockIdx.x, gridDim.x);
    const int t_total = tiles_x * tiles_y * a.B * n_tiles;
    const int khalf = a.ksplit == 2 ? (wg & 1) : 0;
    const int t_first = a.ksplit == 2 ? wg >> 1 : wg, t_step = a.ksplit == 2 ? t_total : (int)gridDim.x;

    const int all_slices = a.Cin / kW4KS;
    const int s_begin = khalf ? all_slices / 2 : 0;      // this workgroup's slices [s_begin, s_begin + n_slices)
    const int n_slices = a.ksplit == 2 ? (khalf ? all_slices - all_slices / 2 : all_slices / 2) : all_slices;
    const long long ubytes = (long long)n_slices * kW4UFloats * 4;
    const bool split = a.ksplit == 2;

    const bool stager = wave >= 12;

    if (stager) {
        const int sw = wave - 12;          // t_part = sw & 1, pairs (sw >> 1) * 64 .. + 63
        if (sw & 1) w4_staging_role<1>(a, ldsU, ldsVR, raws, sw >> 1, tid, lane, t_first, t_step, t_total, tiles_x, tiles_y, n_tiles, khalf, s_begin, n_slices);
        else w4_staging_role<0>(a, ldsU, ldsVR, raws, sw >> 1, tid, lane, t_first, t_step, t_total, tiles_x, tiles_y, n_tiles, khalf, s_begin, n_slices);
        return;
    }

    // ---- MFMA waves: xi = 3 wave + j, 32 tiles x 64 channels
    for (int t = t_first; t < t_total; t += t_step) {
    // (the tile's coordinates by division, once per ~60 k-cycle tile: carried incrementally -- wn_advance -- they were ten more scalar
    // registers across a slice loop that already parks scalars in vector-register lanes)
    const WnTile tl = wn_tile(t, tiles_x, tiles_y, n_tiles, 4 * kW4TX, kWinoBN);
    const int tile_id = t, img = tl.img, oy0 = tl.oy0, ox0 = tl.ox0, n0 = tl.n0;
    // (the lane index re-derived per tile, opaque: what is computed from it is otherwise hoisted out of the tile loop and kept in
    // registers the 128-register slice loop does not have -- see conv3x3_wino4n_f32_kernel)
    int lane_t;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
    const int kh = lane_t >> 5, l31 = lane_t & 31;
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.u) + ((size_t)tl.nt * all_slices + s_begin) * kW4UFloats, (short)0,
                                                                          (int)(ubytes > 0x7fffffffll ? 0x7fffffff : ubytes), 0x00020000);
    f32x16_w acc[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][nb][e] = 0.0f;
    // The filters go STRAIGHT from L2 into this wave's registers: a wave owns its three xi, so no other wave ever reads its filter
    // fragments, and (pack_weights_wino4_host) lane l31 finds channels l31 and l31 + 32 of (xi, k) as one 8-byte word -- six 8-byte
    // loads per slice and wave.  (Through LDS -- DMA in, fragment reads out -- they were 72 of the 137 KB of LDS traffic per slice,
    // and LDS-pipe time is added to the slice like vector-ALU time.)
    // ONE register set (round 6): group g's fragment of the NEXT slice is requested right behind the two MFMAs that consumed it, into
    // the same registers -- a whole slice of MFMAs before its next use.  (Round 5 declared two sets, "a slice ahead into the other
    // set"; at 128 registers hipcc folded them into one and sank all six requests to the END of the slice, in front of the barrier:
    // the next slice's first MFMA then waited for an L2 round trip (`s_waitcnt vmcnt(5)` right behind the barrier), on all three
    // waves of a SIMD at once.  The order is pinned with sched_group_barrier below.)
    f32x2_w ureg[6];
    const unsigned u_voff = (unsigned)((((3 * wave) * kW4KS + kh) * kWinoBN + l31 * 2) * 4);
#define RPN_W4_LOAD_U1(SLICE, G)                                                                                            \
    ureg[G] = __builtin_bit_cast(f32x2_w, __builtin_amdgcn_raw_buffer_load_b64(                                             \
        urs, u_voff + (unsigned)(((((G) % 3) * kW4KS + 2 * ((G) / 3)) * kWinoBN) * 4), (SLICE) * kW4UFloats * 4, 0))
    // Slice s's six operand values (one per group g = (k pair, xi)) are read during slice s - 1, in front of the barrier that ends it
    // (V triple-buffered, transformed two slices ahead): xi j's two operands right behind its second group's MFMAs, into the same
    // registers -- so a slice's first MFMA issues right behind the barrier.  Order pinned per group: 2 MFMAs, the filter request,
    // (groups 3 .. 5) the operand reads.
    const int v_lane = kh * 32 + l31;
    float av[6];
#pragma unroll
    for (int g = 0; g < 6; ++g) RPN_W4_LOAD_U1(0, g);
    __syncthreads();                                                       // (A) (B) (C): see w4_staging_role
    __syncthreads();
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 6; ++g) av[g] = ldsVR[((3 * wave + (g % 3)) * kW4KS + 2 * (g / 3)) * 32 + v_lane];
    int vcur = 0;
    for (int s = 0; s < n_slices; ++s) {
        const int nxt = s + 1 < n_slices ? s + 1 : s;                      // (past the end: the last slice again, unused)
        vcur = vcur + kW4VFloats == 3 * kW4VFloats ? 0 : vcur + kW4VFloats;   // V buffer of slice s + 1 (past the end: stale, unused)
        const float *Vn = ldsVR + vcur;
#pragma unroll
        for (int g = 0; g < 6; ++g) {                                      // group g = (k pair g / 3, xi j = g % 3)
            const int j = g % 3;
            acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g], ureg[g][0], acc[j][0], 0, 0, 0);
            acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g], ureg[g][1], acc[j][1], 0, 0, 0);
            RPN_W4_LOAD_U1(nxt, g);
            if (g >= 3) {            // xi j's two operands (k pairs 0, 1: one ds_read2) once both of its groups have issued
                av[g - 3] = Vn[((3 * wave + j) * kW4KS) * 32 + v_lane];
                av[g] = Vn[((3 * wave + j) * kW4KS + 2) * 32 + v_lane];
            }
        }
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if (g >= 3) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_s_barrier();                                      // (bare: see conv3x3_wino4n_f32_kernel)
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // (the operand reads past the end: landed before LDS is re-used)
#undef RPN_W4_LOAD_U1

    // ---- epilogue: park the 32 channels of N block ph ([xi 36][tile 32][32 channels] = 144 KB over the whole LDS block: the slice
    // loop's buffers are dead), then every thread transforms one (tile, channel) pair; the accumulators of the other N block stay in
    // registers meanwhile (48 + the transform's ~60: inside the 128 a sixteen-wave workgroup may have).  (First version: four rounds
    // of 16 channels transformed by the 256 staging threads alone: 32 k cycles per workgroup, 10 % of a 256-channel layer.)
    int lane_e;                                                           // (opaque again: see lane_t)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int tid_e = wave * 64 + lane_e, khe = lane_e >> 5, l31e = lane_e & 31;
    float *park = wave < 6 ? ldsU : ldsVR - kW4Half;                       // xi = 3 wave + j: waves 0 .. 5 own xi 0 .. 17
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        split ? a.part + ((size_t)tile_id * 2 + khalf) * (512 * kWinoBN) : nullptr, (short)0, split ? 512 * kWinoBN * 4 : 0, 0x00020000);
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int xi = 3 * wave + j;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * khe;           // tile
                park[(xi * 32 + row) * 32 + l31e] = acc[j][ph][e];
            }
        }
        __syncthreads();
        w4_output_pair(a, ldsU, ldsVR, tid_e, ph, img, oy0, ox0, n0, split, prs, w4_out_rsrc(a, img));
        __syncthreads();
    }
    if (a.ksplit == 2) w4_ksplit_finish(a, ldsU, tid_e, tile_id, img, oy0, ox0, n0);
    }   // tile loop
}

// =================================================================================================================================
// F(4x4, 3x3), WIDE form: 4 x 4 tiles = 16 x 16 output pixels x 128 output channels per workgroup, on v_mfma_f32_16x16x4_f32 (M = 16
// tiles, N = 16 channels, K = the slice's 4 input channels; the same 64 flops per cycle as 32x32x2).  The same accumulator footprint
// as the 32-tile x 64-channel form (36 xi x 16 x 128 floats = 288 KB of registers) but HALF the staging per MFMA: the input transform
// of a tile serves 128 output channels instead of 64 -- the patch (18 x 18 x 4 instead of 18 x 34 x 4), the transform's vector and
// LDS work and V (9 KB per slice) all halve, and only the filter fragments double (straight from L2 into the owning wave's
// registers: 24 per slice and wave, two 16-byte loads per xi).  LDS-pipe and vector-ALU time are ADDED to the float32 MFMA time on
// this chip, so this is the lever.  Layers with Cout % 128 == 0 whose 16 x 16-pixel tiles fill the chip (wino_variant 16).
//   U packing: [n_tile][slice][xi][k 4][c 16][nb 8]: lane (k = lane >> 4, c = lane & 15) holds channel nb * 16 + c of its k for the
//   eight N blocks as 32 contiguous bytes.   V: [xi][k][16 tiles].   raw: [channel][18][20] + padding, rows 8 .. 15 skewed (the
//   layout comment at kWnPlane).
constexpr int kWnBN = 128, kWnNT = 16, kWnPatch = 18, kWnPitch = 20, kWnPlane = kWnPatch * kWnPitch + 56;
constexpr int kWnUFloats = kW4Xi * kW4KS * kWnBN;                        // 18432 floats = 72 KB per slice and N tile
constexpr int kWnVFloats = kW4Xi * kW4KS * kWnNT;                        // 2304 floats = 9 KB
// raw-patch layout (round 6, checked exhaustively over (row, column pair) for both read forms hipcc emits): a staging lane = (channel
// lane >> 4, tile lane & 15) reads column PAIRS of its 6 x 6 patch.  As ds_read2_b64 (two adjacent pairs, 8-byte aligned: what the
// vectoriser makes of most of them) the sixteen tiles of a channel are one bank group over 32 dwords: tile columns are 4 floats apart,
// tile rows 80 = 16 (mod 32), so tile rows 0 / 2 and 1 / 3 met on the same banks (2-way conflicts on every read: the 37 % conflict
// share of profiles/r06_f32w_pmc.txt) -- patch rows 8 .. 15 are stored two floats to the right (tile rows land on 0, 16, 2, 18).
// As ds_read_b64 two channels are one group over 64 dwords: the plane stride is 32 (mod 64).
static_assert(kWnPlane % 64 == 32 && kWnPitch >= kWnPatch + 2, "raw plane stride / skew room");

size_t wino4n_weight_floats(int Cin, int Cout) { return (size_t)((Cout + kWnBN - 1) / kWnBN) * (Cin / kW4KS) * kWnUFloats; }

void pack_weights_wino4n_host(const float *hwio, const float *scale, int Cin, int Cout, float *dst)
{
    static const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int n_tiles = (Cout + kWnBN - 1) / kWnBN, n_slices = Cin / kW4KS;
    for (int nt = 0; nt < n_tiles; ++nt)
        for (int s = 0; s < n_slices; ++s)
            for (int k = 0; k < kW4KS; ++k)
                for (int nn = 0; nn < kWnBN; ++nn) {
                    const int c = s * kW4KS + k, n = nt * kWnBN + nn;
                    double g[3][3];
                    for (int r = 0; r < 3; ++r)
                        for (int q = 0; q < 3; ++q)
                            g[r][q] = n < Cout ? (double)hwio[((size_t)(r * 3 + q) * Cin + c) * Cout + n] * (scale ? (double)scale[n] : 1.0) : 0.0;
                    double t[6][3];
                    for (int i = 0; i < 6; ++i)
                        for (int q = 0; q < 3; ++q) t[i][q] = G[i][0] * g[0][q] + G[i][1] * g[1][q] + G[i][2] * g[2][q];
                    for (int i = 0; i < 6; ++i)
                        for (int j = 0; j < 6; ++j) {
                            const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                            dst[(((size_t)nt * n_slices + s) * kW4Xi + (6 * i + j)) * (kW4KS * kWnBN) + (k * 16 + (nn & 15)) * 8 + (nn >> 4)] = (float)u;
                        }
                }
}

// output transform of one (tile, channel) pair of an epilogue phase of the wide form: thread -> (channel = tid & 63 of the phase's
// 64, tile = tid >> 6: one tile per WAVE); MsA / MsB = the parked accumulators [xi][16 tiles][64 channels] of xi 0 .. 17 / 18 .. 35;
// one pass over the 36 values (see w4_output_pair)
__device__ __forceinline__ void w4n_output_pair(const WinoArgs &a, const float *MsA, const float *MsB, int tid, int ph, int img, int oy0, int ox0, int n0,
                                                __amdgpu_buffer_rsrc_t out_rs)
{
    const int e_n = tid & 63, e_tile = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e_ty = e_tile >> 2, e_tx = e_tile & 3;
    const int n = n0 + ph * 64 + e_n;
    const float bias = (a.bias && n < a.Cout) ? a.bias[n] : 0.0f;
    const int oy = oy0 + 4 * e_ty, ox = ox0 + 4 * e_tx;
    if (n >= a.Cout) return;
    float R[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float m[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) m[i] = (i < 3 ? MsA : MsB)[((6 * (i % 3) + j) * kWnNT + e_tile) * 64 + (e_n ^ ((e_tile >> 2) * 16))];     // xi = 6 i + j (segment swizzle: see the parking stores)
        const float pp = m[1] + m[2], qq = m[1] - m[2], rr = m[3] + m[4], ss = m[3] - m[4];
        R[0][j] = m[0] + pp + rr;
        R[1][j] = __builtin_fmaf(2.0f, ss, qq);
        R[2][j] = __builtin_fmaf(4.0f, rr, pp);
        R[3][j] = __builtin_fmaf(8.0f, ss, qq) + m[5];
    }
    float y[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p) w4_at(R[p], y[p]);
    const int pool = w4_pool(a);
    const bool interior = pool ? ((oy >> 1) + 2 <= (a.H >> 1) && (ox >> 1) + 2 <= (a.W >> 1)) : (oy + 4 <= a.H && ox + 4 <= a.W);   // wave-uniform
    const unsigned voff = pool ? (unsigned)((((oy >> 1) * (a.W >> 1) + (ox >> 1)) * a.Cout + n) * 4) : (unsigned)(((oy * a.W + ox) * a.Cout + n) * 4);
    w4_store_outputs(a, y, out_rs, voff, oy, ox, bias, interior);
}

// The staging role of the wide form (waves 12 .. 15, one per SIMD), one instantiation per wave: PART / HALF select the wave's QUARTER of
// every (tile, channel) pair's 6 x 6 input transform (output rows 3 PART .. 3 PART + 2, output columns 3 HALF .. 3 HALF + 2).  As
// run-time (wave-uniform) values they compiled into a branch around every 1-D transform with the patch reads issued two at a time
// and waited for at once (`s_waitcnt lgkmcnt(1)`, `lgkmcnt(0)` eighteen times per slice): a staging wave's slice was a chain of
// ~18 exposed LDS round trips.  As template parameters the slice is straight-line: 18 reads in flight, one wait.
template <int PART, int CHALF>
__device__ __forceinline__ void w4n_staging_role(const WinoArgs &a, float *ldsA, float *ldsVR, int tid, int lane, int t_first, int t_step,
                                                 int t_total, int tiles_x, int tiles_y, int n_tiles, int n_slices)
{
    // Round 6, second form: the staging waves work in PAIRS of slices (8 input channels): a wave owns output rows 3 PART .. 3 PART + 2
    // of the transforms of FOUR channels (4 CHALF .. + 3 = slice 2 P + CHALF of pair P) -- the column stage (36 operations) is then
    // shared by all six output columns instead of being repeated by the two waves that split them: 72 vector instructions per pair
    // and wave instead of 2 x 58 -- and the workgroup meets at a barrier once per pair.  V is a ring of six slice buffers (three
    // pairs), the raw patch (8 planes per buffer, two buffers) lives in ldsA, which the slice loop does not use otherwise.
    constexpr unsigned kOob = 0x80000000u;
    constexpr int kRawBuf = 2 * kW4KS * kWnPlane;            // floats per raw buffer (8 channel planes)
    constexpr int kItems = 2 * kWnPatch * kWnPatch;          // (pixel, channel half) items of a pair's patch: 648 16-byte loads
    static_assert(2 * kRawBuf <= 18 * kWnNT * 64 && 6 * kWnVFloats <= 18 * kWnNT * 64, "LDS layout");
    const int hid = tid - 768;             // staging thread 0 .. 255
    __builtin_amdgcn_s_setprio(3);                 // (see conv3x3_wino4_f32_kernel)
    unsigned r_off[3];
    int r_lds[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int item = hid + j * 256;
        const int half = item >= kWnPatch * kWnPatch ? 1 : 0, pix = item - half * (kWnPatch * kWnPatch);
        const int py = pix / kWnPatch, px = pix - py * kWnPatch;
        r_lds[j] = (4 * half) * kWnPlane + py * kWnPitch + px + ((py >> 3) & 1) * 2;      // (rows 8 .. 15 skewed by two floats: see kWnPlane)
    }
    auto set_offsets = [&](const WnTile &tl) {                    // this thread's source offsets inside the tile's image
        int h = hid;                                              // (opaque: the item's row / column are recomputed per tile, not kept in
        asm volatile("" : "+v"(h));                               // nine registers across the tile loop)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int item = h + j * 256;
            const int half = item >= kWnPatch * kWnPatch ? 1 : 0, pix = item - half * (kWnPatch * kWnPatch);
            const int py = pix / kWnPatch, px = pix - py * kWnPatch;
            const int iy = tl.oy0 - 1 + py, ix = tl.ox0 - 1 + px;
            const bool v = item < kItems && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            r_off[j] = v ? (unsigned)(((iy * a.W + ix) * a.Cin) * 4 + 16 * half) : kOob;
        }
    };
    const bool third = hid + 512 < kItems;
    const int t_tile = lane & 15, t_cl = lane >> 4;               // transform role: tile, channel 4 CHALF + t_cl of the pair
    const int t_ty = t_tile >> 2, t_tx = t_tile & 3;
    u32x4_w rr[2][3], p2[3], p3[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) rr[0][j] = rr[1][j] = p2[j] = p3[j] = u32x4_w{0u, 0u, 0u, 0u};
#define RPN_WN_LOAD_PAIR(DST, PAIR)                                                                                         \
    _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_)                                                                        \
        DST[j_] = __builtin_amdgcn_raw_buffer_load_b128(xrs, r_off[j_], (PAIR) * (2 * kW4KS * 4), 0)   /* (no item: kOob, zeros) */
#define RPN_WN_RAW_TO_LDS(BUF, SRC)                                                                                         \
    {                                                                                                                       \
        float *raw_ = ldsA + (BUF) * kRawBuf;                                                                               \
        _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_)                                                                    \
            if (j_ < 2 || third) {                                                                                          \
                const f32x4_w v_ = __builtin_bit_cast(f32x4_w, SRC[j_]);                                                    \
                _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) raw_[r_lds[j_] + i_ * kWnPlane] = v_[i_];                  \
            }                                                                                                               \
    }
    auto transform = [&](int buf, int vofs) {                // raws[buf] -> the pair's two V buffers at float offset vofs (slice CHALF)
        // (indices in float PAIRS: a column pair is one 8-byte read; a tile's patch rows 0 .. 3 carry the skew of patch row 4 ty, its
        // rows 4, 5 that of row 4 ty + 4: second base -- see kWnPlane)
        static_assert(kRawBuf % 2 == 0 && kWnPlane % 2 == 0 && kWnPitch % 2 == 0, "pair index");
        int ro2 = (buf * kRawBuf + (4 * CHALF + t_cl) * kWnPlane + (4 * t_ty) * kWnPitch + 4 * t_tx) / 2 + ((t_ty >> 1) & 1);
        int ro45 = ro2 - ((t_ty >> 1) & 1) + (((t_ty + 1) >> 1) & 1);
        asm volatile("" : "+v"(ro2), "+v"(ro45));
        const f32x2_w *rp2 = reinterpret_cast<const f32x2_w *>(ldsA) + ro2;
        const f32x2_w *rp45 = reinterpret_cast<const f32x2_w *>(ldsA) + ro45;
        float T[3][6];
#pragma unroll
        for (int jp = 0; jp < 3; ++jp) {
            float d0[6], d1[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const f32x2_w dd = (r < 4 ? rp2 : rp45)[(r * kWnPitch) / 2 + jp];
                d0[r] = dd[0];
                d1[r] = dd[1];
            }
            float t0[3], t1[3];
            w4_bt3<PART>(d0, t0);
            w4_bt3<PART>(d1, t1);
#pragma unroll
            for (int i = 0; i < 3; ++i) { T[i][2 * jp] = t0[i]; T[i][2 * jp + 1] = t1[i]; }
        }
        int vo = vofs + CHALF * kWnVFloats + (18 * PART) * (kW4KS * kWnNT) + t_cl * kWnNT + t_tile;
        asm volatile("" : "+v"(vo));
        float *vp = ldsVR + vo;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float v[6];
            w4_bt6(T[i], v);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
#ifdef RPN_EXP_WN_F16RATE    /* timing experiment (wrong results): what a transformed value would cost as hi + lo float16 halves */
                const _Float16 h_ = (_Float16)v[j];
                const _Float16 l_ = (_Float16)(v[j] - (float)h_);
                v[j] = __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, h_) | ((unsigned)__builtin_bit_cast(unsigned short, l_) << 16));
#endif
                vp[(6 * i + j) * (kW4KS * kWnNT)] = v[j];
            }
        }
    };
    // Pipeline: pair P is requested in iteration P - 4 (register set P & 1), written to raws[P & 1] in iteration P - 3, transformed into
    // the V ring's pair buffer P % 3 in iteration P - 2, read by the MFMA waves from iteration P - 1 on, multiplied in iteration P.
    // No conditions in the loop body (requests clamped to the last pair, the tail works on stale data that nobody reads).
    const int n_pairs = n_slices >> 1, last = n_pairs - 1;
    // the first two pairs of a tile: requested here for the workgroup's first tile, in front of the previous tile's epilogue for
    // every other one (below; all four pairs held across the epilogue were 48 registers beside its ~75: spills); pairs 2, 3 at the
    // top of the tile (they are needed behind barrier (B))
#define RPN_WN_REQUEST_TILE()                                                                                               \
{                                                                                                                       \
    RPN_WN_LOAD_PAIR(rr[0], 0);                                                                                         \
    RPN_WN_LOAD_PAIR(rr[1], 1 < last ? 1 : last);                                                                       \
}
    WnTile tl = wn_tile(t_first, tiles_x, tiles_y, n_tiles);
    const WnTile st = wn_tile(t_step, tiles_x, tiles_y, n_tiles);          // digits of the step (wn_advance)
    __amdgpu_buffer_rsrc_t xrs = wn_image_rsrc(a, tl.img);
    set_offsets(tl);
    RPN_WN_REQUEST_TILE();
    for (int t = t_first; t < t_total; t += t_step) {
    const int img = tl.img, oy0 = tl.oy0, ox0 = tl.ox0, n0 = tl.n0;
    RPN_WN_LOAD_PAIR(p2, 2 < last ? 2 : last);
    RPN_WN_LOAD_PAIR(p3, 3 < last ? 3 : last);
    RPN_WN_RAW_TO_LDS(0, rr[0]);
    RPN_WN_RAW_TO_LDS(1, rr[1]);
    __syncthreads();                                                       // (A)
    transform(0, 0);
    transform(1, 2 * kWnVFloats);
    __syncthreads();                                                       // (B)
#pragma unroll
    for (int j = 0; j < 3; ++j) rr[0][j] = p2[j];
    RPN_WN_RAW_TO_LDS(0, rr[0]);
#pragma unroll
    for (int j = 0; j < 3; ++j) rr[1][j] = p3[j];
    __syncthreads();                                                       // (C): pairs 0, 1 in the V ring, pair 2 in raws[0], pair 3 in register set 1
#ifdef RPN_EXP_WN_NOSTAGE    /* timing experiment (wrong results): the staging waves only keep the barriers */
#define RPN_WN_STAGE(P, SET) __syncthreads();
#else
#define RPN_WN_STAGE(P, SET)                                                                                                \
{                                                                                                                       \
    RPN_WN_LOAD_PAIR(rr[SET], (P) + 4 < last ? (P) + 4 : last);                                                         \
    transform(SET, vnext);                                           /* pair P + 2: raws[P & 1] -> ring buffer (P + 2) % 3 */ \
    RPN_WN_RAW_TO_LDS((SET) ^ 1, rr[(SET) ^ 1]);                     /* pair P + 3 */                                   \
    vnext = vnext + 2 * kWnVFloats == 6 * kWnVFloats ? 0 : vnext + 2 * kWnVFloats;                                      \
    RPN_WN_STAMP((P) < 14 ? 4 * (P) + 2 : 99);                                                                          \
    __syncthreads();                                                                                                    \
    RPN_WN_STAMP((P) < 14 ? 4 * (P) + 3 : 99);                                                                          \
}
#endif
    {
        int vnext = 4 * kWnVFloats;                                  // ring buffer of pair P + 2
        (void)vnext;
        int P = 0;
        for (; P + 1 < n_pairs; P += 2) {
            RPN_WN_STAGE(P, 0);
            RPN_WN_STAGE(P + 1, 1);
        }
        if (P < n_pairs) RPN_WN_STAGE(P, 0);
    }
    // the next tile's first four pairs: in flight across this tile's epilogue (twelve 16-byte registers per thread)
    if (t + t_step < t_total) {
        wn_advance(tl, st, tiles_x, tiles_y, n_tiles);
        xrs = wn_image_rsrc(a, tl.img);
        set_offsets(tl);
        RPN_WN_REQUEST_TILE();
    }
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        __syncthreads();                                         // the phase's accumulators are in LDS
        w4n_output_pair(a, ldsA, ldsVR, tid, ph, img, oy0, ox0, n0, w4_out_rsrc(a, img));
        __syncthreads();                                         // the staging area may be overwritten
    }
    }   // tile loop
#undef RPN_WN_STAGE
#undef RPN_WN_REQUEST_TILE
#undef RPN_WN_LOAD_PAIR
#undef RPN_WN_RAW_TO_LDS
}


__global__ void __launch_bounds__(kW4Threads, 1)
conv3x3_wino4n_f32_kernel(WinoArgs a, int tiles_x, int tiles_y, int n_tiles)
{
    // two arrays of 72 KB: the slice loop uses ldsVR only (V x 2: 18 KB, raw x 2: 11.6 KB); the epilogue parks the 36 x 16 x 64
    // accumulators of a phase (144 KB) over both: xi 0 .. 17 in ldsA, xi 18 .. 35 in ldsVR
    constexpr int kHalf = 18 * kWnNT * 64;                                 // floats per array
    static_assert(6 * kWnVFloats <= kHalf && 4 * kW4KS * kWnPlane <= kHalf, "LDS layout");
    __shared__ __attribute__((aligned(16))) float ldsA[kHalf];
    __shared__ __attribute__((aligned(16))) float ldsVR[kHalf];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // persistent workgroups: tiles t_first, t_first + gridDim.x, ... (see wn_tile); both roles walk the same tiles with the same
    // number of barriers per tile
    const int t_first = wino_xcd_remap(blockIdx.x, gridDim.x), t_step = gridDim.x;
    const int t_total = tiles_x * tiles_y * a.B * n_tiles;
    const int n_slices = a.Cin / kW4KS;
    const long long ubytes = (long long)n_slices * kWnUFloats * 4;

    const bool stager = wave >= 12;

    if (stager) {
        const int sw = wave - 12;          // t_part = sw & 1, t_half = sw >> 1
        if (sw == 0) w4n_staging_role<0, 0>(a, ldsA, ldsVR, tid, lane, t_first, t_step, t_total, tiles_x, tiles_y, n_tiles, n_slices);
        else if (sw == 1) w4n_staging_role<1, 0>(a, ldsA, ldsVR, tid, lane, t_first, t_step, t_total, tiles_x, tiles_y, n_tiles, n_slices);
        else if (sw == 2) w4n_staging_role<0, 1>(a, ldsA, ldsVR, tid, lane, t_first, t_step, t_total, tiles_x, tiles_y, n_tiles, n_slices);
        else w4n_staging_role<1, 1>(a, ldsA, ldsVR, tid, lane, t_first, t_step, t_total, tiles_x, tiles_y, n_tiles, n_slices);
        return;
    }

    // ---- MFMA waves: xi = 3 wave + j, 16 tiles x 128 channels = eight 16 x 16 blocks per xi
    WnTile tl = wn_tile(t_first, tiles_x, tiles_y, n_tiles);
    const WnTile tstep = wn_tile(t_step, tiles_x, tiles_y, n_tiles);       // (the walk is incremental: wn_advance)
    for (int t = t_first; t < t_total; t += t_step) {
    const int img = tl.img, oy0 = tl.oy0, ox0 = tl.ox0, n0 = tl.n0;
    // (the lane index re-derived per tile, opaque: what is computed from it is otherwise hoisted out of the tile loop and -- the
    // slice loop runs at exactly 128 registers -- spilled to scratch around it)
    int lane_t;                          // (volatile asm: the builtin's value is loop-invariant too and would be hoisted and kept)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
    const int k4 = lane_t >> 4, l15 = lane_t & 15;
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.u) + (size_t)tl.nt * n_slices * kWnUFloats, (short)0,
                                                                          (int)(ubytes > 0x7fffffffll ? 0x7fffffff : ubytes), 0x00020000);
    f32x4_w acc[3][8];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[j][nb] = f32x4_w{0.f, 0.f, 0.f, 0.f};
    // filter fragments: ONE register set; xi j's fragments of the next slice are requested right behind its MFMAs of this slice (a
    // whole slice of latency budget; two sets would be 48 registers beside the 96 accumulators)
    f32x4_w ureg[3][2];
    const unsigned u_voff = (unsigned)(((3 * wave) * (kW4KS * kWnBN) + lane_t * 8) * 4);
#define RPN_WN_LOAD_U(SLICE, J)                                                                                             \
    {                                                                                                                       \
        /* (xi j's 2 KB step rides in the SCALAR offset: as a vector-offset constant beyond 4095 it cost an address register) */                                               \
        ureg[J][0] = __builtin_bit_cast(f32x4_w, __builtin_amdgcn_raw_buffer_load_b128(urs, u_voff, (SLICE) * kWnUFloats * 4 + (J) * (kW4KS * kWnBN) * 4, 0));        \
        ureg[J][1] = __builtin_bit_cast(f32x4_w, __builtin_amdgcn_raw_buffer_load_b128(urs, u_voff, (SLICE) * kWnUFloats * 4 + (J) * (kW4KS * kWnBN) * 4 + 16, 0));  \
    }
    RPN_WN_STAMP(60);
    RPN_WN_STAMP_RT(58);
    RPN_WN_LOAD_U(0, 0);
    RPN_WN_LOAD_U(0, 1);
    RPN_WN_LOAD_U(0, 2);
    __syncthreads();                                                       // (A) (B) (C): see w4n_staging_role
    __syncthreads();
    __syncthreads();
    RPN_WN_STAMP(61);
    // Slice s's operands (3 values of V per lane, one per xi) are read during slice s - 1, in front of the barrier that ends it (V is
    // triple-buffered and transformed two slices ahead): xi j's operand right behind xi j's MFMAs, INTO THE SAME REGISTER -- so the
    // first MFMA of a slice issues right behind the barrier (its operand was read two thirds of a slice earlier; the last xi's
    // read, issued in front of the barrier, has the next slice's first sixteen MFMAs to land).  Per slice the order is pinned: per
    // xi its eight MFMAs, the next slice's two filter requests of THAT xi, its operand read.  (Left alone hipcc mixes the three xi
    // and sinks the requests of two of them to the end of the slice, in front of the barrier, and the next slice then waits for
    // them -- `s_waitcnt vmcnt(0)` behind its third MFMA: an L2 round trip exposed per slice on all three waves of a SIMD at once.)
    const int v_lane = k4 * kWnNT + l15;
    float av[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) av[j] = ldsVR[((3 * wave + j) * kW4KS) * kWnNT + v_lane];
    int vcur = 0;
    for (int s = 0; s < n_slices; ++s) {
#ifdef RPN_EXP_WN_U0         /* timing experiment (wrong results): every filter request hits slice 0 */
        const int nxt = 0;
#else
        const int nxt = s + 1 < n_slices ? s + 1 : s;                 // (past the end: the last slice again, unused)
#endif
        vcur = vcur + kWnVFloats == 6 * kWnVFloats ? 0 : vcur + kWnVFloats;       // ring buffer of slice s + 1 (past the end: stale, unused)
        const float *Vn = ldsVR + vcur;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#ifdef RPN_EXP_WN_F16RATE    /* timing experiment (wrong results): the slice's GEMM at the f16x3 rate -- 24 x 2048 flop = 3 products x 3 MFMAs of 16x16x32 */
            using f16x8_e = __attribute__((ext_vector_type(8))) _Float16;
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
                acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_e, ureg[j][1]), __builtin_bit_cast(f16x8_e, ureg[j][0]), acc[j][nb], 0, 0, 0);
            acc[j][3][0] += av[j];
#else
#pragma unroll
            for (int nb = 0; nb < 8; ++nb)
                acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], ureg[j][nb >> 2][nb & 3], acc[j][nb], 0, 0, 0);
#endif
            if (j == 0) { RPN_WN_LOAD_U(nxt, 0); } else if (j == 1) { RPN_WN_LOAD_U(nxt, 1); } else { RPN_WN_LOAD_U(nxt, 2); }
            av[j] = Vn[((3 * wave + j) * kW4KS) * kWnNT + v_lane];
        }
#if !defined(RPN_EXP_WN_NOSCHED) && !defined(RPN_EXP_WN_F16RATE)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#endif
        // ONE barrier per PAIR of slices (the staging waves work in pairs: w4n_staging_role), behind the pair's second slice.  A bare
        // barrier: this role writes no LDS in the loop, and its last operand read must NOT be waited for here (__syncthreads' fence
        // would: `s_waitcnt lgkmcnt(0)`) -- the MFMAs that consume it sit in front of the next barrier, which is what keeps the
        // staging waves from overwriting that ring buffer (two pairs later) before it is read
        if (s & 1) {
            RPN_WN_STAMP(s < 28 ? 2 * s : 99);
            __builtin_amdgcn_s_barrier();
            RPN_WN_STAMP(s < 28 ? 2 * s + 1 : 99);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // (the operand reads past the end: landed before LDS is re-used)
    RPN_WN_STAMP(62);
#undef RPN_WN_LOAD_U

    // ---- epilogue: two phases of 64 channels: park [xi 36][tile 16][64 channels] = 144 KB over both arrays, every thread transforms
    // one (tile, channel) pair; C / D of the 16x16 MFMA: column (channel) = lane & 15, row (tile) = 4 * (lane >> 4) + register
    // (the lane index re-derived here, opaque: everything the epilogue computes from the thread index is loop-invariant in the
    // tile loop, gets hoisted in front of it and -- the slice loop runs at exactly 128 registers -- was spilled to scratch)
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int tid_e = wave * 64 + lane_e, k4e = lane_e >> 4, l15e = lane_e & 15;
    float *park = wave < 6 ? ldsA : ldsVR - kHalf;                        // xi = 3 wave + j: waves 0 .. 5 own xi 0 .. 17
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int xi = 3 * wave + j;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r)      // (16-channel segment XOR tile-row group: the four lane groups of a store are four tile rows
                                                 //  = 256 floats apart, the same banks without it; the reader undoes it with a wave-uniform XOR)
                    park[(xi * kWnNT + 4 * k4e + r) * 64 + ((q ^ k4e) * 16) + l15e] = acc[j][4 * ph + q][r];
        }
        __syncthreads();
        w4n_output_pair(a, ldsA, ldsVR, tid_e, ph, img, oy0, ox0, n0, w4_out_rsrc(a, img));
        __syncthreads();
    }
    RPN_WN_STAMP(63);
    RPN_WN_STAMP_RT(59);
    wn_advance(tl, tstep, tiles_x, tiles_y, n_tiles);
    }   // tile loop
}
#undef RPN_WINO_LDS_PTR

// Which Winograd form a layer runs on (per model handle, from the grid at its largest batch -- never per call: the forms sum in
// different orders).  Returns 16 (wide F(4x4, 3x3): 16 x 16 pixels x 128 channels per workgroup, where Cout % 128 == 0 and those tiles
// give every CU a workgroup), 4 (F(4x4, 3x3), 16 x 32 pixels x 64 channels, where THOSE tiles fill the chip), 8 (the same with the
// input channels split over two workgroups per tile, where they fill half of it: the 31 x 31 layers at batch 8) or 2 (F(2x2, 3x3),
// 16 x 16 pixels x 64 channels: small grids).  The CU count is that of the device CURRENT at the call (rpn_model_create /
// rpn_conv2d run on the caller's device), cached per device id.
static int wino_cus_of_current_device()
{
    static int cache[64] = {0};                          // (0: not asked yet; benign race: every thread stores the same value)
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 256;
    if (dev < 64 && cache[dev] > 0) return cache[dev];
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    if (dev < 64) cache[dev] = n;
    return n;
}
int wino_variant(int B, int H, int W, int Cin, int Cout)
{
    static const int knob = RPN_LAB_KNOB("RPN_WINO_F", 0);          // (laboratory: 2 | 4 | 8 | 16 forces a form)
    const int n_cus = wino_cus_of_current_device();
    if (Cin % kW4KS != 0) return 2;
    if (knob == 2 || knob == 4 || knob == 8 || (knob == 16 && Cout % kWnBN == 0)) return knob;
#ifndef RPN_EXP_W4_NOWIDE
    if (Cout % kWnBN == 0 && knob != 1 &&
        (long long)((W + 15) / 16) * ((H + 15) / 16) * B * (Cout / kWnBN) >= n_cus)
        return 16;                                       // the wide form: 16 x 16 pixels x 128 channels per workgroup
#endif
    const long long wgs4 = (long long)((W + 4 * kW4TX - 1) / (4 * kW4TX)) * ((H + 4 * kW4TY - 1) / (4 * kW4TY)) * B * ((Cout + kWinoBN - 1) / kWinoBN);
    if (wgs4 >= n_cus) return 4;
    return 2 * wgs4 >= n_cus && Cin >= 64 && wgs4 <= kW4Tickets ? 8 : 2;       // 8: F(4x4, 3x3), the input channels split over two workgroups per tile
}

// device workspace of a variant-8 layer: kW4Tickets tickets (one per tile; zero-initialised ONCE by the owner, the kernel leaves every
// ticket at zero) at a FIXED place in front -- launches of different batch sizes share the buffer, and a ticket must never lie where
// another launch writes partial tiles --, then 2 partial tiles of 512 x 64 floats per tile
static long long wino4_tiles(int B, int H, int W, int Cout)
{
    return (long long)((W + 4 * kW4TX - 1) / (4 * kW4TX)) * ((H + 4 * kW4TY - 1) / (4 * kW4TY)) * B * ((Cout + kWinoBN - 1) / kWinoBN);
}
size_t wino_workspace_bytes(int B, int H, int W, int Cin, int Cout, int variant)
{
    (void)Cin;
    if (variant != 8) return 0;
    return (size_t)kW4Tickets * 4 + (size_t)wino4_tiles(B, H, W, Cout) * 2 * 512 * kWinoBN * 4;
}

// Everything launch_conv3x3_wino checks besides the pointers, as a predicate: a model handle asks it when the graph is BUILT (add_conv),
// so that a layer the launcher would refuse (an image beyond the kernels' 32-bit buffer offsets, a tile count beyond the ticket block)
// stays on the direct float32 kernel instead of failing at every forward.
bool wino_launchable(int B, int H, int W, int Cin, int Cout, int variant)
{
    if (!wino_supported(Cin, Cout) || B < 1 || H < 1 || W < 1 || (variant != 2 && variant != 4 && variant != 8 && variant != 16)) return false;
    if ((long long)H * W * Cin * 4 > 0x7fffffffll || (long long)H * W * Cout * 4 > 0x7fffffffll) return false;   // 32-bit buffer offsets per image, input and output
    if (variant == 8 && (Cin / kW4KS < 2 || wino4_tiles(B, H, W, Cout) > kW4Tickets)) return false;
    if (variant == 16 && Cout % kWnBN != 0) return false;
    return true;
}

hipError_t launch_conv3x3_wino(const float *x, const float *u, const float *bias, float *out, int B, int H, int W, int Cin,
                               int Cout, int act, bool pool, hipStream_t s, int variant, void *workspace)
{
    if (act == ACT_SIGMOID || !wino_launchable(B, H, W, Cin, Cout, variant)) return hipErrorInvalidValue;
    WinoArgs a{};
    a.x = x; a.u = u; a.bias = bias; a.out = out;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.act = act; a.pool = pool ? 1 : 0;
    a.ksplit = 1;
    if (variant == 8) {
        if (!workspace) return hipErrorInvalidValue;
        a.ksplit = 2;
        a.tickets = static_cast<unsigned *>(workspace);
        a.part = reinterpret_cast<float *>(static_cast<char *>(workspace) + (size_t)kW4Tickets * 4);
    }
    if (variant == 16) {
        if (Cout % kWnBN != 0) return hipErrorInvalidValue;
        const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16, n_tiles = Cout / kWnBN;
        const long long nblocks = (long long)tiles_x * tiles_y * B * n_tiles;
        if (nblocks > 0x7fffffffll) return hipErrorInvalidValue;
        // persistent: one workgroup per CU walks its tiles (wn_tile); fewer tiles than CUs: one tile each
        const long long grid = nblocks < (long long)wino_cus_of_current_device() ? nblocks : (long long)wino_cus_of_current_device();
        hipLaunchKernelGGL(conv3x3_wino4n_f32_kernel, dim3((unsigned)grid), dim3(kW4Threads), 0, s, a, tiles_x, tiles_y, n_tiles);
        return hipGetLastError();
    }
    if (variant == 4 || variant == 8) {
        const int tiles_x = (W + 4 * kW4TX - 1) / (4 * kW4TX), tiles_y = (H + 4 * kW4TY - 1) / (4 * kW4TY);
        const int n_tiles = (Cout + kWinoBN - 1) / kWinoBN;
        const long long nblocks = (long long)tiles_x * tiles_y * B * n_tiles * a.ksplit;
        if (nblocks > 0x7fffffffll) return hipErrorInvalidValue;
        // variant 4: persistent, one workgroup per CU walks its tiles (wn_tile); variant 8: two workgroups per tile, one tile each
        const long long cus = wino_cus_of_current_device();
        const long long grid = (a.ksplit == 2 || nblocks < cus) ? nblocks : cus;
        hipLaunchKernelGGL(conv3x3_wino4_f32_kernel, dim3((unsigned)grid), dim3(kW4Threads), 0, s, a, tiles_x, tiles_y, n_tiles);
        return hipGetLastError();
    }
    const int tiles_x = (W + 2 * kWinoTT - 1) / (2 * kWinoTT), tiles_y = (H + 2 * kWinoTT - 1) / (2 * kWinoTT);
    const int n_tiles = (Cout + kWinoBN - 1) / kWinoBN;
    const long long nblocks = (long long)tiles_x * tiles_y * B * n_tiles;
    if (nblocks > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(conv3x3_wino_f32_kernel, dim3((unsigned)nblocks), dim3(kWinoThreads), 0, s, a, tiles_x, tiles_y, n_tiles);
    return hipGetLastError();
}

}  // namespace rpn
