// mnv2_block_kernels.hip -- one launch per MobileNetV2 inverted-residual block (gfx950).
//
// Reference graph: models/rpn_mobilenet_v2.py:16-17 = keras-applications 1.0.8 MobileNetV2(alpha = 1) up to
// block_13_expand_relu.  One block = expand 1x1 (+BN, ReLU6) -> depthwise 3x3 (+BN, ReLU6; stride 1 'same' or
// stride 2 behind ZeroPadding2D(correct_pad)) -> project 1x1 (+BN, linear) [+ input when stride 1 and Cin == Cout].
// Unfused (conv_kernels.hip) that is three launches and the 6x-expanded tensor crosses HBM twice; at 500x500 the
// backbone then sits 4x off its memory roofline and every launch is a few tens of microseconds of mostly latency.
//
// Here a workgroup (256 threads) owns a 4 x 8 tile of the block's OUTPUT pixels of one image and keeps everything
// in between on the CU:
//   1. the input halo tile ((4-1)s+3) x ((8-1)s+3) pixels x Cin is loaded once into LDS (pixel-major, row stride
//      Cin + 2 floats = 2 * odd: the 16x16x4 A-fragment read -- 16 pixels x 2 k per half-wave -- is conflict-free);
//   2. per chunk of CE = 48 expanded channels:
//        expand   halo pixels x CE  = X (halo x Cin) * We        v_mfma_f32_16x16x4_f32 (exact f32: an ordered fmaf
//                 chain), + bias, ReLU6, ZEROED outside the image (the depthwise pads the EXPANDED tensor), written
//                 channel-major to LDS (row stride = 4 mod 32: conflict-free 16-byte writes from the C layout);
//        dwconv   32 pixels x CE on the vector ALU, lanes = pixels (conflict-free 4-byte LDS reads), weights of the
//                 chunk staged in LDS; + bias, ReLU6 -> LDS, pixel-major (A operand of the projection);
//        project  acc (32 px x Cout) += D (32 x CE) * Wp[chunk]   MFMA, accumulators stay in registers over chunks;
//   3. + bias (+ the residual, read from the input tile already in LDS) -> NHWC float32.
// The expanded tensor never exists in HBM.  Weight fragments (B operands) are read straight from global memory in
// the MFMA B layout (4 rows x 64 contiguous bytes per wave instruction; the whole weight set of a block is <= 0.5 MB
// and is shared by every workgroup, so it lives in L2 / L1).  The chunks are software-pipelined over two
// wave groups (see the kernel); 31 KB (Cin = 24) .. 95 KB (stride 2, Cin = 32) of LDS per workgroup.
//
// STEM variant: the "expand" stage is Conv1 (3x3 stride 2, 3 -> 32 channels, K = 27 padded to 28) computed from an
// im2col tile gathered from the image, followed by expanded_conv's depthwise and its 32 -> 16 projection: stem +
// block 0 in one launch.
//
// Roofline: MFMA-bound on the exact-f32 MFMA (157.3 TFLOP/s); algorithmic flops per output pixel =
// 2*Cin*Cexp/s^2... (expand runs on the block's INPUT pixels) + 18*Cexp + 2*Cexp*Cout; the halo makes the expand stage
// compute (6*10)/(4*8) = 1.9x (stride 1) / (9*17)/(8*16) = 1.2x its algorithmic work.
#include "conv_kernels.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// Debug build only (-DRPN_STAMP, scripts/ir_stamp_probe.py): cycle stamps of the pipeline's phases, 64 slots per
// workgroup for the first 512 workgroups: [0] start, [1] input tile in LDS, then per step it: [2+4it] wave 0 after E,
// [3+4it] wave 0 after P, [4+4it] wave 4 after D, [5+4it] wave 0 after the barrier.
#ifdef RPN_STAMP
__device__ unsigned long long g_ir_stamps[512 * 128];
#define IR_STAMP(w, k)                                                                                    \
    do {                                                                                                  \
        if (threadIdx.x == 64 * (w) && blockIdx.x < 512 && (k) < 128 && a.stamp)                          \
            g_ir_stamps[blockIdx.x * 128 + (k)] = __builtin_readcyclecounter();                            \
    } while (0)
#define IR_STAMP_RT(w, k)                                                                                 \
    do {                                                                                                  \
        if (threadIdx.x == 64 * (w) && blockIdx.x < 512 && (k) < 128 && a.stamp)                          \
            g_ir_stamps[blockIdx.x * 128 + (k)] = __builtin_amdgcn_s_memrealtime();                        \
    } while (0)
extern "C" int rpn_debug_read_ir_stamps(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ir_stamps), (size_t)n * 8);
}
extern "C" int rpn_debug_clear_ir_stamps(void)
{
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_ir_stamps)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(g_ir_stamps));
}
// ... and of ir_block_hrx3_kernel (scripts/hr_stamp_probe.py): per variant v (block 1 / 2 / 3), workgroups 512 .. 767, 40 slots:
// [0] start, [1] A operand built, per chunk c: [2+3c] E written (after the barrier), [3+3c] D written (after the barrier),
// [4+3c] P issued, [39] outputs stored
__device__ unsigned long long g_hr_stamps[3 * 256 * 40];
#define HR_STAMP(v, k)                                                                                    \
    do {                                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x >= 512 && blockIdx.x < 768 && (k) < 40)                       \
            g_hr_stamps[((v) * 256 + (blockIdx.x - 512)) * 40 + (k)] = __builtin_readcyclecounter();       \
    } while (0)
extern "C" int rpn_debug_read_hr_stamps(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hr_stamps), (size_t)n * 8);
}
#else
#define IR_STAMP(w, k) ((void)0)
#define IR_STAMP_RT(w, k) ((void)0)
#define HR_STAMP(v, k) ((void)0)
#endif

namespace rpn {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int IR_TH = 4, IR_TW = 8;           // output tile
constexpr int IR_THREADS = 512;               // waves 0-3: the two GEMMs (one wave per SIMD); waves 4-7: depthwise

struct IrArgs {
    const float *x;       // block input, NHWC float32 (STEM: the image, 3 channels)
    float *out;           // block output, NHWC float32
    const float *we, *be; // expand weights [KP][CEXP] (BN scale folded), bias [CEXP]
    const float *wd, *bd; // depthwise weights [9][CEXP] (folded), bias [CEXP]
    const float *wp, *bp; // projection weights [CEXP][COUTP] (folded, zero-padded columns), bias [COUTP]
    int B, H, W;          // input height / width (STEM: of the image)
    int DH, DW;           // grid the depthwise reads = expanded tensor (== H, W; STEM: Conv1's output size)
    int OH, OW;           // block output size
    int pad;              // top / left zero padding of the depthwise (STEM: of Conv1; its depthwise pads 1)
    int tiles_x, tiles_y;
    int stamp;            // -DRPN_STAMP builds: this launch records phase stamps
};

__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.0f), 6.0f); }

// KP: reduction length of the expand stage (Cin; STEM: 28); CE: expanded channels per chunk; S: depthwise stride.
//
// Software pipeline over the chunks, one workgroup barrier per step.  In step i
//     waves 0-3 (MFMA):     E(i)   expand chunk i        Xs, WeS[i & 1]          -> Es[i & 1]
//                           P(i-2) project chunk i-2     Ds[i & 1], WpS[i & 1]   -> accumulators
//     waves 4-7 (service):  D(i-1) depthwise chunk i-1   Es[(i-1)&1]             -> Ds[(i-1)&1]
//                           stage  We chunk i+1 -> WeS[(i+1)&1],  Wp chunk i-1 -> WpS[(i-1)&1]  (global -> regs -> LDS),
//                                  depthwise weights of chunk i -> registers (used by D(i) in step i+1)
// so every operand the matrix waves touch is already in LDS when a step starts (L2 latency is paid by the service
// waves, one step ahead, beside their depthwise work) and the depthwise runs beside the GEMMs of the neighbouring
// chunks.  Measured on the 32x32 blocks (one workgroup per CU): first version (4 waves, weights from L2 inside the
// k-loop, depthwise between the GEMMs) 24 % MFMA utilisation; see DESIGN.md for the current figures.
//
// LDS images (float32; strides chosen for the access that reads or writes them with 4-byte-per-lane instructions):
//   Xs  [halo px][KP + 2]      A operand of E: 16 px x 2 k per half-wave, stride 2 * odd          -> conflict-free
//   WeS [2][KP][SWE]           B operand of E: 2 k x 16 n per half-wave, stride = 16 mod 32       -> conflict-free
//   Es  [2][halo px][CE + 4]   written from the C layout (16 n x 2 row groups 4 px apart: stride = 4 mod 8), read by
//                              the depthwise with lanes = channels                                -> conflict-free
//   Ds  [2][32 px][CE + 2]     written with lanes = channels, read as the A operand of P (stride 2 * odd)
//   WpS [2][CE][SWP]           B operand of P, stride = 16 mod 32
template <int KP, int CEXP, int CE, int COUT, int S, bool RES, bool STEM>
__global__ void __launch_bounds__(IR_THREADS)
ir_block_kernel(IrArgs a)
{
    constexpr int SX = KP + 2;
    constexpr int IH = (IR_TH - 1) * S + 3, IW = (IR_TW - 1) * S + 3;
    constexpr int NH = IH * IW;                              // halo pixels
    constexpr int MB = (NH + 15) / 16;                       // M-blocks of the expand GEMM
    constexpr int MH = MB * 16;
    constexpr int MBW = (MB + 3) / 4;                        // M-blocks per wave
    constexpr int SEP = CE + 4;
    constexpr int NBE = CE / 16;                             // N-blocks of the expand GEMM per chunk
    constexpr int NCHUNK = CEXP / CE;
    constexpr int SD = CE + 2;
    constexpr int COUTP = (COUT + 15) / 16 * 16;
    constexpr int NBO = COUTP / 16;                          // N-blocks of the projection
    constexpr int NJ = (NBO + 1) / 2;                        // ... per wave (waves split 2 M-blocks x 2 N-block parities)
    constexpr int SWE = CE % 32 == 16 ? CE : CE + 16;
    constexpr int SWP = COUTP % 32 == 16 ? COUTP : COUTP + 16;
    constexpr int KS = KP / 4;                               // k-steps of the expand GEMM
    constexpr int EB = 4;                                    // k-steps per register batch of E operands
    constexpr int NEB = (KS + EB - 1) / EB;
    constexpr int PS = CE / 4;                               // k-steps of the projection per chunk
    constexpr int PB = PS % 6 == 0 ? 6 : 4;                  // k-steps per register batch of P operands
    constexpr int NPB = PS / PB;
    constexpr int WE4 = KP * CE / 4, NWE = (WE4 + 255) / 256;         // float4s of a We chunk, per service thread
    constexpr int WP4 = CE * COUTP / 4, NWP = (WP4 + 255) / 256;
    static_assert(KP % 4 == 0 && (SX / 2) % 2 == 1 && (SD / 2) % 2 == 1 && CEXP % CE == 0 && CE % 16 == 0 && CE <= 64, "layout");
    static_assert(SEP % 8 == 4 && SWE % 32 == 16 && SWP % 32 == 16 && PS % PB == 0, "layout");
    static_assert(!RES || (S == 1 && KP == COUT && !STEM), "residual needs stride 1 and Cin == Cout");

    // STEM: Xs is the raw image patch ((2(IH-1)+3) rows x (2(IW-1)+3) pixels x 3 channels, row stride PSTR) and the A
    // operand of Conv1 is read straight from it: im2col entry (pixel (hy, hx), k = (r*3+q)*3+c) = patch[(2hy + r)*PSTR +
    // (2hx + q)*3 + c] = base(pixel) + off(k) -- no im2col tile is ever built
    constexpr int PR = 2 * (IH - 1) + 3, PC = (2 * (IW - 1) + 3) * 3, PSTR = PC + 2;
    constexpr int XS_FLOATS = STEM ? PR * PSTR : MH * SX;
    __shared__ __attribute__((aligned(16))) float Xs[XS_FLOATS];
    __shared__ __attribute__((aligned(16))) float Es[2][MH * SEP];
    __shared__ __attribute__((aligned(16))) float Ds[2][IR_TH * IR_TW * SD];
    __shared__ __attribute__((aligned(16))) float WeS[2][KP * SWE];
    __shared__ __attribute__((aligned(16))) float WpS[2][CE * SWP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int stid = tid - 256;                              // service thread index (waves 4-7)
    int t = blockIdx.x;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int img = t / a.tiles_y;
    const int oy0 = ty * IR_TH, ox0 = tx * IR_TW;
    const int dpad = STEM ? 1 : a.pad;
    const int gy0 = oy0 * S - dpad, gx0 = ox0 * S - dpad;   // halo origin on the depthwise's input grid
    IR_STAMP(0, 0);
    // Issue arbitration between the two waves of a SIMD is by priority, then age: the service waves (4-7) are the younger
    // ones and would only get the slots the matrix waves leave (measured: their step started when the matrix waves' ended).
    if (wave >= 4) __builtin_amdgcn_s_setprio(2);

    // ---- service waves: weight staging (global -> registers now, registers -> LDS at the end of the step) ----------
    f32x4 we_st[NWE], wp_st[NWP];
    auto we_fetch = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWE; ++i) {
            const int idx = stid + 256 * i;
            if (WE4 % 256 == 0 || idx < WE4) {
                const int k = idx / (CE / 4), q = idx - k * (CE / 4);
                we_st[i] = *reinterpret_cast<const f32x4 *>(a.we + (size_t)k * CEXP + c * CE + 4 * q);
            }
        }
    };
    auto we_store = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWE; ++i) {
            const int idx = stid + 256 * i;
            if (WE4 % 256 == 0 || idx < WE4) {
                const int k = idx / (CE / 4), q = idx - k * (CE / 4);
                *reinterpret_cast<f32x4 *>(&WeS[slot][k * SWE + 4 * q]) = we_st[i];
            }
        }
    };
    auto wp_fetch = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWP; ++i) {
            const int idx = stid + 256 * i;
            if (WP4 % 256 == 0 || idx < WP4) {
                const int k = idx / (COUTP / 4), q = idx - k * (COUTP / 4);
                wp_st[i] = *reinterpret_cast<const f32x4 *>(a.wp + (size_t)(c * CE + k) * COUTP + 4 * q);
            }
        }
    };
    auto wp_store = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWP; ++i) {
            const int idx = stid + 256 * i;
            if (WP4 % 256 == 0 || idx < WP4) {
                const int k = idx / (COUTP / 4), q = idx - k * (COUTP / 4);
                *reinterpret_cast<f32x4 *>(&WpS[slot][k * SWP + 4 * q]) = wp_st[i];
            }
        }
    };
    // depthwise: lane = (channel dn, pixel group dg); weights + bias of one chunk per lane
    constexpr int PG = CE <= 16 ? 4 : (CE <= 32 ? 2 : 1);   // pixel groups per output row
    constexpr int PXG = IR_TW / PG;                          // output pixels per lane
    const int dn = lane % (64 / PG), dg = lane / (64 / PG);
    const int dnc = dn < CE ? dn : CE - 1;                  // lanes beyond the chunk repeat its last channel (no predicates)
    float wd_cur[10], wd_nxt[10];
    auto wd_fetch = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 9; ++k) wd_nxt[k] = a.wd[(size_t)k * CEXP + c * CE + dnc];
        wd_nxt[9] = a.bd[c * CE + dnc];
    };
    if (wave >= 4) we_fetch(0);                              // in flight while the input tile is loaded

    // ---- 1. input tile -> LDS ------------------------------------------------------------------------------------
    if constexpr (!STEM) {
        constexpr int CQ = KP / 4;
        const float *__restrict__ xin = a.x + (size_t)img * a.H * a.W * KP;
        for (int idx = tid; idx < MH * CQ; idx += IR_THREADS) {
            const int m = idx / CQ, cq = idx - m * CQ;
            const int hy = m / IW, hx = m - hy * IW;
            const int gy = gy0 + hy, gx = gx0 + hx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < NH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                v = *reinterpret_cast<const float4 *>(xin + ((size_t)gy * a.W + gx) * KP + 4 * cq);
            float2 *dst = reinterpret_cast<float2 *>(&Xs[m * SX + 4 * cq]);
            dst[0] = make_float2(v.x, v.y);
            dst[1] = make_float2(v.z, v.w);
        }
    } else {
        // raw patch: image rows 2*gy0 - pad .. + PR, columns (2*gx0 - pad) .. + PC/3, zero outside the image (ZeroPadding2D)
        const float *__restrict__ xin = a.x + (size_t)img * a.H * a.W * 3;
        const int iy0 = 2 * gy0 - a.pad, ix0 = 2 * gx0 - a.pad;
        for (int idx = tid; idx < PR * PC; idx += IR_THREADS) {
            const int pr = idx / PC, pc = idx - pr * PC;
            const int iy = iy0 + pr, ixc = ix0 * 3 + pc;                    // ixc = column * 3 + channel
            float v = 0.0f;
            if (iy >= 0 && iy < a.H && ixc >= 0 && ixc < a.W * 3) v = xin[(size_t)iy * a.W * 3 + ixc];
            Xs[pr * PSTR + pc] = v;
        }
    }
    // which of this lane's expand outputs (M-block mbi of the wave, row 4*lk + i) are pixels inside the image
    unsigned vmask = 0;
    if (wave < 4) {
#pragma unroll
        for (int mbi = 0; mbi < MBW; ++mbi)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = (wave + 4 * mbi) * 16 + 4 * lk + i;
                const int hy = m / IW, hx = m - hy * IW;
                const int gy = gy0 + hy, gx = gx0 + hx;
                if (m < NH && gy >= 0 && gy < a.DH && gx >= 0 && gx < a.DW) vmask |= 1u << (mbi * 4 + i);
            }
    } else {
        we_store(0);
        if (NCHUNK > 1) we_fetch(1);                         // put in place at the start of step 0
    }
    __syncthreads();
    IR_STAMP(0, 1);

    // STEM: per-lane patch offsets of the A fragments: base of the lane's pixel (per M-block), offset of its k per k-step
    int xbase[MBW], xoff[KS];
    if constexpr (STEM) {
#pragma unroll
        for (int mbi = 0; mbi < MBW; ++mbi) {
            const int m = min((wave & 3) * 16 + 64 * mbi + lr, NH - 1);     // rows >= NH are masked out: any valid address
            const int hy = m / IW, hx = m - hy * IW;
            xbase[mbi] = 2 * hy * PSTR + 6 * hx;
        }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int k = 4 * kk + lk;                                     // k = 27: zero weight row, any valid address
            const int r = k / 9, q = (k / 3) % 3, c = k % 3;
            xoff[kk] = k < 27 ? r * PSTR + q * 3 + c : 0;
        }
    }
    f32x4 pacc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) pacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bias_nxt[NBE];                                     // expand bias of the next chunk (matrix waves)
#pragma unroll
    for (int nb = 0; nb < NBE; ++nb) bias_nxt[nb] = a.be[nb * 16 + lr];
    const int mbp = wave & 1, nbp = (wave >> 1) & 1;       // projection: this wave's M-block and N-block parity
    const bool proj_wave = NBO >= 2 || nbp == 0;

    for (int it = 0; it < NCHUNK + 2; ++it) {
        if (wave < 4) {
            const bool do_e = it < NCHUNK, do_p = it >= 2;
            if (do_e) {
                // ---- E(it): (halo pixels x KP) * (KP x CE) -> + bias, ReLU6, zero outside the image -> Es[it & 1] -------
                float biasv[NBE];                        // loaded one step ahead (bias_nxt): no L2 round trip inside E
#pragma unroll
                for (int nb = 0; nb < NBE; ++nb) biasv[nb] = bias_nxt[nb];
                if (it + 1 < NCHUNK) {
#pragma unroll
                    for (int nb = 0; nb < NBE; ++nb) bias_nxt[nb] = a.be[(it + 1) * CE + nb * 16 + lr];
                }
                const float *wes = WeS[it & 1];
                float aq[2][EB][MBW], bq[2][EB][NBE];
                auto ld_e = [&](int b, float (&af)[EB][MBW], float (&bf)[EB][NBE]) {
#pragma unroll
                    for (int k4 = 0; k4 < EB; ++k4) {
                        const int kk = b * EB + k4;
                        if (kk < KS) {
#pragma unroll
                            for (int mbi = 0; mbi < MBW; ++mbi) {
                                const int mb = wave + 4 * mbi;
                                if (MB % 4 == 0 || mb < MB)
                                    af[k4][mbi] = STEM ? Xs[xbase[mbi] + xoff[kk]] : Xs[(mb * 16 + lr) * SX + 4 * kk + lk];
                            }
#pragma unroll
                            for (int nb = 0; nb < NBE; ++nb) bf[k4][nb] = wes[(4 * kk + lk) * SWE + nb * 16 + lr];
                        }
                    }
                };
                f32x4 eacc[MBW][NBE];
#pragma unroll
                for (int mbi = 0; mbi < MBW; ++mbi)
#pragma unroll
                    for (int nb = 0; nb < NBE; ++nb) eacc[mbi][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
                ld_e(0, aq[0], bq[0]);
#pragma unroll
                for (int b = 0; b < NEB; ++b) {
                    // the operands of batch b+1 are requested BEFORE the MFMAs of batch b; without the scheduling
                    // barriers the compiler sinks every LDS read to its use (shortest live range) and each pair of
                    // MFMAs waits for a fresh LDS round trip (measured: 71 instead of 32 cycles per MFMA)
                    if (b + 1 < NEB) ld_e(b + 1, aq[(b + 1) & 1], bq[(b + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k4 = 0; k4 < EB; ++k4) {
                        if (b * EB + k4 < KS) {
#pragma unroll
                            for (int mbi = 0; mbi < MBW; ++mbi) {
                                const int mb = wave + 4 * mbi;
                                if (MB % 4 == 0 || mb < MB) {
#pragma unroll
                                    for (int nb = 0; nb < NBE; ++nb)
                                        eacc[mbi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                            aq[b & 1][k4][mbi], bq[b & 1][k4][nb], eacc[mbi][nb], 0, 0, 0);
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                float *es = Es[it & 1];
#pragma unroll
                for (int mbi = 0; mbi < MBW; ++mbi) {
                    const int mb = wave + 4 * mbi;
                    if (MB % 4 == 0 || mb < MB) {
#pragma unroll
                        for (int nb = 0; nb < NBE; ++nb)
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                es[(mb * 16 + 4 * lk + i) * SEP + nb * 16 + lr] =
                                    ((vmask >> (mbi * 4 + i)) & 1u) ? relu6f(eacc[mbi][nb][i] + biasv[nb]) : 0.0f;
                    }
                }
            }
            IR_STAMP(0, 2 + 4 * it);
            if (do_p && proj_wave) {
                // ---- P(it-2): acc (32 px x COUT) += D (32 x CE) * Wp[chunk it-2]; D = Ds[it & 1], Wp = WpS[it & 1] ------
                const float *ds = Ds[it & 1];
                const float *wps = WpS[it & 1];
                float pa[2][PB], pb[2][PB][NJ];
                auto ld_p = [&](int b, float (&af)[PB], float (&bf)[PB][NJ]) {
#pragma unroll
                    for (int k4 = 0; k4 < PB; ++k4) {
                        const int kk = b * PB + k4;
                        af[k4] = ds[(mbp * 16 + lr) * SD + 4 * kk + lk];
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const int nb = nbp + 2 * j;
                            if (NBO % 2 == 0 || nb < NBO) bf[k4][j] = wps[(4 * kk + lk) * SWP + nb * 16 + lr];
                        }
                    }
                };
                ld_p(0, pa[0], pb[0]);
#pragma unroll
                for (int b = 0; b < NPB; ++b) {
                    if (b + 1 < NPB) ld_p(b + 1, pa[(b + 1) & 1], pb[(b + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k4 = 0; k4 < PB; ++k4)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const int nb = nbp + 2 * j;
                            if (NBO % 2 == 0 || nb < NBO)
                                pacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[b & 1][k4], pb[b & 1][k4][j], pacc[j], 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            IR_STAMP(0, 3 + 4 * it);
        } else {
            // ---- service waves --------------------------------------------------------------------------------------------
            // Staging runs one step ahead of its use AND one step behind its loads: what was requested at the start of the
            // previous step (We(it+1), Wp(it-1), the depthwise weights of chunk it-1) has had a whole step to arrive and
            // is put in place now; then the requests for the next step go out; then the depthwise.  (No lane predicates
            // on these loads: with exec-masked loads in flight the compiler parks a vmcnt(0) in front of the depthwise.)
#pragma unroll
            for (int k = 0; k < 10; ++k) wd_cur[k] = wd_nxt[k];
            if (it + 1 < NCHUNK) we_store((it + 1) & 1);
            if (it >= 1 && it <= NCHUNK) wp_store((it - 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (it + 2 < NCHUNK) we_fetch(it + 2);
            if (it < NCHUNK) wp_fetch(it);
            if (it < NCHUNK) wd_fetch(it);
            IR_STAMP(4, 65 + 4 * it);
            if (it >= 1 && it <= NCHUNK) {
                // D(it-1): depthwise 3x3 + bias + ReLU6 on chunk it-1.  wave - 4 = output row; lane = (channel dn, pixel
                // group dg): PXG consecutive output pixels of the row per lane (8 for CE > 32, 4 for CE = 32, 2 for CE = 16,
                // so that small chunks still use all 64 lanes).  The lane's 3 x ((PXG-1)*S + 3) input window is read
                // once, then 9 FMAs per output.  Lanes beyond CE (CE = 48) compute channel CE-1 again and do not store.
                const int c = it - 1;
                const float *es = Es[c & 1] + dnc;
                float *ds = Ds[c & 1] + dnc;
                const int py = wave - 4;
                constexpr int WW = (PXG - 1) * S + 3;
                float win[3][WW];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int x = 0; x < WW; ++x) win[r][x] = es[((py * S + r) * IW + dg * PXG * S + x) * SEP];
                __builtin_amdgcn_sched_barrier(0);
                IR_STAMP(4, 64 + 4 * it);
#pragma unroll
                for (int px = 0; px < PXG; ++px) {
                    float acc = wd_cur[9];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int q = 0; q < 3; ++q) acc = fmaf(win[r][px * S + q], wd_cur[r * 3 + q], acc);
                    if (dn < CE) ds[(py * IR_TW + dg * PXG + px) * SD] = relu6f(acc);
                }
            }
            IR_STAMP(4, 4 + 4 * it);
        }
        __syncthreads();
        IR_STAMP(0, 5 + 4 * it);
    }

    // ---- 3. + bias (+ residual from the input tile) -> NHWC ----------------------------------------------------------
    if (wave < 4 && proj_wave) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int nb = nbp + 2 * j;
            const int co = nb * 16 + lr;
            if ((NBO % 2 == 0 || nb < NBO) && co < COUT) {
                const float bias = a.bp[co];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int p = mbp * 16 + 4 * lk + i;
                    const int py = p >> 3, px = p & 7;
                    const int oy = oy0 + py, ox = ox0 + px;
                    if (oy < a.OH && ox < a.OW) {
                        float v = pacc[j][i] + bias;
                        if constexpr (RES) v += Xs[((py + 1) * IW + px + 1) * SX + co];
                        a.out[(((size_t)img * a.OH + oy) * a.OW + ox) * COUT + co] = v;
                    }
                }
            }
        }
    }
}

// ---- stem + expanded_conv block, 8 x 16 output tiles ---------------------------------------------------------------
// Conv1 (3x3 stride 2, 3 -> 32, BN, ReLU6) -> expanded_conv depthwise 3x3 (BN, ReLU6) -> project 1x1 32 -> 16 (BN), the
// first launch of the MobileNetV2 graph.  In ir_block_kernel<STEM> this is a single chunk: nothing for the E / D / P
// pipeline to overlap, three barrier-separated phases on a 32-pixel tile with half the workgroup idle in each, 45 KB of
// LDS -- 3 tiles of 32 pixels in flight per CU, every one a chain of L2 / LDS round trips: 100 us at batch 8 / 500 x 500
// against ~10 us for either roof (1.66 GFLOP on the f32 MFMA; 56 MB of input + output).  Here a 256-thread workgroup
// owns 8 x 16 output pixels (halo 10 x 18: 1.4x instead of 1.9x redundant Conv1 work), all four waves work in every
// phase, and the weights live in registers (loaded once, before the image patch arrives):
//   E  Conv1 on the 180 halo pixels: A gathered from the raw image patch in LDS (im2col entry = base(pixel) + off(k)),
//      12 M-blocks x 2 N-blocks x 7 k-steps of v_mfma_f32_16x16x4_f32 (exact f32), 3 M-blocks per wave;
//      + bias, ReLU6, zero outside Conv1's grid (the depthwise pads the EXPANDED tensor) -> Es
//   D  depthwise: thread = (channel, output row half): 16 outputs from a 3 x 18 window read once -> Ds (overlays the patch)
//   P  (128 px x 32) * (32 x 16) on the MFMA, 2 M-blocks per wave, + bias -> NHWC
constexpr int ST_TW = 16, ST_THREADS = 256;

// ST_TH x 16 output pixels per workgroup: 8 (45 KB of LDS, three workgroups per CU) or 4 (25 KB, six; 1.7x instead of 1.4x
// redundant Conv1 work on the halo, but twice the resident workgroups to cover the phases' round trips).
template <int ST_TH>
__global__ void __launch_bounds__(ST_THREADS, ST_TH == 8 ? 5 : 1)
stem_block_kernel(IrArgs a)
{
    constexpr int IH = ST_TH + 2, IW = ST_TW + 2, NH = IH * IW;          // halo on Conv1's grid: 10 x 18 = 180 | 6 x 18 = 108 pixels
    constexpr int MB = (NH + 15) / 16;                                    // 12 | 7 M-blocks
    constexpr int MBW = (MB + 3) / 4;                                     // 3 | 2 per wave (round robin)
    constexpr int PMW = ST_TH * ST_TW / 64;                               // projection M-blocks per wave: 2 | 1
    constexpr int PR = 2 * (IH - 1) + 3, PC = (2 * (IW - 1) + 3) * 3, PSTR = PC + 2;   // image patch: 21 rows x 111 floats
    constexpr int SEP = 36, SD = 34;                                      // row strides of Es / Ds (see ir_block_kernel)
    constexpr int KS = 7;                                                 // k-steps of Conv1 (K = 27 -> 28)
    static_assert(ST_TH == 8 || ST_TH == 4, "tile");
    // Ds overlays Es (every depthwise thread holds its 3 x 18 window in registers before the first Ds element is written: one more
    // barrier), so that the workgroup needs 37 KB instead of 45: FOUR workgroups per CU instead of three (registers allow four)
    constexpr int ES_FLOATS = MB * 16 * SEP > ST_TH * ST_TW * SD ? MB * 16 * SEP : ST_TH * ST_TW * SD;
    static_assert(PR * PSTR <= ES_FLOATS, "the patch fits under Es");
    __shared__ __attribute__((aligned(16))) float Es[ES_FLOATS];          // the image patch, then Conv1's outputs on the halo, then Ds
    float *const XD = Es;                                                  // (the patch is dead once E's gathers are in registers)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    int t = blockIdx.x;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int img = t / a.tiles_y;
    const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
    const int gy0 = oy0 - 1, gx0 = ox0 - 1;                               // halo origin on Conv1's grid

    // weights -> registers (in flight while the patch is loaded)
    float we_r[KS][2], wp_r[8], wd_r[10];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) we_r[kk][nb] = a.we[(4 * kk + lk) * 32 + nb * 16 + lr];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) wp_r[kk] = a.wp[(4 * kk + lk) * 16 + lr];
    const int dc = tid & 31, dg = tid >> 5;                               // depthwise: channel, pixel group (8: a row | half a row)
    constexpr int DPX = ST_TH * ST_TW / 8;                                // outputs per depthwise thread: 16 | 8
    const int drow = ST_TH == 8 ? dg : (dg >> 1), dcol0 = ST_TH == 8 ? 0 : (dg & 1) * 8;
#pragma unroll
    for (int k = 0; k < 9; ++k) wd_r[k] = a.wd[k * 32 + dc];
    wd_r[9] = a.bd[dc];
    const float be0 = a.be[lr], be1 = a.be[16 + lr], bpv = a.bp[lr];

    // ---- image patch: rows 2 gy0 - pad .. + 21, columns (2 gx0 - pad) .. + 37, zero outside the image (ZeroPadding2D)
    {
        const float *__restrict__ xin = a.x + (size_t)img * a.H * a.W * 3;
        const int iy0 = 2 * gy0 - a.pad, ixc0 = (2 * gx0 - a.pad) * 3;
        for (int idx = tid; idx < PR * PC; idx += ST_THREADS) {
            const int pr = idx / PC, pc = idx - pr * PC;
            const int iy = iy0 + pr, ixc = ixc0 + pc;                      // ixc = column * 3 + channel
            float v = 0.0f;
#ifndef RPN_EXP_STEM_NOLOAD      /* timing experiments only (results are wrong) */
            if (iy >= 0 && iy < a.H && ixc >= 0 && ixc < a.W * 3) v = xin[(size_t)iy * a.W * 3 + ixc];
#endif
            XD[pr * PSTR + pc] = v;
        }
    }
    // which of this lane's Conv1 outputs (M-block wave + 4 mbi, row 4 lk + i) lie inside Conv1's grid
    unsigned vmask = 0;
    int xbase[MBW], xoff[KS];
#pragma unroll
    for (int mbi = 0; mbi < MBW; ++mbi) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = (wave + 4 * mbi) * 16 + 4 * lk + i;
            const int hy = m / IW, hx = m - hy * IW;
            const int gy = gy0 + hy, gx = gx0 + hx;
            if (m < NH && gy >= 0 && gy < a.DH && gx >= 0 && gx < a.DW) vmask |= 1u << (mbi * 4 + i);
        }
        const int m = min((wave + 4 * mbi) * 16 + lr, NH - 1);            // rows >= NH are masked out: any valid address
        const int hy = m / IW, hx = m - hy * IW;
        xbase[mbi] = 2 * hy * PSTR + 6 * hx;
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const int k = 4 * kk + lk;                                       // k = 27: zero weight row, any valid address
        const int r = k / 9, q = (k / 3) % 3, c = k % 3;
        xoff[kk] = k < 27 ? r * PSTR + q * 3 + c : 0;
    }
    __syncthreads();

    // ---- E: Conv1 ----------------------------------------------------------------------------------------------------
    {
        f32x4 eacc[MBW][2];
#pragma unroll
        for (int mbi = 0; mbi < MBW; ++mbi)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) eacc[mbi][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        float av[KS][MBW];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int mbi = 0; mbi < MBW; ++mbi) av[kk][mbi] = XD[xbase[mbi] + xoff[kk]];     // all 21 gathers in flight
        __syncthreads();                                                  // every gather has landed: Es may overwrite the patch
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int mbi = 0; mbi < MBW; ++mbi)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#ifndef RPN_EXP_STEM_NOE
                    if (MB % 4 == 0 || wave + 4 * mbi < MB)
                        eacc[mbi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk][mbi], we_r[kk][nb], eacc[mbi][nb], 0, 0, 0);
#else
                    eacc[mbi][nb][0] += av[kk][mbi] * we_r[kk][nb];
#endif
#pragma unroll
        for (int mbi = 0; mbi < MBW; ++mbi)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!(MB % 4 == 0 || wave + 4 * mbi < MB)) continue;
                const bool in = (vmask >> (mbi * 4 + i)) & 1u;
                float *e = &Es[((wave + 4 * mbi) * 16 + 4 * lk + i) * SEP + lr];
                e[0] = in ? relu6f(eacc[mbi][0][i] + be0) : 0.0f;
                e[16] = in ? relu6f(eacc[mbi][1][i] + be1) : 0.0f;
            }
    }
    __syncthreads();                                                      // Es complete

    // ---- D: depthwise 3x3 + bias + ReLU6: thread = (channel dc, output row dg), 16 outputs ----------------------------
#ifndef RPN_EXP_STEM_NOD
    {
        const float *es = Es + (drow * IW + dcol0) * SEP + dc;
        float *ds = Es + (drow * ST_TW + dcol0) * SD + dc;
        float win[3][DPX + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int x = 0; x < DPX + 2; ++x) win[r][x] = es[(r * IW + x) * SEP];
        __syncthreads();                                                  // all windows are in registers: Ds may overwrite Es
#pragma unroll
        for (int px = 0; px < DPX; ++px) {
            float acc = wd_r[9];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q) acc = fmaf(win[r][px + q], wd_r[r * 3 + q], acc);
            ds[px * SD] = relu6f(acc);
        }
    }
#endif
    __syncthreads();

    // ---- P: projection 32 -> 16 + bias -> NHWC -----------------------------------------------------------------------
    {
        f32x4 pacc[PMW];
        float pa[8][PMW];
#pragma unroll
        for (int mbi = 0; mbi < PMW; ++mbi) pacc[mbi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int mbi = 0; mbi < PMW; ++mbi) pa[kk][mbi] = Es[((PMW * wave + mbi) * 16 + lr) * SD + 4 * kk + lk];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int mbi = 0; mbi < PMW; ++mbi)
#ifndef RPN_EXP_STEM_NOP
                pacc[mbi] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[kk][mbi], wp_r[kk], pacc[mbi], 0, 0, 0);
#else
                pacc[mbi][0] += pa[kk][mbi] * wp_r[kk];
#endif
#pragma unroll
        for (int mbi = 0; mbi < PMW; ++mbi)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = (PMW * wave + mbi) * 16 + 4 * lk + i;       // = output row p / 16, column p % 16
                const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
                if (oy < a.OH && ox < a.OW) a.out[(((size_t)img * a.OH + oy) * a.OW + ox) * 16 + lr] = pacc[mbi][i] + bpv;
            }
    }
}

// ---- high-resolution blocks (blocks 1-3: 16 / 24 input channels), 4 x 8 (or 8 x 8) output tiles ---------------------------
// ir_block_kernel's two-group pipeline pays one barrier-separated step of a few microseconds per chunk of expanded
// channels whatever the work in it; with 16-32 input channels a step holds a few hundred cycles of MFMA, the 512-thread
// workgroup is mostly latency, and the f32 MFMA shares its issue port with the vector work of the service waves.  Here
// (the stem kernel's recipe) a 256-thread workgroup owns TH x 8 output pixels and ALL four waves work in every phase:
//   per chunk of CE expanded channels:  E  expand on the halo pixels (M-blocks dealt to the waves round robin; B operand =
//                                          this chunk's We columns, straight from global memory into registers)
//                                          + bias, ReLU6, zero outside the image -> Es          | barrier
//                                       D  depthwise 3x3 + bias + ReLU6, thread = (channel, pixel group) -> Ds | barrier
//                                       P  projection accumulate: wave w owns output M-block w (16 pixels)
//   + bias (+ residual from the input tile in LDS) -> NHWC.
// Two barriers per chunk (P(c) and E(c+1) touch disjoint buffers).  Overlap comes from the other workgroups of the CU: the
// 4-row tile (25-32 KB of LDS, six workgroups per CU) beats the 8-row one (three) on every block and batch size measured.
constexpr int HR_T = 8, HR_THREADS = 256;

// TH x 8 output pixels per workgroup (TH = 8 | 4: with 4 rows the projection's two M-blocks are shared by wave pairs
// that split the output-channel blocks, as in ir_block_kernel).
template <int CIN, int CEXP, int CE, int COUT, int S, bool RES, int TH = 8>
__global__ void __launch_bounds__(HR_THREADS)
ir_block_hr_kernel(IrArgs a)
{
    constexpr int SX = CIN + 2;
    constexpr int NPX = TH * HR_T;                                       // output pixels: 64 | 32
    constexpr int IH = (TH - 1) * S + 3, IW = (HR_T - 1) * S + 3, NH = IH * IW;   // halo on the input grid
    constexpr int MB = (NH + 15) / 16, MH = MB * 16, MBW = (MB + 3) / 4;
    constexpr int SEP = CE + 4, SD = CE + 2;
    constexpr int NBE = CE / 16, NCHUNK = CEXP / CE;
    constexpr int COUTP = (COUT + 15) / 16 * 16, NBO = COUTP / 16;
    constexpr int KS = CIN / 4, PS = CE / 4;
    constexpr int DG = CE == 16 ? 16 : (CE == 32 ? 8 : 4);               // depthwise pixel groups
    constexpr int PXG = NPX / DG;                                         // output pixels per depthwise thread (2 .. 16)
    constexpr int PMB = NPX / 16;                                         // M-blocks of the projection: 4 | 2
    constexpr int NJ = PMB == 4 ? NBO : (NBO + 1) / 2;                    // ... N-blocks per wave
    static_assert(TH == 8 || TH == 4, "tile");
    constexpr int DROWS = PXG <= HR_T ? 1 : PXG / HR_T, DCOLS = PXG <= HR_T ? PXG : HR_T;   // ... as rows x columns
    static_assert(CIN % 4 == 0 && (SX / 2) % 2 == 1 && (SD / 2) % 2 == 1 && SEP % 8 == 4 && CEXP % CE == 0 && CE % 16 == 0 &&
                      CE <= 48 && DG * CE <= HR_THREADS, "layout");
    static_assert(!RES || (S == 1 && CIN == COUT), "residual needs stride 1 and Cin == Cout");
    __shared__ __attribute__((aligned(16))) float Xs[MH * SX];
    __shared__ __attribute__((aligned(16))) float Es[MH * SEP];
    __shared__ __attribute__((aligned(16))) float Ds[NPX * SD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    int t = blockIdx.x;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int img = t / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * HR_T;
    const int gy0 = oy0 * S - a.pad, gx0 = ox0 * S - a.pad;              // halo origin on the input grid
    const int pmb = PMB == 4 ? wave : (wave & 1), pnb0 = PMB == 4 ? 0 : (wave >> 1);   // projection: M-block, first N-block
    constexpr int PNS = PMB == 4 ? 1 : 2;                                               // ... N-block stride

    // ---- input halo tile -> LDS (zero outside the image: those rows only feed masked-out expand outputs) ---------------
    {
        constexpr int CQ = CIN / 4;
        const float *__restrict__ xin = a.x + (size_t)img * a.H * a.W * CIN;
        for (int idx = tid; idx < MH * CQ; idx += HR_THREADS) {
            const int m = idx / CQ, cq = idx - m * CQ;
            const int hy = m / IW, hx = m - hy * IW;
            const int gy = gy0 + hy, gx = gx0 + hx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < NH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                v = *reinterpret_cast<const float4 *>(xin + ((size_t)gy * a.W + gx) * CIN + 4 * cq);
            float2 *dst = reinterpret_cast<float2 *>(&Xs[m * SX + 4 * cq]);
            dst[0] = make_float2(v.x, v.y);
            dst[1] = make_float2(v.z, v.w);
        }
    }
    // which of this lane's expand outputs (M-block wave + 4 mbi, row 4 lk + i) are pixels inside the image
    unsigned vmask = 0;
#pragma unroll
    for (int mbi = 0; mbi < MBW; ++mbi)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = (wave + 4 * mbi) * 16 + 4 * lk + i;
            const int hy = m / IW, hx = m - hy * IW;
            const int gy = gy0 + hy, gx = gx0 + hx;
            if (m < NH && gy >= 0 && gy < a.DH && gx >= 0 && gx < a.DW) vmask |= 1u << (mbi * 4 + i);
        }
    const int dc = tid % CE, dg = tid / CE;                              // depthwise: channel, pixel group
    const int dy0 = (dg * PXG) / HR_T, dx0 = (dg * PXG) % HR_T;          // first output pixel of the group
    f32x4 pacc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) pacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    for (int c = 0; c < NCHUNK; ++c) {
        // ---- E(c): (halo pixels x CIN) * We[:, chunk] -> + bias, ReLU6, zero outside the image -> Es ------------------
        {
            float we_r[KS][NBE], be_r[NBE];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int nb = 0; nb < NBE; ++nb) we_r[kk][nb] = a.we[(size_t)(4 * kk + lk) * CEXP + c * CE + nb * 16 + lr];
#pragma unroll
            for (int nb = 0; nb < NBE; ++nb) be_r[nb] = a.be[c * CE + nb * 16 + lr];
#pragma unroll
            for (int mbi = 0; mbi < MBW; ++mbi) {
                const int mb = wave + 4 * mbi;
                if (MB % 4 == 0 || mb < MB) {
                    float av[KS];
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) av[kk] = Xs[(mb * 16 + lr) * SX + 4 * kk + lk];
                    f32x4 eacc[NBE];
#pragma unroll
                    for (int nb = 0; nb < NBE; ++nb) eacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                        for (int nb = 0; nb < NBE; ++nb)
                            eacc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], we_r[kk][nb], eacc[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < NBE; ++nb)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            Es[(mb * 16 + 4 * lk + i) * SEP + nb * 16 + lr] =
                                ((vmask >> (mbi * 4 + i)) & 1u) ? relu6f(eacc[nb][i] + be_r[nb]) : 0.0f;
                }
            }
        }
        __syncthreads();
        // ---- D(c): depthwise 3x3 + bias + ReLU6; thread = (channel dc, pixel group dg) --------------------------------
        if (dg < DG) {
            float wd_r[10];
#pragma unroll
            for (int k = 0; k < 9; ++k) wd_r[k] = a.wd[(size_t)k * CEXP + c * CE + dc];
            wd_r[9] = a.bd[c * CE + dc];
#pragma unroll
            for (int ry = 0; ry < DROWS; ++ry) {
                const int py = dy0 + ry;
                constexpr int WW = (DCOLS - 1) * S + 3;
                float win[3][WW];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int x = 0; x < WW; ++x) win[r][x] = Es[((py * S + r) * IW + dx0 * S + x) * SEP + dc];
#pragma unroll
                for (int px = 0; px < DCOLS; ++px) {
                    float acc = wd_r[9];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int q = 0; q < 3; ++q) acc = fmaf(win[r][px * S + q], wd_r[r * 3 + q], acc);
                    Ds[(py * HR_T + dx0 + px) * SD + dc] = relu6f(acc);
                }
            }
        }
        __syncthreads();
        // ---- P(c): acc (16 px of M-block `wave` x COUT) += D (16 x CE) * Wp[chunk] -------------------------------------
        {
            float pa[PS];
#pragma unroll
            for (int kk = 0; kk < PS; ++kk) pa[kk] = Ds[(pmb * 16 + lr) * SD + 4 * kk + lk];
#pragma unroll
            for (int kk = 0; kk < PS; ++kk)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int nb = pnb0 + PNS * j;
                    if (PMB == 4 || NBO % 2 == 0 || nb < NBO) {
                        const float wv = a.wp[(size_t)(c * CE + 4 * kk + lk) * COUTP + nb * 16 + lr];
                        pacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[kk], wv, pacc[j], 0, 0, 0);
                    }
                }
        }
    }
    // ---- + bias (+ residual from the input tile) -> NHWC -------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int nb = pnb0 + PNS * j;
        const int co = nb * 16 + lr;
        if ((PMB == 4 || NBO % 2 == 0 || nb < NBO) && co < COUT) {
            const float bias = a.bp[co];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = pmb * 16 + 4 * lk + i;
                const int py = p >> 3, px = p & 7;
                const int oy = oy0 + py, ox = ox0 + px;
                if (oy < a.OH && ox < a.OW) {
                    float v = pacc[j][i] + bias;
                    if constexpr (RES) v += Xs[((py + 1) * IW + px + 1) * SX + co];
                    a.out[(((size_t)img * a.OH + oy) * a.OW + ox) * COUT + co] = v;
                }
            }
        }
    }
}

// ---- f16x3 variant for the low-resolution blocks (Cin = 64 / 96, stride 1) -----------------------------------------
// Same pipeline (E / D / P over two wave groups, one barrier per step), but the two GEMMs run on
// v_mfma_f32_16x16x32_f16 with every float32 operand carried as hi + lo float16 halves and each product formed as
// lo*hi + hi*lo + hi*hi with float32 accumulation (product error ~2^-21, the arithmetic the f16x3 mode
// already uses in the 3x3 layers).  Why only here: the float32 MFMA runs at 1/16 of the 16-bit rate AND shares the
// SIMD's issue port with the vector ALU, so on these blocks (one workgroup per CU, 72 + 36 f32 MFMAs per wave and
// step) the matrix work was the whole step; on the 16-bit MFMA it is 27 + 18 instructions of 16 cycles on the separate
// matrix pipe and the step becomes the service waves' vector work.  The high-resolution blocks (Cin <= 32) are bound by
// that vector work already and keep the exact-f32 kernel.
// Operand images in LDS are FRAGMENT-MAJOR: one 1 KB block per (16-row block, 32-deep k-step, hi | lo) holding the 64
// lanes' 16-byte fragments in lane order (lane = 16 * (k / 8 % 4) + row), so every MFMA operand read is one linear,
// conflict-free ds_read_b128.  Weights are packed that way on the host (per chunk of 32 expanded channels), so staging
// them is a linear copy.  CE = 32.
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

struct IrX3Args {
    const float *x;
    float *out;
    const u32x4 *we;      // [chunk][nb 2][ks K/32][hi|lo][64 lanes] fragments of the expand weights * 2^shift_e
    const u32x4 *wp;      // [chunk][nb COUT/16][hi|lo][64 lanes] fragments of the projection weights * 2^shift_p
    const float *be, *wd, *bd, *bp;
    float scale_e, scale_p;    // 2^-shift
    int B, H, W, tiles_x, tiles_y;
    unsigned *status;     // float16 range flag (block input beyond 65504), or null
    int stamp;
    // K-split for small grids (one image: 32 tiles on 256 CUs): `ksplit` workgroups per tile (1 | 2 | 3 | 6) each take a run of
    // the expanded-channel chunks; their partial projections go to `part` ([tile][ksplit][32 px][COUT] floats) and the
    // workgroup that arrives last (tickets[tile], left at zero) adds them in the FIXED order every launch uses -- see the kernel
    int ksplit;
    float *part;
    unsigned *tickets;
};

__device__ __forceinline__ f32x4 mfma_x3(u32x4 ahi, u32x4 alo, u32x4 bhi, u32x4 blo, f32x4 c)
{
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, alo), __builtin_bit_cast(f16x8, bhi), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ahi), __builtin_bit_cast(f16x8, blo), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ahi), __builtin_bit_cast(f16x8, bhi), c, 0, 0, 0);
}

// Slot swizzle of the fragment-major A images the kernels BUILD in LDS (input tile, depthwise output).  A 1 KB block holds
// 64 slots of 16 bytes, slot = 16 * kq + row for the lane (row, kq) that reads it.  The builders scatter: for one pixel (row)
// the 32 channels of a chunk land in the four kq slots of the hi block and of the lo block -- slots 16 apart = the same
// four LDS banks: an 8-way conflict on every store (rocprofv3, round 3: 25-49 % of the LDS cycles of these kernels).  With
// slot = 16 * kq + (row ^ (kq | sel << 2)) -- sel = hi / lo for the depthwise image, the parity of the k-step for the input
// image -- the eight stores of a pixel fall into eight different bank groups, and the reading side stays conflict-free: a
// ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32), the XOR only permutes rows inside
// the aligned sets {0-3}, {4-7}, {8-11}, {12-15} (kq < 4) or swaps such sets pairwise (sel), so every group still covers
// 16 different slots mod 16.
// `rot` (0 | 2): a second k-step of the same image XORs kq with 2 first -- ir_block_hrx3_kernel at CE = 48 stores a pixel's
// channels 32 .. 47 (k-step 1) and, from the same 32-lane group, channels 0 .. 15 of the pixel 8 further on (k-step 0, same low
// row bits): without it those two fall into the same bank groups (2-way; measured 15 % of that kernel's LDS cycles).
__device__ __forceinline__ int frag_slot(int kq, int row, int sel, int rot = 0) { return kq * 16 + (row ^ ((kq ^ rot) | (sel << 2))); }
__device__ __forceinline__ int frag_lane(int lane, int sel, int rot = 0) { return lane ^ (((lane >> 4) ^ rot) | (sel << 2)); }

template <int KP, int CEXP, int COUT, bool RES>
__global__ void __launch_bounds__(IR_THREADS)
ir_block_x3_kernel(IrX3Args a)
{
    constexpr int S = 1, CE = 32;
    constexpr int IH = IR_TH + 2, IW = IR_TW + 2, NH = IH * IW;           // 6 x 10 halo pixels
    constexpr int MB = 4, MH = 64;                                        // expand M-blocks: one per matrix wave
    constexpr int KS = KP / 32;                                           // k-steps of the expand GEMM
    constexpr int NBE = CE / 16, NCHUNK = CEXP / CE;
    constexpr int SEP = CE + 4;
    constexpr int NBO = COUT / 16, NJ = NBO / 2;                          // projection N-blocks, per wave
    constexpr int WE_P = NBE * KS * 2 * 64, WP_P = NBO * 2 * 64;          // 16-byte pieces per chunk of We / Wp
    constexpr int NWE = (WE_P + 255) / 256, NWP = (WP_P + 255) / 256;
    static_assert(KP % 32 == 0 && CEXP % CE == 0 && COUT % 32 == 0 && (!RES || KP == COUT), "shape");

    __shared__ u32x4 XsF[MB * KS * 2 * 64];                               // [mb][ks][hi|lo][lane]
    __shared__ __attribute__((aligned(16))) float Es[2][MH * SEP];
    __shared__ u32x4 DsF[2][2 * 2 * 64];                                  // [slot][mb 2][hi|lo][lane]  (K = CE = one k-step)
    __shared__ u32x4 WeS[2][WE_P];
    __shared__ u32x4 WpS[2][WP_P];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int stid = tid;                                     // weight staging is done by the MATRIX waves here (they have the slack)
    // Projection sums are ALWAYS formed as a fixed tree over six equal runs ("leaves") of the chunks:
    //     total = ((leaf0 + leaf1) + (leaf2 + leaf3)) + (leaf4 + leaf5),   each leaf an MFMA chain from zero over its chunks.
    // One workgroup per tile (ksplit = 1) folds the leaves as it goes; with ksplit = 3 a workgroup computes one pair, with 6 one
    // leaf, with 2 the first two pairs' sum | the third pair, and the last arriver finishes the same tree from the stored partials.  Every split factor therefore gives the same
    // bits -- the grid-dependent choice (few tiles: one image) cannot make a batch-1 result differ from the same image inside
    // a larger batch (test_model_batch_invariance*).
    constexpr int LEAF = NCHUNK / 6;
    static_assert(NCHUNK % 6 == 0, "the projection tree has six leaves");
    const int ksplit = a.ksplit;
    const int tile = (int)blockIdx.x / ksplit, part_h = (int)blockIdx.x - tile * ksplit;
    // this workgroup's chunks [c0, c0 + nch): equal runs, except ksplit = 2 = the tree's (leaf0..3 | leaf4..5)
    const int nch = ksplit == 2 ? (part_h == 0 ? 4 * LEAF : 2 * LEAF) : NCHUNK / ksplit;
    const int c0 = ksplit == 2 ? part_h * 4 * LEAF : part_h * nch;
    int t = tile;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int img = t / a.tiles_y;
    const int oy0 = ty * IR_TH, ox0 = tx * IR_TW;
    const int gy0 = oy0 - 1, gx0 = ox0 - 1;
    IR_STAMP(0, 0);
    IR_STAMP_RT(0, 104);
    if (wave >= 4) __builtin_amdgcn_s_setprio(2);            // (see ir_block_kernel)

    u32x4 we_st[NWE], wp_st[NWP];
    auto we_fetch = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWE; ++i)
            if (WE_P % 256 == 0 || stid + 256 * i < WE_P) we_st[i] = a.we[(size_t)c * WE_P + stid + 256 * i];
    };
    auto we_store = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWE; ++i)
            if (WE_P % 256 == 0 || stid + 256 * i < WE_P) WeS[slot][stid + 256 * i] = we_st[i];
    };
    auto wp_fetch = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWP; ++i)
            if (WP_P % 256 == 0 || stid + 256 * i < WP_P) wp_st[i] = a.wp[(size_t)c * WP_P + stid + 256 * i];
    };
    auto wp_store = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWP; ++i)
            if (WP_P % 256 == 0 || stid + 256 * i < WP_P) WpS[slot][stid + 256 * i] = wp_st[i];
    };
    constexpr int PG = 2, PXG = IR_TW / PG;                              // depthwise: lane = (channel dn, pixel group dg)
    const int dn = lane & 31, dg = lane >> 5;
    float wd_cur[10], wd_nxt[10];
    // depthwise weights of a chunk: ONE 48-byte record per channel -- 9 taps, the bias, 2 floats of padding (pack_ir_x3_dw) --
    // = three 16-byte loads per lane and step (ten 4-byte loads from the [9][CEXP] matrix + bias cost 13 % of blocks 4, 5, 7-9:
    // their ISSUE, ~660 cycles at the head of every step beside the matrix waves' weight requests)
    auto wd_fetch = [&](int c) __attribute__((always_inline)) {
        const float4 *rec = reinterpret_cast<const float4 *>(a.wd) + ((size_t)c * CE + dn) * 3;
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        wd_nxt[0] = q0.x; wd_nxt[1] = q0.y; wd_nxt[2] = q0.z; wd_nxt[3] = q0.w;
        wd_nxt[4] = q1.x; wd_nxt[5] = q1.y; wd_nxt[6] = q1.z; wd_nxt[7] = q1.w;
        wd_nxt[8] = q2.x; wd_nxt[9] = q2.y;
    };
    if (wave < 4) we_fetch(c0);

    // ---- 1. input tile: float32 NHWC -> hi / lo float16, fragment-major ---------------------------------------------
    {
        constexpr int CQ = KP / 4;
        const float *__restrict__ xin = a.x + (size_t)img * a.H * a.W * KP;
        _Float16 *xh = reinterpret_cast<_Float16 *>(XsF);
        for (int idx = tid; idx < MH * CQ; idx += IR_THREADS) {
            const int m = idx / CQ, cq = idx - m * CQ;
            const int hy = m / IW, hx = m - hy * IW;
            const int gy = gy0 + hy, gx = gx0 + hx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < NH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                v = *reinterpret_cast<const float4 *>(xin + ((size_t)gy * a.W + gx) * KP + 4 * cq);
            if (a.status && !(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) <= 65504.0f))
                atomicOr(a.status, 1u);
            const f16x4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
            const f16x4 lo = {(_Float16)(v.x - (float)hi[0]), (_Float16)(v.y - (float)hi[1]), (_Float16)(v.z - (float)hi[2]),
                              (_Float16)(v.w - (float)hi[3])};
            const int c0 = 4 * cq, ks = c0 >> 5, kq = (c0 >> 3) & 3, j0 = c0 & 7;
            const int blk = ((m >> 4) * KS + ks) * 2;                     // hi block; lo block = blk + 1
            const int off = (frag_slot(kq, m & 15, ks & 1) * 8 + j0);    // halves inside the 1 KB block (swizzled slot)
            *reinterpret_cast<f16x4 *>(xh + (size_t)blk * 512 + off) = hi;
            *reinterpret_cast<f16x4 *>(xh + (size_t)(blk + 1) * 512 + off) = lo;
        }
    }
    // E mapping: matrix wave w owns the M-block PAIR emp = w & 1 (halo rows 32 emp .. 32 emp + 31) and the N-block enb = w >> 1 of
    // every chunk.  Its A fragments (the input tile: constant over the chunks) are read from LDS ONCE, below, and stay in
    // registers; per step it reads only its N-block's weight fragments (2 KS ds_read_b128 instead of 6 KS when a wave owned one
    // M-block and both N-blocks: the LDS pipe, shared by the CU's eight waves, is what a step waits for -- NOTES.md, round 4).
    const int emp = wave & 1, enb = (wave >> 1) & 1;
    float vmul[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // 1 where this lane's expand output (M-block 2 emp + h, row 4 lk + i) is a pixel of the image
    if (wave < 4) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = (2 * emp + h) * 16 + 4 * lk + i;
                const int hy = m / IW, hx = m - hy * IW;
                const int gy = gy0 + hy, gx = gx0 + hx;
                vmul[4 * h + i] = (m < NH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? 1.0f : 0.0f;
            }
        we_store(0);
        if (nch > 1) we_fetch(c0 + 1);                       // put in place at the start of step 0
    }
    __syncthreads();
    IR_STAMP(0, 1);
    u32x4 xa[2][KS][2];                                      // [M-block of the pair][k-step][hi | lo]
    if (wave < 4) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                xa[h][ks][0] = XsF[(((2 * emp + h) * KS + ks) * 2 + 0) * 64 + frag_lane(lane, ks & 1)];
                xa[h][ks][1] = XsF[(((2 * emp + h) * KS + ks) * 2 + 1) * 64 + frag_lane(lane, ks & 1)];
            }
    }

    f32x4 pacc[NJ], ppair[NJ], ptot[NJ];                     // the current leaf; the current pair of leaves; the tree so far
#pragma unroll
    for (int j = 0; j < NJ; ++j) pacc[j] = ppair[j] = ptot[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bias_nxt = a.be[c0 * CE + enb * 16 + lr];
    const int mbp = wave & 1, nbp = (wave >> 1) & 1;

    for (int it = 0; it < nch + 2; ++it) {
        if (wave < 4) {
            // Weight staging, one step ahead of its use and one step behind its loads: what was requested at the start of
            // the previous step (We(it+1), Wp(it-1)) has had a whole step to arrive and is put in place now, then the next
            // requests go out.  Done by the matrix waves: on the 16-bit MFMA their step is shorter than the service waves'.
#ifndef RPN_EXP_X3_NOFETCH       /* timing experiment only (results are wrong): no weight traffic inside the chunk loop */
            if (it + 1 < nch) we_store((it + 1) & 1);
            if (it >= 1 && it <= nch) wp_store((it - 1) & 1);
#endif
            // the expand bias of the NEXT chunk is requested BEFORE the weight fragments: vmcnt retires loads in order, and
            // the register copy that hands it to the next step sits at the end of E -- behind the weight requests it would
            // wait for all of them (measured: +1.1 k cycles per step)
            const float biasv = bias_nxt;
            if (it + 1 < nch) bias_nxt = a.be[(c0 + it + 1) * CE + enb * 16 + lr];
            __builtin_amdgcn_sched_barrier(0);
#ifndef RPN_EXP_X3_NOFETCH
            if (it + 2 < nch) we_fetch(c0 + it + 2);
            if (it < nch) wp_fetch(c0 + it);
#endif
            // operands of P(it-2) first: they are ready when the step starts, and their LDS latency then hides behind E
            u32x4 dhi, dlo, pb[NJ][2];
            if (it >= 2) {
                const u32x4 *dsf = DsF[it & 1];
                const u32x4 *wps = WpS[it & 1];
                dhi = dsf[(mbp * 2 + 0) * 64 + frag_lane(lane, 0)];
                dlo = dsf[(mbp * 2 + 1) * 64 + frag_lane(lane, 1)];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int nb = nbp + 2 * j;
                    pb[j][0] = wps[(nb * 2 + 0) * 64 + lane];
                    pb[j][1] = wps[(nb * 2 + 1) * 64 + lane];
                }
            }
            if (it < nch) {
                // ---- E(it): two M-blocks (32 halo pixels) x one N-block x KS k-steps x 3 MFMAs ------------------------
                const u32x4 *wes = WeS[it & 1];
                u32x4 wb[KS][2];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    wb[ks][0] = wes[((enb * KS + ks) * 2 + 0) * 64 + lane];
                    wb[ks][1] = wes[((enb * KS + ks) * 2 + 1) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4 eacc[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) eacc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int h = 0; h < 2; ++h) eacc[h] = mfma_x3(xa[h][ks][0], xa[h][ks][1], wb[ks][0], wb[ks][1], eacc[h]);
                float *es = Es[it & 1];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        es[((2 * emp + h) * 16 + 4 * lk + i) * SEP + enb * 16 + lr] =   // (* 1 or * 0: ReLU6's output is finite and >= 0)
                            relu6f(eacc[h][i] * a.scale_e + biasv) * vmul[4 * h + i];
            }
            IR_STAMP(0, 2 + 4 * it);
            if (it >= 2) {
                // ---- P(it-2): acc (32 px x COUT) += D (32 x 32) * Wp[chunk it-2] ---------------------------------------------
#pragma unroll
                for (int j = 0; j < NJ; ++j) pacc[j] = mfma_x3(dhi, dlo, pb[j][0], pb[j][1], pacc[j]);
                const int cg = c0 + it - 2;                     // the chunk just projected: a leaf ends behind every LEAF-th
                if ((cg + 1) % LEAF == 0) {
                    const int leaf = cg / LEAF;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        ppair[j] = ((leaf & 1) && ksplit != 6) ? ppair[j] + pacc[j] : pacc[j];    // (ksplit = 6: the one leaf, as it is)
                        if (leaf & 1) ptot[j] = leaf == 1 ? ppair[j] : ptot[j] + ppair[j];
                        pacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            IR_STAMP(0, 3 + 4 * it);
        } else {
            // Staging runs one step ahead of its use AND one step behind its loads: what was requested at the start of the
            // previous step (We(it+1), Wp(it-1), the depthwise weights of chunk it-1) has had a whole step to arrive and
            // is put in place now; then the requests for the next step go out; then the depthwise.
#pragma unroll
            for (int k = 0; k < 10; ++k) wd_cur[k] = wd_nxt[k];
            IR_STAMP(4, 66 + 4 * it);
            __builtin_amdgcn_sched_barrier(0);
#ifndef RPN_EXP_X3_NOWD
            if (it < nch) wd_fetch(c0 + it);
#endif
            IR_STAMP(4, 65 + 4 * it);
            if (it >= 1 && it <= nch) {
                // ---- D(it-1): depthwise 3x3 + bias + ReLU6, then hi / lo float16 into P's A-operand image ----------------
                const int c = it - 1;
                const float *es = Es[c & 1] + dn;
                const int py = wave - 4;
                constexpr int WW = (PXG - 1) * S + 3;
                float win[3][WW];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int x = 0; x < WW; ++x) win[r][x] = es[((py * S + r) * IW + dg * PXG * S + x) * SEP];
                __builtin_amdgcn_sched_barrier(0);
                IR_STAMP(4, 64 + 4 * it);
                unsigned *dw32 = reinterpret_cast<unsigned *>(DsF[c & 1]);
                const int kq = dn >> 3, jp = (dn & 7) >> 1, odd = dn & 1;
#pragma unroll
                for (int px = 0; px < PXG; ++px) {
                    float acc = wd_cur[9];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int q = 0; q < 3; ++q) acc = fmaf(win[r][px * S + q], wd_cur[r * 3 + q], acc);
                    const float v = relu6f(acc);
                    const _Float16 h = (_Float16)v;
                    const _Float16 l = (_Float16)(v - (float)h);
                    const unsigned hu = __builtin_bit_cast(unsigned short, h), lu = __builtin_bit_cast(unsigned short, l);
                    // lanes (2t, 2t+1) = channels (2t, 2t+1): the even lane writes the pair's hi dword, the odd lane its lo dword
                    // (neighbour exchange as a DPP quad permutation [1,0,3,2]: one vector instruction; __shfl_xor is a
                    // ds_bpermute with an LDS round trip behind each of the 4 outputs)
                    const unsigned got = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? hu : lu), 0xB1, 0xF, 0xF, false);
                    const unsigned word = odd ? (got | (lu << 16)) : (hu | (got << 16));
                    const int p = py * IR_TW + dg * PXG + px;
                    const int blk = (p >> 4) * 2 + odd;                  // [mb][hi|lo]
                    dw32[blk * 256 + frag_slot(kq, p & 15, odd) * 4 + jp] = word;
                }
            }
            IR_STAMP(4, 4 + 4 * it);
        }
        __syncthreads();
        IR_STAMP(0, 5 + 4 * it);
    }

    // the output epilogue's operands (projection bias, the residual = this tile's own input pixels) are requested here, in
    // front of the seam: their L2 round trip runs under the hand-off instead of behind it
    float obias[NJ], ores[NJ][4];
    if (wave < 4) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int co = (nbp + 2 * j) * 16 + lr;
            obias[j] = a.bp[co];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = mbp * 16 + 4 * lk + i;
                const int oy = oy0 + (p >> 3), ox = ox0 + (p & 7);
                ores[j][i] = 0.0f;
                if constexpr (RES)
                    if (oy < a.H && ox < a.W) ores[j][i] = a.x[(((size_t)img * a.H + oy) * a.W + ox) * KP + co];
            }
        }
    }
    if (ksplit > 1) {
        // ---- K-split: this workgroup's pair (ksplit = 3) or leaf (6) -> `part`; the last arriver adds them in tree order -------
        // Visibility WITHOUT cache-wide fences (MI355X_MICROARCH.md, inter-workgroup visibility, second form): every handed-off
        // word is written with a device-scope (sc1, write-through) store and read with a device-scope load; a storing wave
        // drains its stores (vmcnt(0)) in front of the workgroup barrier that precedes the ticket.  (First version: an
        // agent-scope release in front of the ticket and an acquire behind it -- an L2 write-back and an L2 invalidate per
        // workgroup, with hundreds of workgroups of the same kernel still computing out of that L2: configs[4] 0.384 -> 0.455 ms.)
        // The slab layout is the accumulator layout: lane l of matrix wave w keeps its f32x4 of N-block pair j at
        // [(j * 4 + w) * 64 + l] -- ONE 16-byte sc1 store per accumulator (1 KB contiguous per wave instruction) and one
        // 16-byte sc1 load per partial for the last arriver, all of them in flight together.  (First version: 4-byte
        // device-scope atomics, 8 stores + 48 dependent-free but scalar loads per lane: the seam cost 11.5 k cycles of a
        // 26 k-cycle workgroup at one image -- scalar sc1 accesses are one fabric transaction each.)
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
            a.part + (size_t)tile * ksplit * (32 * COUT), (short)0, ksplit * 32 * COUT * 4, 0x00020000);
        constexpr int kSc1 = 16;                                         // cache-policy bit 4 = sc1 (device scope)
        IR_STAMP(0, 100);
        if (wave < 4) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f32x4 v = (ksplit == 2 && part_h == 0) ? ptot[j] : ppair[j];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), prs,
                                                       (part_h * 32 * COUT + ((j * 4 + wave) * 64 + lane) * 4) * 4, 0, kSc1);
            }
        }
        __shared__ unsigned last_flag;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(a.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = ticket == (unsigned)ksplit - 1u ? 1u : 0u;
            if (last) __hip_atomic_store(a.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
            last_flag = last;
        }
        __syncthreads();
        IR_STAMP(0, 101);
        IR_STAMP_RT(0, 105);
        if (last_flag == 0u) return;
        if (wave < 4) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                f32x4 q[6];
#pragma unroll
                for (int h = 0; h < 6; ++h)
                    q[h] = h < ksplit ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            prs, (h * 32 * COUT + ((j * 4 + wave) * 64 + lane) * 4) * 4, 0, kSc1))
                                      : f32x4{0.f, 0.f, 0.f, 0.f};
                ptot[j] = ksplit == 2 ? q[0] + q[1]
                                      : (ksplit == 3 ? (q[0] + q[1]) + q[2] : ((q[0] + q[1]) + (q[2] + q[3])) + (q[4] + q[5]));
            }
        }
    }
    IR_STAMP(0, 102);
    if (wave < 4) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int co = (nbp + 2 * j) * 16 + lr;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = mbp * 16 + 4 * lk + i;
                const int oy = oy0 + (p >> 3), ox = ox0 + (p & 7);
                if (oy < a.H && ox < a.W) {
                    const size_t pix = ((size_t)img * a.H + oy) * a.W + ox;
                    float v = ptot[j][i] * a.scale_p + obias[j];
                    if constexpr (RES) v += ores[j][i];
                    a.out[pix * COUT + co] = v;
                }
            }
        }
    }
#ifdef RPN_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IR_STAMP(0, 103);
    IR_STAMP_RT(0, 105);
#endif
}

// (LABORATORY BUILD ONLY, -DRPN_LAB, RPN_MN_X3W=1: measured NOT faster -- blocks 7-9 17.5 -> 17.8 us at batch 8, 12.4 -> 11.5 us
// at one image; the 96-channel variants spill at the 128-register budget of 1024 threads and run 2x slower; blocks 4-5 lose
// residency (two workgroups per CU instead of four).  The step is not per-wave latency after all: ~820 of its ~1.9 k cycles
// are LDS-pipe occupancy (weight staging writes + every matrix wave re-reading the chunk's We) and ~420 VMEM issue, both
// per CU and unchanged by the wave count.  See NOTES.md.)
#ifdef RPN_LAB
// ---- the same block on SIXTEEN waves (round 4) ----------------------------------------------------------------------------
// The idea: ir_block_x3_kernel's steps are latency, not throughput: in-kernel stamps at batch 8 (scripts/ir_stamp_probe.py) show
// ~1.9 k cycles per chunk with the matrix pipe 6-10 % busy -- E is one M-block x 2 N-blocks per wave (12-18 dependent MFMAs,
// 8 epilogue stores), D four output pixels per lane (18 window reads, 36 FMAs, 4 conversions and stores), each a chain of
// LDS / vector round trips on ONE wave per SIMD and group.  Here the same tile, the same LDS images and the same products in
// the same order (bit-identical results) run on eight matrix + eight service waves: E = one (M-block, N-block) per wave, P =
// one (M-block, N-block quarter) per wave, D = two output pixels per lane; weight staging spread over 512 threads.
template <int KP, int CEXP, int COUT, bool RES>
__global__ void __launch_bounds__(2 * IR_THREADS)
ir_block_x3w_kernel(IrX3Args a)
{
    constexpr int S = 1, CE = 32, G = 8, THREADS = 2 * IR_THREADS;        // matrix waves 0 .. G-1, service waves G .. 2G-1
    constexpr int IH = IR_TH + 2, IW = IR_TW + 2, NH = IH * IW;           // 6 x 10 halo pixels
    constexpr int MB = 4, MH = 64;
    constexpr int KS = KP / 32;
    constexpr int NBE = CE / 16, NCHUNK = CEXP / CE;
    constexpr int SEP = CE + 4;
    constexpr int NBO = COUT / 16, NJ = (NBO + 3) / 4;                    // projection N-blocks; per wave (nb = nbq + 4 j)
    constexpr int WE_P = NBE * KS * 2 * 64, WP_P = NBO * 2 * 64;
    constexpr int SW = 64 * G;                                            // staging threads (the matrix waves)
    constexpr int NWE = (WE_P + SW - 1) / SW, NWP = (WP_P + SW - 1) / SW;
    static_assert(KP % 32 == 0 && CEXP % CE == 0 && COUT % 32 == 0 && (!RES || KP == COUT) && NBE == 2, "shape");

    __shared__ u32x4 XsF[MB * KS * 2 * 64];
    __shared__ __attribute__((aligned(16))) float Es[2][MH * SEP];
    __shared__ u32x4 DsF[2][2 * 2 * 64];
    __shared__ u32x4 WeS[2][WE_P];
    __shared__ u32x4 WpS[2][WP_P];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int stid = tid;
    constexpr int LEAF = NCHUNK / 6;                                      // (the projection tree: see ir_block_x3_kernel)
    static_assert(NCHUNK % 6 == 0, "the projection tree has six leaves");
    const int ksplit = a.ksplit;
    const int tile = (int)blockIdx.x / ksplit, part_h = (int)blockIdx.x - tile * ksplit;
    const int nch = ksplit == 2 ? (part_h == 0 ? 4 * LEAF : 2 * LEAF) : NCHUNK / ksplit;
    const int c0 = ksplit == 2 ? part_h * 4 * LEAF : part_h * nch;
    int t = tile;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int img = t / a.tiles_y;
    const int oy0 = ty * IR_TH, ox0 = tx * IR_TW;
    const int gy0 = oy0 - 1, gx0 = ox0 - 1;
    IR_STAMP(0, 0);
    IR_STAMP_RT(0, 104);
    if (wave >= G) __builtin_amdgcn_s_setprio(2);

    u32x4 we_st[NWE], wp_st[NWP];
    auto we_fetch = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWE; ++i)
            if (WE_P % SW == 0 || stid + SW * i < WE_P) we_st[i] = a.we[(size_t)c * WE_P + stid + SW * i];
    };
    auto we_store = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWE; ++i)
            if (WE_P % SW == 0 || stid + SW * i < WE_P) WeS[slot][stid + SW * i] = we_st[i];
    };
    auto wp_fetch = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWP; ++i)
            if (WP_P % SW == 0 || stid + SW * i < WP_P) wp_st[i] = a.wp[(size_t)c * WP_P + stid + SW * i];
    };
    auto wp_store = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWP; ++i)
            if (WP_P % SW == 0 || stid + SW * i < WP_P) WpS[slot][stid + SW * i] = wp_st[i];
    };
    // depthwise: lane = (channel dn, pixel pair): service wave sw -> output row sw / 2, pixels 2 * ((sw & 1) * 2 + lane / 32) + {0, 1}
    constexpr int PXG = 2;
    const int dn = lane & 31;
    const int dpy = (wave - G) >> 1, dx0 = ((((wave - G) & 1) << 1) | (lane >> 5)) * PXG;
    float wd_cur[10], wd_nxt[10];
    // depthwise weights of a chunk: ONE 48-byte record per channel -- 9 taps, the bias, 2 floats of padding (pack_ir_x3_dw) --
    // = three 16-byte loads per lane and step (ten 4-byte loads from the [9][CEXP] matrix + bias cost 13 % of blocks 4, 5, 7-9:
    // their ISSUE, ~660 cycles at the head of every step beside the matrix waves' weight requests)
    auto wd_fetch = [&](int c) __attribute__((always_inline)) {
        const float4 *rec = reinterpret_cast<const float4 *>(a.wd) + ((size_t)c * CE + dn) * 3;
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        wd_nxt[0] = q0.x; wd_nxt[1] = q0.y; wd_nxt[2] = q0.z; wd_nxt[3] = q0.w;
        wd_nxt[4] = q1.x; wd_nxt[5] = q1.y; wd_nxt[6] = q1.z; wd_nxt[7] = q1.w;
        wd_nxt[8] = q2.x; wd_nxt[9] = q2.y;
    };
    if (wave < G) we_fetch(c0);

    // ---- 1. input tile: float32 NHWC -> hi / lo float16, fragment-major (swizzled slots: frag_slot) ------------------------
    {
        constexpr int CQ = KP / 4;
        const float *__restrict__ xin = a.x + (size_t)img * a.H * a.W * KP;
        _Float16 *xh = reinterpret_cast<_Float16 *>(XsF);
        for (int idx = tid; idx < MH * CQ; idx += THREADS) {
            const int m = idx / CQ, cq = idx - m * CQ;
            const int hy = m / IW, hx = m - hy * IW;
            const int gy = gy0 + hy, gx = gx0 + hx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < NH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                v = *reinterpret_cast<const float4 *>(xin + ((size_t)gy * a.W + gx) * KP + 4 * cq);
            if (a.status && !(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) <= 65504.0f))
                atomicOr(a.status, 1u);
            const f16x4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
            const f16x4 lo = {(_Float16)(v.x - (float)hi[0]), (_Float16)(v.y - (float)hi[1]), (_Float16)(v.z - (float)hi[2]),
                              (_Float16)(v.w - (float)hi[3])};
            const int ch0 = 4 * cq, ks = ch0 >> 5, kq = (ch0 >> 3) & 3, j0 = ch0 & 7;
            const int blk = ((m >> 4) * KS + ks) * 2;
            const int off = (frag_slot(kq, m & 15, ks & 1) * 8 + j0);
            *reinterpret_cast<f16x4 *>(xh + (size_t)blk * 512 + off) = hi;
            *reinterpret_cast<f16x4 *>(xh + (size_t)(blk + 1) * 512 + off) = lo;
        }
    }
    // matrix wave w: E on (M-block emb, N-block enb); P on M-block mbp, N-blocks nbq + 4 j
    const int emb = wave & 3, enb = (wave >> 2) & 1;
    const int mbp = wave & 1, nbq = (wave >> 1) & 3;
    float vmul[4] = {0.f, 0.f, 0.f, 0.f};
    if (wave < G) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = emb * 16 + 4 * lk + i;
            const int hy = m / IW, hx = m - hy * IW;
            const int gy = gy0 + hy, gx = gx0 + hx;
            vmul[i] = (m < NH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? 1.0f : 0.0f;
        }
        we_store(0);
        if (nch > 1) we_fetch(c0 + 1);
    }
    __syncthreads();
    IR_STAMP(0, 1);

    f32x4 pacc[NJ], ppair[NJ], ptot[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) pacc[j] = ppair[j] = ptot[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bias_nxt = a.be[c0 * CE + enb * 16 + lr];

    for (int it = 0; it < nch + 2; ++it) {
        if (wave < G) {
            if (it + 1 < nch) we_store((it + 1) & 1);
            if (it >= 1 && it <= nch) wp_store((it - 1) & 1);
            const float biasv = bias_nxt;
            if (it + 1 < nch) bias_nxt = a.be[(c0 + it + 1) * CE + enb * 16 + lr];
            __builtin_amdgcn_sched_barrier(0);
            if (it + 2 < nch) we_fetch(c0 + it + 2);
            if (it < nch) wp_fetch(c0 + it);
            u32x4 dhi, dlo, pb[NJ][2];
            if (it >= 2) {
                const u32x4 *dsf = DsF[it & 1];
                const u32x4 *wps = WpS[it & 1];
                dhi = dsf[(mbp * 2 + 0) * 64 + frag_lane(lane, 0)];
                dlo = dsf[(mbp * 2 + 1) * 64 + frag_lane(lane, 1)];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int nb = nbq + 4 * j;
                    if (NBO % 4 == 0 || nb < NBO) {
                        pb[j][0] = wps[(nb * 2 + 0) * 64 + lane];
                        pb[j][1] = wps[(nb * 2 + 1) * 64 + lane];
                    }
                }
            }
            if (it < nch) {
                // ---- E(it): (M-block emb) x (N-block enb) x KS k-steps x 3 MFMAs ----------------------------------------
                const u32x4 *wes = WeS[it & 1];
                u32x4 xa[KS][2], wb[KS][2];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    xa[ks][0] = XsF[((emb * KS + ks) * 2 + 0) * 64 + frag_lane(lane, ks & 1)];
                    xa[ks][1] = XsF[((emb * KS + ks) * 2 + 1) * 64 + frag_lane(lane, ks & 1)];
                    wb[ks][0] = wes[((enb * KS + ks) * 2 + 0) * 64 + lane];
                    wb[ks][1] = wes[((enb * KS + ks) * 2 + 1) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4 eacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) eacc = mfma_x3(xa[ks][0], xa[ks][1], wb[ks][0], wb[ks][1], eacc);
                float *es = Es[it & 1];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    es[(emb * 16 + 4 * lk + i) * SEP + enb * 16 + lr] = relu6f(eacc[i] * a.scale_e + biasv) * vmul[i];
            }
            IR_STAMP(0, 2 + 4 * it);
            if (it >= 2) {
                // ---- P(it-2) ----------------------------------------------------------------------------------------------
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (NBO % 4 == 0 || nbq + 4 * j < NBO) pacc[j] = mfma_x3(dhi, dlo, pb[j][0], pb[j][1], pacc[j]);
                const int cg = c0 + it - 2;
                if ((cg + 1) % LEAF == 0) {
                    const int leaf = cg / LEAF;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        ppair[j] = ((leaf & 1) && ksplit != 6) ? ppair[j] + pacc[j] : pacc[j];
                        if (leaf & 1) ptot[j] = leaf == 1 ? ppair[j] : ptot[j] + ppair[j];
                        pacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            IR_STAMP(0, 3 + 4 * it);
        } else {
#pragma unroll
            for (int k = 0; k < 10; ++k) wd_cur[k] = wd_nxt[k];
            __builtin_amdgcn_sched_barrier(0);
            if (it < nch) wd_fetch(c0 + it);
            if (it >= 1 && it <= nch) {
                // ---- D(it-1): two output pixels per lane --------------------------------------------------------------------
                const int c = it - 1;
                const float *es = Es[c & 1] + dn;
                constexpr int WW = (PXG - 1) * S + 3;
                float win[3][WW];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int x = 0; x < WW; ++x) win[r][x] = es[((dpy * S + r) * IW + dx0 * S + x) * SEP];
                __builtin_amdgcn_sched_barrier(0);
                unsigned *dw32 = reinterpret_cast<unsigned *>(DsF[c & 1]);
                const int kq = dn >> 3, jp = (dn & 7) >> 1, odd = dn & 1;
#pragma unroll
                for (int px = 0; px < PXG; ++px) {
                    float acc = wd_cur[9];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int q = 0; q < 3; ++q) acc = fmaf(win[r][px * S + q], wd_cur[r * 3 + q], acc);
                    const float v = relu6f(acc);
                    const _Float16 h = (_Float16)v;
                    const _Float16 l = (_Float16)(v - (float)h);
                    const unsigned hu = __builtin_bit_cast(unsigned short, h), lu = __builtin_bit_cast(unsigned short, l);
                    const unsigned got = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? hu : lu), 0xB1, 0xF, 0xF, false);
                    const unsigned word = odd ? (got | (lu << 16)) : (hu | (got << 16));
                    const int p = dpy * IR_TW + dx0 + px;
                    const int blk = (p >> 4) * 2 + odd;
                    dw32[blk * 256 + frag_slot(kq, p & 15, odd) * 4 + jp] = word;
                }
            }
            IR_STAMP(G, 4 + 4 * it);
        }
        __syncthreads();
        IR_STAMP(0, 5 + 4 * it);
    }

    // epilogue operands in front of the seam (see ir_block_x3_kernel)
    float obias[NJ], ores[NJ][4];
    if (wave < G) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int nb = nbq + 4 * j;
            const int co = (nb < NBO ? nb : 0) * 16 + lr;
            obias[j] = a.bp[co];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = mbp * 16 + 4 * lk + i;
                const int oy = oy0 + (p >> 3), ox = ox0 + (p & 7);
                ores[j][i] = 0.0f;
                if constexpr (RES)
                    if (oy < a.H && ox < a.W) ores[j][i] = a.x[(((size_t)img * a.H + oy) * a.W + ox) * KP + co];
            }
        }
    }
    if (ksplit > 1) {
        // K-split seam: 16-byte sc1 slab stores / loads in accumulator layout (see ir_block_x3_kernel); slab slot of the
        // accumulator of (N-block nb, M-block mbp) = nb * 2 + mbp
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
            a.part + (size_t)tile * ksplit * (32 * COUT), (short)0, ksplit * 32 * COUT * 4, 0x00020000);
        constexpr int kSc1 = 16;
        IR_STAMP(0, 100);
        if (wave < G) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nb = nbq + 4 * j;
                if (NBO % 4 == 0 || nb < NBO) {
                    const f32x4 v = (ksplit == 2 && part_h == 0) ? ptot[j] : ppair[j];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), prs,
                                                           (part_h * 32 * COUT + ((nb * 2 + mbp) * 64 + lane) * 4) * 4, 0, kSc1);
                }
            }
        }
        __shared__ unsigned last_flag;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(a.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = ticket == (unsigned)ksplit - 1u ? 1u : 0u;
            if (last) __hip_atomic_store(a.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = last;
        }
        __syncthreads();
        IR_STAMP(0, 101);
        IR_STAMP_RT(0, 105);
        if (last_flag == 0u) return;
        if (wave < G) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nb = nbq + 4 * j;
                if (NBO % 4 == 0 || nb < NBO) {
                    f32x4 q[6];
#pragma unroll
                    for (int h = 0; h < 6; ++h)
                        q[h] = h < ksplit ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                prs, (h * 32 * COUT + ((nb * 2 + mbp) * 64 + lane) * 4) * 4, 0, kSc1))
                                          : f32x4{0.f, 0.f, 0.f, 0.f};
                    ptot[j] = ksplit == 2 ? q[0] + q[1]
                                          : (ksplit == 3 ? (q[0] + q[1]) + q[2] : ((q[0] + q[1]) + (q[2] + q[3])) + (q[4] + q[5]));
                }
            }
        }
    }
    IR_STAMP(0, 102);
    if (wave < G) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int nb = nbq + 4 * j;
            if (!(NBO % 4 == 0 || nb < NBO)) continue;
            const int co = nb * 16 + lr;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = mbp * 16 + 4 * lk + i;
                const int oy = oy0 + (p >> 3), ox = ox0 + (p & 7);
                if (oy < a.H && ox < a.W) {
                    const size_t pix = ((size_t)img * a.H + oy) * a.W + ox;
                    float v = ptot[j][i] * a.scale_p + obias[j];
                    if constexpr (RES) v += ores[j][i];
                    a.out[pix * COUT + co] = v;
                }
            }
        }
    }
#ifdef RPN_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IR_STAMP(0, 103);
    IR_STAMP_RT(0, 105);
#endif
}

#endif   // RPN_LAB

// ---- pointwise conv (1 x 1, stride 1) + bias + ReLU6 on the 16-bit MFMA, float32 NHWC in -> SPLIT16 out ------------------
// block_13_expand (96 -> 576, models/rpn_mobilenet_v2.py:17: the tap layer that feeds rpn_conv) under f16x3: the expand
// GEMM of ir_block_x3_kernel on its own (hi + lo float16 operands, lo*hi + hi*lo + hi*hi per product, float32 accumulation),
// weights in pack_ir_x3_expand's fragment-major image.  It replaces the float32-MFMA implicit GEMM there (1/16 of the rate:
// 27 us at batch 8, 11 us at one image).  A wave owns 16 pixels: its A fragments (a lane = (pixel, 8 channels): two 16-byte
// loads per k-step) are built straight from global memory and stay in registers; the workgroup's NT output channels' weight
// fragments are staged once in LDS (one linear copy) and read with linear, conflict-free ds_read_b128.  Epilogue: the wave's
// 16 x NT outputs go to LDS as SPLIT16 records ({hi[0:8], lo[0:8], hi[8:16], lo[8:16]} per 16 channels: even lane = the hi
// dword of a channel pair, odd lane = its lo dword, DPP pair exchange) over the weight tile, and leave as 16-byte stores of
// NT * 4 contiguous bytes per pixel.
struct PwX3Args {
    const float *x;       // (P, KP) float32
    const u32x4 *w;       // pack_ir_x3_expand(w [KP][COUT] * 2^shift)
    const float *bias;    // (COUT)
    unsigned *out;        // (P, COUT / 16) SPLIT16 records of 64 bytes
    float scale;          // 2^-shift
    long long P;
    int COUT;
    unsigned *status;     // float16 range flag, or null
};

template <int KP, int NT, int MW>
__global__ void __launch_bounds__(64 * MW)
pw_x3_kernel(PwX3Args a)
{
    constexpr int KS = KP / 32, NB = NT / 16;
    constexpr int WPIECES = NB * KS * 2 * 64;                              // 16-byte pieces of the weight tile
    constexpr int ROW_DW = NT + 4;                                          // a pixel's records + 16 bytes of padding: rows 4 apart fall into different banks
    constexpr int OUT_DW = 16 * ROW_DW;                                     // dwords of one wave's output records (16 px x NT ch x 4 B, padded)
    static_assert(KP % 32 == 0 && NT % 32 == 0, "shape");
    static_assert(MW * OUT_DW * 4 <= WPIECES * 16 || true, "");
    constexpr int LDS_PIECES = WPIECES > MW * OUT_DW / 4 ? WPIECES : MW * OUT_DW / 4;
    __shared__ u32x4 Ws[LDS_PIECES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int ntiles = a.COUT / NT;
    const int nt = (int)blockIdx.x % ntiles;
    const long long mt = (long long)blockIdx.x / ntiles;
    const long long p0 = (mt * MW + wave) * 16;                            // this wave's first pixel

    // weight tile: chunks [nt * NT / 32, (nt + 1) * NT / 32) of the packed image are one contiguous run
    {
        const u32x4 *src = a.w + (size_t)nt * WPIECES;
        u32x4 st[(WPIECES + 64 * MW - 1) / (64 * MW)];
#pragma unroll
        for (int i = 0; i < (WPIECES + 64 * MW - 1) / (64 * MW); ++i)
            if (WPIECES % (64 * MW) == 0 || tid + 64 * MW * i < WPIECES) st[i] = src[tid + 64 * MW * i];
        // A operand: lane (row lr, k-quarter lk) <- 8 consecutive channels of pixel p0 + lr per k-step (requested behind the
        // weight loads, in flight with them)
        float4 xa[KS][2];
        const long long px = p0 + lr;
        const bool live = px < a.P;
        const float *xr = a.x + (size_t)(live ? px : 0) * KP + lk * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            xa[ks][0] = live ? *reinterpret_cast<const float4 *>(xr + ks * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
            xa[ks][1] = live ? *reinterpret_cast<const float4 *>(xr + ks * 32 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < (WPIECES + 64 * MW - 1) / (64 * MW); ++i)
            if (WPIECES % (64 * MW) == 0 || tid + 64 * MW * i < WPIECES) Ws[tid + 64 * MW * i] = st[i];
        __syncthreads();

        u32x4 ahi[KS], alo[KS];
        bool bad = false;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float v[8] = {xa[ks][0].x, xa[ks][0].y, xa[ks][0].z, xa[ks][0].w, xa[ks][1].x, xa[ks][1].y, xa[ks][1].z, xa[ks][1].w};
            f16x8 h, l;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bad = bad || !(fabsf(v[j]) <= 65504.0f);
                h[j] = (_Float16)v[j];
                l[j] = (_Float16)(v[j] - (float)h[j]);
            }
            ahi[ks] = __builtin_bit_cast(u32x4, h);
            alo[ks] = __builtin_bit_cast(u32x4, l);
        }
        if (a.status && bad) atomicOr(a.status, 1u);

        f32x4 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int c = nb >> 1, h2 = nb & 1;                                // packed image: [chunk][nb 2][ks][hi|lo][lane]
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 bhi = Ws[(((c * 2 + h2) * KS + ks) * 2 + 0) * 64 + lane];
                const u32x4 blo = Ws[(((c * 2 + h2) * KS + ks) * 2 + 1) * 64 + lane];
                acc[nb] = mfma_x3(ahi[ks], alo[ks], bhi, blo, acc[nb]);
            }
        }
        __syncthreads();                                                       // the weight tile is dead: records go over it

        unsigned *rec = reinterpret_cast<unsigned *>(Ws) + wave * OUT_DW;      // [16 px][NT / 16 records][16 dwords]
        const int odd = lr & 1;
        const int dw = (lr >> 3) * 8 + odd * 4 + ((lr & 7) >> 1);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float bias = a.bias[nt * NT + nb * 16 + lr];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = relu6f(acc[nb][i] * a.scale + bias);
                const _Float16 h = (_Float16)v;
                const _Float16 l = (_Float16)(v - (float)h);
                const unsigned hu = __builtin_bit_cast(unsigned short, h), lu = __builtin_bit_cast(unsigned short, l);
                const unsigned got = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? hu : lu), 0xB1, 0xF, 0xF, false);
                const unsigned word = odd ? (got | (lu << 16)) : (hu | (got << 16));
                rec[(4 * lk + i) * ROW_DW + nb * 16 + dw] = word;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // (the wave reads back only what it wrote itself)
        constexpr int ROWP = NT / 4;                                           // 16-byte pieces per pixel
        const u32x4 *rec4 = reinterpret_cast<const u32x4 *>(rec);
#pragma unroll
        for (int t = 0; t < (16 * ROWP) / 64; ++t) {
            const int piece = lane + 64 * t;
            const int row = piece / ROWP, col = piece - row * ROWP;
            if (p0 + row < a.P)
                *reinterpret_cast<u32x4 *>(a.out + (size_t)(p0 + row) * a.COUT + nt * NT + col * 4) = rec4[row * (ROW_DW / 4) + col];
        }
    }
}

bool pw_x3_supported(int cin, int cout) { return cin == 96 && cout % 96 == 0; }

hipError_t launch_pw_x3(const float *x, long long P, int cin, int cout, const void *w, const float *bias, float scale,
                        void *out, unsigned *status, hipStream_t s)
{
    if (!pw_x3_supported(cin, cout) || P <= 0) return hipErrorInvalidValue;
    PwX3Args a{};
    a.x = x; a.w = reinterpret_cast<const u32x4 *>(w); a.bias = bias; a.out = reinterpret_cast<unsigned *>(out);
    a.scale = scale; a.P = P; a.COUT = cout; a.status = status;
    // small grids (one image): 64-pixel x 32-channel tiles, hundreds of short workgroups; else 128 x 96
    static const int pw_tile = RPN_LAB_KNOB("RPN_PW_TILE", 0);     // 1: 64 x 32 tiles always; 2: 128 x 96 always; 3: 128 x 32
    if (pw_tile == 3) {
        const long long blocks = (P + 127) / 128 * (cout / 32);
        hipLaunchKernelGGL((pw_x3_kernel<96, 32, 8>), dim3((unsigned)blocks), dim3(512), 0, s, a);
    } else if ((P <= 4096 && pw_tile != 2) || pw_tile == 1) {
        const long long blocks = (P + 63) / 64 * (cout / 32);
        hipLaunchKernelGGL((pw_x3_kernel<96, 32, 4>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    } else {
        const long long blocks = (P + 127) / 128 * (cout / 96);
        hipLaunchKernelGGL((pw_x3_kernel<96, 96, 8>), dim3((unsigned)blocks), dim3(512), 0, s, a);
    }
    return hipGetLastError();
}

// ---- f16x3 variant of the high-resolution blocks (blocks 1-3: Cin = 16 / 24) -------------------------------------------
// ir_block_hr_kernel with both GEMMs on v_mfma_f32_16x16x32_f16 (hi + lo float16 operands, three MFMAs per product, float32
// accumulation: the arithmetic of ir_block_x3_kernel).  rocprofv3 on the float32 form (profiles/r03_mn8_pmc.txt): the waves
// spend 44-48 % of their life stalled at ISSUE and the matrix pipe is busy 30-35 % -- the float32 MFMA (32 cycles per
// 16 x 16 x 4 step) shares the SIMD's issue with the depthwise's vector work.  Here
//   * the expand GEMM's K = Cin (16 / 24) is one 32-deep step (zero-padded): 3 MFMAs of 16 cycles per 16 px x 16 channels
//     instead of 4 / 6 of 32.  Its A operand -- the wave's halo pixels as hi / lo fragments -- is built ONCE from global
//     memory (a lane = (pixel, 8 channels): two 16-byte loads) and stays in registers across the chunks: no input tile in LDS;
//   * the depthwise (float32, vector ALU) splits its outputs into the projection's fragment-major A image (lane pairs
//     exchange halves by DPP: one dword store per output);
//   * the projection takes ceil(CE / 32) steps of 3 MFMAs per 16 x 16 block; B fragments of both GEMMs come straight from the
//     host-packed fragment-major images in global memory (L2-resident: every workgroup reads the same few KB).
// Same tiles (4 x 8 outputs, 256 threads, every wave in every phase, two barriers per chunk) and chunk sizes as the f32 form.
struct IrHrX3Args {
    const float *x;
    float *out;
    const u32x4 *we;      // [chunk][nb CE/16][hi|lo][64 lanes]: expand weights * 2^shift_e, K zero-padded to 32
    const u32x4 *wp;      // [chunk][nb COUTP/16][ks][hi|lo][64 lanes]: projection weights * 2^shift_p, chunk rows padded to 32 ks
    const float *be, *wd, *bd, *bp;
    float scale_e, scale_p;
    int B, H, W, OH, OW, pad, tiles_x, tiles_y;
    unsigned *status;
    // K tree (NLEAF > 1 instantiations: the stride-2 blocks 3 and 6, whose one-image grids are 128 / 32 tiles walking 9 / 4 chunks):
    // the projection is summed over NLEAF fixed runs of chunks -- (l0 + l1) + l2, or (l0 + l1) + (l2 + l3) -- at EVERY batch size;
    // `ksplit` (1 | 2 | NLEAF) workgroups per tile share the leaves, partial sums go to `part` ([tile][ksplit][32 px x COUTP]) and
    // the last arriver (tickets[tile], left at zero) finishes the same tree: same bits at every factor (as in ir_block_x3_kernel).
    int ksplit;
    float *part;
    unsigned *tickets;
};

template <int CIN, int CEXP, int CE, int COUT, int S, bool RES, int NLEAF = 1, int KSPLIT = 1>
__global__ void __launch_bounds__(HR_THREADS)
ir_block_hrx3_kernel(IrHrX3Args a)
{
    constexpr int TH = 4, NPX = TH * HR_T;                               // 32 output pixels
    constexpr int IH = (TH - 1) * S + 3, IW = (HR_T - 1) * S + 3, NH = IH * IW;
    constexpr int MB = (NH + 15) / 16, MH = MB * 16, MBW = (MB + 3) / 4;
    constexpr int SEP = CE + 4;
    constexpr int NBE = CE / 16, NCHUNK = CEXP / CE;
    constexpr int KSP = (CE + 31) / 32;                                  // projection k-steps per chunk
    constexpr int COUTP = (COUT + 15) / 16 * 16, NBO = COUTP / 16, NJ = (NBO + 1) / 2;
    constexpr int DG = CE == 16 ? 16 : (CE == 32 ? 8 : 4);               // depthwise pixel groups
    constexpr int PXG = NPX / DG;                                         // output pixels per depthwise thread
    constexpr int DROWS = PXG <= HR_T ? 1 : PXG / HR_T, DCOLS = PXG <= HR_T ? PXG : HR_T;
    static_assert(CIN % 8 == 0 && CIN <= 32 && SEP % 8 == 4 && CEXP % CE == 0 && CE % 16 == 0 && CE <= 48 && DG * CE <= HR_THREADS,
                  "layout");
    static_assert(!RES || (S == 1 && CIN == COUT), "residual needs stride 1 and Cin == Cout");
    static_assert((NLEAF == 1 || NLEAF == 3 || NLEAF == 4) && NCHUNK % NLEAF == 0, "projection tree");
    constexpr int LEAFC = NCHUNK / NLEAF;                                 // chunks per leaf
    // Halo-row pitch of the E -> D image (IWP > IW: a padded pitch, E then writes through per-lane row offsets `eoff`).  Tried for the
    // 48-channel chunks, whose depthwise lane groups used to straddle two pixel groups one halo row apart (8 banks mod 32: 15 % of
    // those instantiations' LDS cycles): a pitch of 12 | 18 pixels fixed the reads and broke the E writes of rows that cross a halo
    // row (12.4 %).  The depthwise mapping below (one pixel group per WAVE, 48 of its 64 lanes active) removes the straddle instead.
    constexpr int IWP = IW;
    constexpr int ES_ROWS = IWP == IW ? MH : ((MH + IW - 1) / IW) * IWP;
    static_assert(IWP >= IW, "pitch");
    __shared__ __attribute__((aligned(16))) float Es[ES_ROWS * SEP];
    __shared__ u32x4 DsF[2 * KSP * 2 * 64];                               // [mb 2][ks][hi|lo][lane]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    // (the split factor is a template parameter: with run-time chunk bounds the unsplit batch-8 form lost 6 % -- 21.3 -> 22.6 us)
    static_assert(KSPLIT == 1 || (NLEAF > 1 && (KSPLIT == NLEAF || (NLEAF == 4 && KSPLIT == 2))), "split factor");
    constexpr int ksplit = KSPLIT;
    const int tile = (int)blockIdx.x / ksplit, part_h = (int)blockIdx.x - tile * ksplit;
    const int c_begin = part_h * (NCHUNK / ksplit), c_end = c_begin + NCHUNK / ksplit;
    int t = tile;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int img = t / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * HR_T;
    const int gy0 = oy0 * S - a.pad, gx0 = ox0 * S - a.pad;              // halo origin on the input grid
    const int pmb = wave & 1, pnb0 = wave >> 1;                          // projection: M-block, first N-block (stride 2)
    const float *__restrict__ xin = a.x + (size_t)img * a.H * a.W * CIN;
    constexpr int HRV = CIN == 16 ? 0 : (S == 1 ? 1 : 2);                // (stamp slot of this variant)
    (void)HRV;
    HR_STAMP(HRV, 0);

    // the padding channels of the last projection k-step (CE = 16 / 48) are never written by the depthwise: zero once
    if constexpr (CE % 32 != 0)
        for (int i = tid; i < 2 * KSP * 2 * 64; i += HR_THREADS) DsF[i] = u32x4{0u, 0u, 0u, 0u};

    // ---- A operand of the expand GEMM: this wave's halo pixels (M-blocks wave, wave + 4, ...) as hi / lo fragments ----------
    u32x4 xh[MBW], xl[MBW];
    float vmul[MBW * 4];                 // 1 where this lane's expand output (row 4 lk + i of M-block mbi) is a pixel of the image, else 0
    int eoff[IWP == IW ? 1 : MBW * 4];   // (padded pitch only) Es offset of that output's halo pixel
    (void)eoff;
#pragma unroll
    for (int mbi = 0; mbi < MBW; ++mbi) {
        const int m = (wave + 4 * mbi) * 16 + lr;
        const int hy = m / IW, hx = m - hy * IW;
        const int gy = gy0 + hy, gx = gx0 + hx;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (m < NH && 8 * lk < CIN && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
            const float4 v0 = *reinterpret_cast<const float4 *>(xin + ((size_t)gy * a.W + gx) * CIN + 8 * lk);
            const float4 v1 = *reinterpret_cast<const float4 *>(xin + ((size_t)gy * a.W + gx) * CIN + 8 * lk + 4);
            v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
        }
        float mx = 0.0f;
        f16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mx = fmaxf(mx, fabsf(v[j]));
            hi[j] = (_Float16)v[j];
            lo[j] = (_Float16)(v[j] - (float)hi[j]);
        }
        if (a.status && !(mx <= 65504.0f)) atomicOr(a.status, 1u);       // (NaN included)
        xh[mbi] = __builtin_bit_cast(u32x4, hi);
        xl[mbi] = __builtin_bit_cast(u32x4, lo);
        // which of this lane's expand OUTPUTS (rows 4 lk + i of the M-block) are pixels inside the image
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mo = (wave + 4 * mbi) * 16 + 4 * lk + i;
            const int oy = mo / IW, ox = mo - oy * IW;
            vmul[mbi * 4 + i] = (mo < NH && gy0 + oy >= 0 && gy0 + oy < a.H && gx0 + ox >= 0 && gx0 + ox < a.W) ? 1.0f : 0.0f;
            if constexpr (IWP != IW) eoff[mbi * 4 + i] = (oy * IWP + ox) * SEP;
        }
    }
    // depthwise: channel, pixel group.  CE = 48: one group per wave (lanes 48 .. 63 idle) -- with tid / 48 a 32-lane LDS read group
    // covered the tail of one pixel group and the head of the next
    const int dc = CE == 48 ? (tid & 63) : tid % CE, dg = CE == 48 ? (tid >> 6) : tid / CE;
    const int dy0 = (dg * PXG) / HR_T, dx0 = (dg * PXG) % HR_T;
    f32x4 pacc[NJ];
    f32x4 tsum[NLEAF > 1 ? NJ : 1], usum[NLEAF == 4 ? NJ : 1];          // the tree so far; leaf 2 of a four-leaf tree
    (void)tsum; (void)usum;
#pragma unroll
    for (int j = 0; j < NJ; ++j) pacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    HR_STAMP(HRV, 1);

    for (int c = c_begin; c < c_end; ++c) {
        // ---- E(c): (halo pixels x 32) * We[:, chunk] -> * 2^-shift + bias, ReLU6, zero outside the image -> Es ------------
        {
            u32x4 wh[NBE], wl[NBE];
            float be_r[NBE];
#pragma unroll
            for (int nb = 0; nb < NBE; ++nb) {
                wh[nb] = a.we[((size_t)(c * NBE + nb) * 2 + 0) * 64 + lane];
                wl[nb] = a.we[((size_t)(c * NBE + nb) * 2 + 1) * 64 + lane];
                be_r[nb] = a.be[c * CE + nb * 16 + lr];
            }
#pragma unroll
            for (int mbi = 0; mbi < MBW; ++mbi) {
                const int mb = wave + 4 * mbi;
                if (MB % 4 == 0 || mb < MB) {
#pragma unroll
                    for (int nb = 0; nb < NBE; ++nb) {
                        const f32x4 e = mfma_x3(xh[mbi], xl[mbi], wh[nb], wl[nb], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            // (* 1 or * 0: ReLU6's output is finite and >= 0, so the product is the value itself or +0)
                            Es[(IWP == IW ? (mb * 16 + 4 * lk + i) * SEP : eoff[mbi * 4 + i]) + nb * 16 + lr] =
                                relu6f(e[i] * a.scale_e + be_r[nb]) * vmul[mbi * 4 + i];
                    }
                }
            }
        }
        __syncthreads();
        HR_STAMP(HRV, 2 + 3 * c);
        // ---- D(c): depthwise 3x3 + bias + ReLU6 (float32), hi / lo float16 into the projection's A image ---------------------
        if (dg < DG && dc < CE) {
            float wd_r[10];                  // one 48-byte record per channel (pack_ir_x3_dw): three 16-byte loads instead of ten 4-byte ones
            {
                const float4 *rec = reinterpret_cast<const float4 *>(a.wd) + ((size_t)c * CE + dc) * 3;
                const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
                wd_r[0] = q0.x; wd_r[1] = q0.y; wd_r[2] = q0.z; wd_r[3] = q0.w;
                wd_r[4] = q1.x; wd_r[5] = q1.y; wd_r[6] = q1.z; wd_r[7] = q1.w;
                wd_r[8] = q2.x; wd_r[9] = q2.y;
            }
            unsigned *dw32 = reinterpret_cast<unsigned *>(DsF);
            const int ks = dc >> 5, kq = (dc >> 3) & 3, jp = (dc & 7) >> 1, odd = dc & 1;
#pragma unroll
            for (int ry = 0; ry < DROWS; ++ry) {
                const int py = dy0 + ry;
                constexpr int WW = (DCOLS - 1) * S + 3;
                float win[3][WW];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int x = 0; x < WW; ++x) win[r][x] = Es[((py * S + r) * IWP + dx0 * S + x) * SEP + dc];
#pragma unroll
                for (int px = 0; px < DCOLS; ++px) {
                    float acc = wd_r[9];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int q = 0; q < 3; ++q) acc = fmaf(win[r][px * S + q], wd_r[r * 3 + q], acc);
                    const float v = relu6f(acc);
                    const _Float16 h = (_Float16)v;
                    const _Float16 l = (_Float16)(v - (float)h);
                    const unsigned hu = __builtin_bit_cast(unsigned short, h), lu = __builtin_bit_cast(unsigned short, l);
                    // lanes (2t, 2t+1) = channels (2t, 2t+1): the even lane writes the pair's hi dword, the odd lane its lo dword
                    const unsigned got = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? hu : lu), 0xB1, 0xF, 0xF, false);
                    const unsigned word = odd ? (got | (lu << 16)) : (hu | (got << 16));
                    const int p = py * HR_T + dx0 + px;
                    dw32[(((p >> 4) * KSP + ks) * 2 + odd) * 256 + frag_slot(kq, p & 15, odd, (ks & 1) << 1) * 4 + jp] = word;
                }
            }
        }
        __syncthreads();
        HR_STAMP(HRV, 3 + 3 * c);
        // ---- P(c): acc (16 px of M-block pmb x COUT) += D (16 x CE) * Wp[chunk] ---------------------------------------------
#pragma unroll
        for (int ks = 0; ks < KSP; ++ks) {
            const u32x4 dhi = DsF[((pmb * KSP + ks) * 2 + 0) * 64 + frag_lane(lane, 0, (ks & 1) << 1)],
                        dlo = DsF[((pmb * KSP + ks) * 2 + 1) * 64 + frag_lane(lane, 1, (ks & 1) << 1)];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nb = pnb0 + 2 * j;
                if (NBO % 2 == 0 || nb < NBO) {
                    const size_t blk = (((size_t)c * NBO + nb) * KSP + ks) * 2;
                    pacc[j] = mfma_x3(dhi, dlo, a.wp[blk * 64 + lane], a.wp[(blk + 1) * 64 + lane], pacc[j]);
                }
            }
        }
        // (the next chunk's E writes Es, its D writes DsF behind the barrier after E: P(c) has read DsF by then -- every wave
        // passes that barrier only after its own P(c))
        HR_STAMP(HRV, 4 + 3 * c);
        if constexpr (NLEAF > 1) {
            if ((c + 1) % LEAFC == 0) {          // a leaf ends: fold it into the tree in the fixed order
                const int leaf = c / LEAFC;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (ksplit == NLEAF || leaf == 0) tsum[j] = pacc[j];                          // (one leaf per workgroup: as it is)
                    else if (NLEAF == 4 && leaf == 2) { if (ksplit == 2) tsum[j] = pacc[j]; else usum[j] = pacc[j]; }
                    else if (NLEAF == 4 && leaf == 3) tsum[j] = ksplit == 2 ? tsum[j] + pacc[j] : tsum[j] + (usum[j] + pacc[j]);
                    else tsum[j] = tsum[j] + pacc[j];                                             // leaf 1; leaf 2 of three
                    pacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    }
    if constexpr (NLEAF > 1) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) pacc[j] = tsum[j];
        if (ksplit > 1) {
            // K-split seam: 16-byte sc1 slab stores / loads in accumulator layout (ir_block_x3_kernel); slab slot of the accumulator
            // of (N-block nb, M-block pmb) = nb * 2 + pmb
            const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
                a.part + (size_t)tile * ksplit * (32 * COUTP), (short)0, ksplit * 32 * COUTP * 4, 0x00020000);
            constexpr int kSc1 = 16;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nb = pnb0 + 2 * j;
                if (NBO % 2 == 0 || nb < NBO)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pacc[j]), prs,
                                                           (part_h * 32 * COUTP + ((nb * 2 + pmb) * 64 + lane) * 4) * 4, 0, kSc1);
            }
            __shared__ unsigned last_flag;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const unsigned ticket = __hip_atomic_fetch_add(a.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned last = ticket == (unsigned)ksplit - 1u ? 1u : 0u;
                if (last) __hip_atomic_store(a.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
                last_flag = last;
            }
            __syncthreads();
            if (last_flag == 0u) return;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nb = pnb0 + 2 * j;
                if (NBO % 2 == 0 || nb < NBO) {
                    f32x4 q[4];
#pragma unroll
                    for (int h = 0; h < 4; ++h)
                        q[h] = h < ksplit ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                prs, (h * 32 * COUTP + ((nb * 2 + pmb) * 64 + lane) * 4) * 4, 0, kSc1))
                                          : f32x4{0.f, 0.f, 0.f, 0.f};
                    pacc[j] = ksplit == 2 ? q[0] + q[1] : (ksplit == 3 ? (q[0] + q[1]) + q[2] : (q[0] + q[1]) + (q[2] + q[3]));
                }
            }
        }
    }
    // ---- * 2^-shift + bias (+ the block input) -> NHWC ---------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int nb = pnb0 + 2 * j;
        const int co = nb * 16 + lr;
        if ((NBO % 2 == 0 || nb < NBO) && co < COUT) {
            const float bias = a.bp[co];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = pmb * 16 + 4 * lk + i;
                const int py = p >> 3, px = p & 7;
                const int oy = oy0 + py, ox = ox0 + px;
                if (oy < a.OH && ox < a.OW) {
                    float v = pacc[j][i] * a.scale_p + bias;
                    if constexpr (RES) v += xin[((size_t)oy * a.W + ox) * CIN + co];      // (L2-hot: this tile just read it)
                    a.out[(((size_t)img * a.OH + oy) * a.OW + ox) * COUT + co] = v;
                }
            }
        }
    }
    HR_STAMP(HRV, 39);
}

// chunk size (expanded channels per step) of the high-resolution f16x3 blocks: the f32 form's
// block 6 (32 -> 192 -> 64, stride 2) on this kernel: RPN_MN_B6 = its chunk size (16 | 48), 0 = the float32 two-group kernel
static int ir_hrx3_b6()
{
    static const int v = RPN_LAB_KNOB("RPN_MN_B6", 48);
    return v;
}
static int ir_hrx3_b1()
{
    static const int v = RPN_LAB_KNOB("RPN_MN_B1CE", 32);              // block 1's chunk size: 16 | 32 | 48 (batch 8: 45.0 / 41.0 / 49.3 us)
    return v == 16 || v == 48 ? v : 32;
}
static int ir_hrx3_b3()
{
    static const int v = RPN_LAB_KNOB("RPN_MN_B3CE", 0);               // block 3's chunk size forced: 16 | 48 (0: by grid, ir_hrx3_chunk_for)
    return v == 48 ? 48 : 16;
}
static int ir_hrx3_ce(int cin, int stride, int ce_ov = 0)
{
    if (ce_ov) return ce_ov;
    return cin == 32 ? ir_hrx3_b6() : (cin == 16 ? ir_hrx3_b1() : ((cin == 24 && stride == 1) ? (RPN_LAB_KNOB("RPN_MN_B2CE", 48) == 16 ? 16 : 48) : ir_hrx3_b3()));
}
// Block 3 (24 -> 144 -> 32, stride 2) has two chunk sizes: 48 (three chunks = the tree's three leaves) on grids of at most 512
// tiles -- one 500 x 500 or 1024 x 1024 image: 13.2 -> 12.4 us, 19.4 -> 14.7 us -- and 16 (nine chunks: more resident workgroups)
// on larger ones (batch 8: 21.1 vs 24.1 us).  The chunk size shapes the packed weights AND the order in which a leaf's channels
// enter its sum, so it is chosen once per model handle, from the grid at the handle's max_batch -- never per call.
int ir_hrx3_chunk_for(int cin, int stride, long long tiles_at_max_batch)
{
    if (cin == 24 && stride == 2 && RPN_LAB_KNOB("RPN_MN_B3CE", 0) == 0) return tiles_at_max_batch <= 512 ? 48 : 16;
    return 0;                                          // (0: the kernel family's default for this block)
}

bool ir_block_hrx3_supported(int cin, int cexp, int cout, int stride, bool residual)
{
    return (cin == 16 && cexp == 96 && cout == 24 && stride == 2 && !residual) ||
           (cin == 24 && cexp == 144 && cout == 24 && stride == 1 && residual) ||
           (cin == 24 && cexp == 144 && cout == 32 && stride == 2 && !residual) ||
           (cin == 32 && cexp == 192 && cout == 64 && stride == 2 && !residual && ir_hrx3_b6() != 0) ||
           (cin == 32 && cexp == 192 && cout == 32 && stride == 1 && residual);     // (blocks 4, 5: only where ir_block_hrx3_preferred)
}

// Blocks 4 and 5 (32 -> 192 -> 32, stride 1, residual) run on BOTH f16x3 block kernels: the 512-thread two-group pipeline with
// its K split wins on small grids (one image: 12.0 vs 15.5 us at 500 x 500, 13.3 vs 15.7 at 1024 x 1024), this kernel's
// 256-thread workgroups (four per CU instead of two) on large ones (batch 8, 1024 tiles: 21.9 -> 18.0 us).  The two kernels
// sum in different orders, so -- like block 3's chunk size -- the choice is made once per model handle, from the grid at
// its max_batch.  RPN_MN_B45HR (laboratory): 0 never, 1 always.
bool ir_block_hrx3_preferred(int cin, int stride, long long tiles_at_max_batch)
{
    if (!(cin == 32 && stride == 1)) return true;
    const int forced = RPN_LAB_KNOB("RPN_MN_B45HR", -1);
    if (forced == 0 || forced == 1) return forced == 1;
    return tiles_at_max_batch >= 768;
}

size_t ir_hrx3_expand_floats(int cexp) { return (size_t)32 * cexp; }                      // K padded to 32: CEXP x 128 bytes
size_t ir_hrx3_project_floats(int cin, int cexp, int cout, int stride, int ce_ov)
{
    const int ce = ir_hrx3_ce(cin, stride, ce_ov), ksp = (ce + 31) / 32, nbo = (cout + 15) / 16;
    return (size_t)(cexp / ce) * nbo * ksp * 2 * 64 * 4;
}

void pack_ir_hrx3_expand(const float *w /* [cin][cexp] */, int cin, int cexp, int stride, int shift, unsigned short *dst, int ce_ov)
{
    const int CE = ir_hrx3_ce(cin, stride, ce_ov), NBE = CE / 16;
    const float mul = ldexpf(1.0f, shift);
    for (int c = 0; c < cexp / CE; ++c)
        for (int nb = 0; nb < NBE; ++nb)
            for (int ln = 0; ln < 64; ++ln)
                for (int j = 0; j < 8; ++j) {
                    const int k = 8 * (ln >> 4) + j, n = c * CE + nb * 16 + (ln & 15);
                    const float v = k < cin ? w[(size_t)k * cexp + n] * mul : 0.0f;
                    const _Float16 h = (_Float16)v;
                    const _Float16 l = (_Float16)(v - (float)h);
                    const size_t blk = (size_t)(c * NBE + nb) * 2;
                    memcpy(dst + (blk * 64 + ln) * 8 + j, &h, 2);
                    memcpy(dst + ((blk + 1) * 64 + ln) * 8 + j, &l, 2);
                }
}

void pack_ir_hrx3_project(const float *w /* [cexp][coutp] */, int cin, int cexp, int cout, int stride, int shift, unsigned short *dst, int ce_ov)
{
    const int CE = ir_hrx3_ce(cin, stride, ce_ov), KSP = (CE + 31) / 32, coutp = (cout + 15) / 16 * 16, NBO = coutp / 16;
    const float mul = ldexpf(1.0f, shift);
    for (int c = 0; c < cexp / CE; ++c)
        for (int nb = 0; nb < NBO; ++nb)
            for (int ks = 0; ks < KSP; ++ks)
                for (int ln = 0; ln < 64; ++ln)
                    for (int j = 0; j < 8; ++j) {
                        const int kk = 32 * ks + 8 * (ln >> 4) + j, n = nb * 16 + (ln & 15);
                        const float v = kk < CE ? w[(size_t)(c * CE + kk) * coutp + n] * mul : 0.0f;
                        const _Float16 h = (_Float16)v;
                        const _Float16 l = (_Float16)(v - (float)h);
                        const size_t blk = (((size_t)c * NBO + nb) * KSP + ks) * 2;
                        memcpy(dst + (blk * 64 + ln) * 8 + j, &h, 2);
                        memcpy(dst + ((blk + 1) * 64 + ln) * 8 + j, &l, 2);
                    }
}

// K-split factor of the stride-2 tree blocks (3: 24 -> 144 -> 32, three leaves; 6: 32 -> 192 -> 64, four) on a grid of `tiles`
// 4 x 8 tiles: one leaf per workgroup while tiles x leaves still fits the chip at its residency (256-thread workgroups, 4-5 per CU).
int ir_block_hrx3_ksplit(int cin, long long tiles, bool have_scratch)
{
    static const int forced = RPN_LAB_KNOB("RPN_MN_HRKS", -1);        // 1: never split (A/B timing)
    if (!have_scratch || forced == 1) return 1;
    if (cin == 32) return tiles <= 64 ? 4 : (tiles <= 160 ? 2 : 1);
    if (cin == 24) return tiles <= 160 ? 3 : 1;       // (512 tiles -- configs[4] -- x 3 no longer fit one round: 19.2 -> 23.3 us)
    return 1;
}

hipError_t launch_ir_block_hrx3(const float *x, int B, int H, int W, int cin, int cexp, int cout, int stride, bool residual,
                                int pad, int OH, int OW, const void *we, const float *be, const float *wd, const float *bd,
                                const void *wp, const float *bp, float scale_e, float scale_p, float *out, unsigned *status,
                                float *scratch, hipStream_t s, int ce_ov)
{
    if (!ir_block_hrx3_supported(cin, cexp, cout, stride, residual)) return hipErrorInvalidValue;
    IrHrX3Args a{};
    a.x = x; a.out = out; a.we = reinterpret_cast<const u32x4 *>(we); a.wp = reinterpret_cast<const u32x4 *>(wp);
    a.be = be; a.wd = wd; a.bd = bd; a.bp = bp; a.scale_e = scale_e; a.scale_p = scale_p;
    a.B = B; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.pad = pad; a.status = status;
    a.tiles_x = (OW + HR_T - 1) / HR_T;
    a.tiles_y = (OH + 3) / 4;
    const long long tiles = (long long)a.tiles_x * a.tiles_y * B;
    if (tiles <= 0 || tiles > 0x7fffffffll) return hipErrorInvalidValue;
    // blocks 3 and 6 are K TREES (IrHrX3Args::ksplit); the scratch is the fused x3 blocks' (partials, then 1024 tickets)
    const bool tree = stride == 2 && (cin == 32 || cin == 24);
    a.ksplit = tree ? ir_block_hrx3_ksplit(cin, tiles, scratch != nullptr) : 1;
    const int coutp = (cout + 15) / 16 * 16;
    if (a.ksplit > 1 && (tiles > 1024 || (size_t)tiles * a.ksplit * 32 * coutp > (size_t)128 * 6 * 32 * 96)) a.ksplit = 1;
    a.part = scratch;
    a.tickets = scratch ? reinterpret_cast<unsigned *>(scratch + (size_t)128 * 6 * 32 * 96) : nullptr;
    const long long nb = tiles * a.ksplit;
#define RPN_HRX3(KS_, ...) hipLaunchKernelGGL((ir_block_hrx3_kernel<__VA_ARGS__, KS_>), dim3((unsigned)nb), dim3(HR_THREADS), 0, s, a)
    if (cin == 32 && stride == 1) {                  // (blocks 4, 5 on large grids: ir_block_hrx3_preferred)
        a.ksplit = 1;
        hipLaunchKernelGGL((ir_block_hrx3_kernel<32, 192, 48, 32, 1, true>), dim3((unsigned)tiles), dim3(HR_THREADS), 0, s, a);
    } else if (cin == 32 && ir_hrx3_b6() == 48) {
        if (a.ksplit == 4) RPN_HRX3(4, 32, 192, 48, 64, 2, false, 4);
        else if (a.ksplit == 2) RPN_HRX3(2, 32, 192, 48, 64, 2, false, 4);
        else RPN_HRX3(1, 32, 192, 48, 64, 2, false, 4);
    } else if (cin == 32) {
        a.ksplit = 1;
        hipLaunchKernelGGL((ir_block_hrx3_kernel<32, 192, 16, 64, 2, false, 4>), dim3((unsigned)tiles), dim3(HR_THREADS), 0, s, a);
    }
    else if (cin == 16 && ir_hrx3_b1() == 32) hipLaunchKernelGGL((ir_block_hrx3_kernel<16, 96, 32, 24, 2, false>), dim3((unsigned)nb), dim3(HR_THREADS), 0, s, a);
    else if (cin == 16 && ir_hrx3_b1() == 48) hipLaunchKernelGGL((ir_block_hrx3_kernel<16, 96, 48, 24, 2, false>), dim3((unsigned)nb), dim3(HR_THREADS), 0, s, a);
    else if (cin == 16) hipLaunchKernelGGL((ir_block_hrx3_kernel<16, 96, 16, 24, 2, false>), dim3((unsigned)nb), dim3(HR_THREADS), 0, s, a);
    else if (stride == 1 && ir_hrx3_ce(cin, stride, ce_ov) == 16) hipLaunchKernelGGL((ir_block_hrx3_kernel<24, 144, 16, 24, 1, true>), dim3((unsigned)nb), dim3(HR_THREADS), 0, s, a);
    else if (stride == 1) hipLaunchKernelGGL((ir_block_hrx3_kernel<24, 144, 48, 24, 1, true>), dim3((unsigned)nb), dim3(HR_THREADS), 0, s, a);
    else if (ir_hrx3_ce(cin, stride, ce_ov) == 48) { if (a.ksplit == 3) RPN_HRX3(3, 24, 144, 48, 32, 2, false, 3); else RPN_HRX3(1, 24, 144, 48, 32, 2, false, 3); }
    else if (a.ksplit == 3) RPN_HRX3(3, 24, 144, 16, 32, 2, false, 3);
    else RPN_HRX3(1, 24, 144, 16, 32, 2, false, 3);
#undef RPN_HRX3
    return hipGetLastError();
}

bool ir_block_x3_supported(int cin, int cexp, int cout, int stride, bool residual)
{
    if (stride != 1 || cexp != 6 * cin) return false;
    static const int b45 = RPN_LAB_KNOB("RPN_MN_X3_B45", 1);     // blocks 4, 5 (32 -> 192 -> 32)
    return (cin == 64 && cout == 64 && residual) || (cin == 64 && cout == 96 && !residual) ||
           (cin == 96 && cout == 96 && residual) || (b45 && cin == 32 && cout == 32 && residual);
}

// Host packing of one 1x1 weight matrix W[k][n] (row-major, ld = n_total) into the fragment-major hi / lo images the
// kernel stages: chunked over `chunk_n` columns when `by_n` (expand: chunks of expanded channels = columns), else over
// `chunk_k` rows (projection: chunks of expanded channels = rows).
void pack_ir_x3_expand(const float *w /* [K][CEXP] */, int K, int CEXP, int shift, unsigned short *dst)
{
    const int CE = 32, KS = K / 32, NBE = 2;
    const float mul = ldexpf(1.0f, shift);
    for (int c = 0; c < CEXP / CE; ++c)
        for (int nb = 0; nb < NBE; ++nb)
            for (int ks = 0; ks < KS; ++ks)
                for (int ln = 0; ln < 64; ++ln)
                    for (int j = 0; j < 8; ++j) {
                        const int k = 32 * ks + 8 * (ln >> 4) + j, n = c * CE + nb * 16 + (ln & 15);
                        const float v = w[(size_t)k * CEXP + n] * mul;
                        const _Float16 h = (_Float16)v;
                        const _Float16 l = (_Float16)(v - (float)h);
                        const size_t blk = ((size_t)(c * NBE + nb) * KS + ks) * 2;
                        memcpy(dst + (blk * 64 + ln) * 8 + j, &h, 2);
                        memcpy(dst + ((blk + 1) * 64 + ln) * 8 + j, &l, 2);
                    }
}

// depthwise weights + bias of the f16x3 blocks (ir_block_x3_kernel): [CEXP][12] = 9 taps, bias, 0, 0 per channel
size_t ir_x3_dw_floats(int cexp) { return (size_t)12 * cexp; }
void pack_ir_x3_dw(const float *wd /* [9][CEXP] */, const float *bd /* [CEXP] */, int CEXP, float *dst)
{
    for (int c = 0; c < CEXP; ++c) {
        for (int k = 0; k < 9; ++k) dst[(size_t)c * 12 + k] = wd[(size_t)k * CEXP + c];
        dst[(size_t)c * 12 + 9] = bd[c];
        dst[(size_t)c * 12 + 10] = dst[(size_t)c * 12 + 11] = 0.0f;
    }
}

void pack_ir_x3_project(const float *w /* [CEXP][COUT] */, int CEXP, int COUT, int shift, unsigned short *dst)
{
    const int CE = 32, NBO = COUT / 16;
    const float mul = ldexpf(1.0f, shift);
    for (int c = 0; c < CEXP / CE; ++c)
        for (int nb = 0; nb < NBO; ++nb)
            for (int ln = 0; ln < 64; ++ln)
                for (int j = 0; j < 8; ++j) {
                    const int k = c * CE + 8 * (ln >> 4) + j, n = nb * 16 + (ln & 15);
                    const float v = w[(size_t)k * COUT + n] * mul;
                    const _Float16 h = (_Float16)v;
                    const _Float16 l = (_Float16)(v - (float)h);
                    const size_t blk = ((size_t)c * NBO + nb) * 2;
                    memcpy(dst + (blk * 64 + ln) * 8 + j, &h, 2);
                    memcpy(dst + ((blk + 1) * 64 + ln) * 8 + j, &l, 2);
                }
}

// K-split factor of a fused f16x3 block on a grid of `tiles` 4 x 8 tiles: 6 workgroups per tile when the tiles cover at
// most a sixth of the CUs (one 500 x 500 image: 32 tiles), 3 up to a third of them, else 1.  Speed only: the projection tree is the same for every factor (ir_block_x3_kernel).
int ir_block_x3_ksplit(long long tiles)
{
    static const int forced = RPN_LAB_KNOB("RPN_MN_KSPLIT", 0);       // 1 | 3 | 6: force a factor (A/B timing)
    if (forced == 1 || forced == 2 || forced == 3 || forced == 6) return forced;
    // ... as long as tiles x factor workgroups still fit the 256 CUs in one round (beyond that the parts queue up behind one
    // another and the split only adds its fixed costs: configs[4] at 3 x 128 workgroups 0.385 -> 0.41 ms)
    // The uneven 2-way form (leaves 0..3 | 4..5: 256 workgroups for configs[4]'s 128 tiles) pays since round 4's 16-byte seam:
    // configs[4] 0.323 -> 0.316 ms (blocks 11-12 22.8 -> 20.8 us); with the 4-byte seam of round 3 it measured no gain.
    static const int two = RPN_LAB_KNOB("RPN_MN_KS2", 1);             // 0: one workgroup per tile above 85 tiles
    return tiles <= 42 ? 6 : (tiles <= 85 ? 3 : ((two && tiles <= 128) ? 2 : 1));
}
size_t ir_block_x3_scratch_floats() { return (size_t)128 * 6 * 32 * 96 + 1024; }    // partials of <= 128 tiles x 6 parts x 32 px x 96 ch, then 1024 tickets

hipError_t launch_ir_block_x3(const float *x, int B, int H, int W, int cin, int cexp, int cout, bool residual,
                              const void *we, const float *be, const float *wd, const float *bd, const void *wp,
                              const float *bp, float scale_e, float scale_p, float *out, unsigned *status, float *scratch,
                              hipStream_t s)
{
    if (!ir_block_x3_supported(cin, cexp, cout, 1, residual)) return hipErrorInvalidValue;
    IrX3Args a{};
    a.x = x; a.out = out; a.we = reinterpret_cast<const u32x4 *>(we); a.wp = reinterpret_cast<const u32x4 *>(wp);
    a.be = be; a.wd = wd; a.bd = bd; a.bp = bp; a.scale_e = scale_e; a.scale_p = scale_p;
    a.B = B; a.H = H; a.W = W; a.status = status;
    a.tiles_x = (W + IR_TW - 1) / IR_TW;
    a.tiles_y = (H + IR_TH - 1) / IR_TH;
    const long long tiles = (long long)a.tiles_x * a.tiles_y * B;
    if (tiles <= 0 || tiles > 0x7fffffffll) return hipErrorInvalidValue;
    a.ksplit = scratch ? ir_block_x3_ksplit(tiles) : 1;
    if (a.ksplit > 1 && (tiles > 1024 || (size_t)tiles * a.ksplit * 32 * cout > (size_t)128 * 6 * 32 * 96)) a.ksplit = 1;
    a.part = scratch;
    a.tickets = scratch ? reinterpret_cast<unsigned *>(scratch + (size_t)128 * 6 * 32 * 96) : nullptr;
    const long long nblocks = tiles * a.ksplit;
#ifdef RPN_STAMP
    {
        const char *sel = getenv("RPN_IR_STAMP_OP");
        int sc = -1, ss = -1;
        a.stamp = sel && sscanf(sel, "%d,%d", &sc, &ss) == 2 && sc == cin && ss == 1 && (cin != 64 || cout == 64);
        // RPN_IR_STAMP_NTH=n,k: of the matching launches, only every k-th one starting with the n-th (one block of a forward)
        static long long matches = 0;
        const char *nth = getenv("RPN_IR_STAMP_NTH");
        int sn = 0, sk = 1;
        if (a.stamp && nth && sscanf(nth, "%d,%d", &sn, &sk) == 2 && sk > 0) a.stamp = (matches++ % sk) == sn;
    }
#endif
#ifdef RPN_LAB
    static const int wide = RPN_LAB_KNOB("RPN_MN_X3W", 0);        // 1: the sixteen-wave form (same bits, measured not faster)
    if (wide) {
        if (cin == 32) hipLaunchKernelGGL((ir_block_x3w_kernel<32, 192, 32, true>), dim3((unsigned)nblocks), dim3(2 * IR_THREADS), 0, s, a);
        else if (cin == 64 && cout == 64) hipLaunchKernelGGL((ir_block_x3w_kernel<64, 384, 64, true>), dim3((unsigned)nblocks), dim3(2 * IR_THREADS), 0, s, a);
        else if (cin == 64) hipLaunchKernelGGL((ir_block_x3w_kernel<64, 384, 96, false>), dim3((unsigned)nblocks), dim3(2 * IR_THREADS), 0, s, a);
        else hipLaunchKernelGGL((ir_block_x3w_kernel<96, 576, 96, true>), dim3((unsigned)nblocks), dim3(2 * IR_THREADS), 0, s, a);
        return hipGetLastError();
    }
#endif
    if (cin == 32) hipLaunchKernelGGL((ir_block_x3_kernel<32, 192, 32, true>), dim3((unsigned)nblocks), dim3(IR_THREADS), 0, s, a);
    else if (cin == 64 && cout == 64) hipLaunchKernelGGL((ir_block_x3_kernel<64, 384, 64, true>), dim3((unsigned)nblocks), dim3(IR_THREADS), 0, s, a);
    else if (cin == 64) hipLaunchKernelGGL((ir_block_x3_kernel<64, 384, 96, false>), dim3((unsigned)nblocks), dim3(IR_THREADS), 0, s, a);
    else hipLaunchKernelGGL((ir_block_x3_kernel<96, 576, 96, true>), dim3((unsigned)nblocks), dim3(IR_THREADS), 0, s, a);
    return hipGetLastError();
}

// RPN_MN_HR=0: blocks 1-3 on the two-group pipeline kernel instead of ir_block_hr_kernel
static int ir_hr_mode()
{
    static const int on = RPN_LAB_KNOB("RPN_MN_HR", 1);
    return on;
}

bool ir_block_supported(int cin, int cexp, int cout, int stride, bool residual)
{
    struct Cfg { int cin, cexp, cout, s, res; };
    static const Cfg table[] = {{16, 96, 24, 2, 0},  {24, 144, 24, 1, 1}, {24, 144, 32, 2, 0}, {32, 192, 32, 1, 1},
                                {32, 192, 64, 2, 0}, {64, 384, 64, 1, 1}, {64, 384, 96, 1, 0}, {96, 576, 96, 1, 1}};
    for (const Cfg &c : table)
        if (c.cin == cin && c.cexp == cexp && c.cout == cout && c.s == stride && (c.res != 0) == residual) return true;
    return false;
}

// x: (B,H,W,cin) [stem: the (B,H,W,3) image]; weights as in IrArgs; pad = top/left zero padding of the depthwise
// (stem: of Conv1).  Returns hipErrorInvalidValue for a block shape outside ir_block_supported().
hipError_t launch_ir_block(const float *x, int B, int H, int W, int cin, int cexp, int cout, int stride, bool residual,
                           bool stem, int pad, int OH, int OW, const float *we, const float *be, const float *wd,
                           const float *bd, const float *wp, const float *bp, float *out, hipStream_t s)
{
    IrArgs a{};
    a.x = x; a.out = out; a.we = we; a.be = be; a.wd = wd; a.bd = bd; a.wp = wp; a.bp = bp;
    a.B = B; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.pad = pad;
    a.DH = H; a.DW = W;
    if (stem) {                                   // Conv1: 3x3 stride 2 'valid' behind ZeroPadding2D((pad, 1))
        a.DH = (H + pad + 1 - 3) / 2 + 1;
        a.DW = (W + pad + 1 - 3) / 2 + 1;
    }
    a.tiles_x = (OW + IR_TW - 1) / IR_TW;
    a.tiles_y = (OH + IR_TH - 1) / IR_TH;
    const long long nblocks = (long long)a.tiles_x * a.tiles_y * B;
    if (nblocks <= 0 || nblocks > 0x7fffffffll) return hipErrorInvalidValue;
#ifdef RPN_STAMP
    {   // RPN_IR_STAMP_OP="cin,stride" selects the block that records stamps in a -DRPN_STAMP build ("3,1": the stem)
        const char *sel = getenv("RPN_IR_STAMP_OP");
        int sc = -1, ss = -1;
        a.stamp = sel && sscanf(sel, "%d,%d", &sc, &ss) == 2 && sc == cin && ss == stride && (cin != 64 || cout == 64);
    }
#endif
#define RPN_IR(KP_, CEXP_, CE_, COUT_, S_, RES_, STEM_)                                                              \
    hipLaunchKernelGGL((ir_block_kernel<KP_, CEXP_, CE_, COUT_, S_, RES_, STEM_>), dim3((unsigned)nblocks),          \
                       dim3(IR_THREADS), 0, s, a)
    if (stem) {
        if (!(cin == 3 && cexp == 32 && cout == 16 && stride == 1 && !residual)) return hipErrorInvalidValue;
        static const int stem8 = RPN_LAB_KNOB("RPN_MN_STEM8", 1);     // 0: the 4 x 8 tile pipeline
        if (stem8) {
            const int th = stem8 == 4 ? 4 : 8;      // (4 x 16 tiles, six workgroups per CU: no faster at batch 8, slower at 1024 x 1024)
            a.tiles_x = (OW + ST_TW - 1) / ST_TW;
            a.tiles_y = (OH + th - 1) / th;
            const long long nb8 = (long long)a.tiles_x * a.tiles_y * B;
            if (nb8 <= 0 || nb8 > 0x7fffffffll) return hipErrorInvalidValue;
            if (th == 8) hipLaunchKernelGGL(stem_block_kernel<8>, dim3((unsigned)nb8), dim3(ST_THREADS), 0, s, a);
            else hipLaunchKernelGGL(stem_block_kernel<4>, dim3((unsigned)nb8), dim3(ST_THREADS), 0, s, a);
        } else {
            RPN_IR(28, 32, 32, 16, 1, false, true);
        }
    } else if (!ir_block_supported(cin, cexp, cout, stride, residual)) {
        return hipErrorInvalidValue;
    } else if (ir_hr_mode() && (cin == 16 || cin == 24) &&
               (stride == 1 || cin == 16 || (long long)((OW + HR_T - 1) / HR_T) * ((OH + 3) / 4) * B >= 512)) {
        // blocks 1-3 (16 / 24 input channels at 250 x 250 / 125 x 125) on the 4 x 8-tile, every-wave-in-every-phase kernel
        // (six 256-thread workgroups per CU).  Batch 8, 500 x 500: 0.094 -> 0.059, 0.067 -> 0.055, 0.045 -> 0.035 ms (block 3
        // only on grids of >= 512 tiles: at one 500 x 500 image the pipeline kernel is faster, 0.017 vs 0.021 ms).  Blocks
        // 4-6 (32 channels in, smaller grids) are as fast or faster on the two-group pipeline: they stay there.
        a.tiles_x = (OW + HR_T - 1) / HR_T;
        a.tiles_y = (OH + 3) / 4;
        const long long nbh = (long long)a.tiles_x * a.tiles_y * B;
        if (nbh <= 0 || nbh > 0x7fffffffll) return hipErrorInvalidValue;
#define RPN_IRHR(CIN_, CEXP_, CE_, COUT_, S_, RES_)                                                                    \
    hipLaunchKernelGGL((ir_block_hr_kernel<CIN_, CEXP_, CE_, COUT_, S_, RES_, 4>), dim3((unsigned)nbh), dim3(HR_THREADS), 0, s, a)
        if (cin == 16) RPN_IRHR(16, 96, 16, 24, 2, false);       // (32-channel chunks: fewer workgroups per CU, slower)
        else if (stride == 1) RPN_IRHR(24, 144, 48, 24, 1, true);
        else RPN_IRHR(24, 144, 16, 32, 2, false);
#undef RPN_IRHR
    } else if (cin == 16) RPN_IR(16, 96, 16, 24, 2, false, false);
    else if (cin == 24 && stride == 1) RPN_IR(24, 144, 48, 24, 1, true, false);
    else if (cin == 24) RPN_IR(24, 144, 16, 32, 2, false, false);
    else if (cin == 32 && stride == 1) RPN_IR(32, 192, 32, 32, 1, true, false);
    else if (cin == 32) RPN_IR(32, 192, 48, 64, 2, false, false);
    else if (cin == 64 && cout == 64) RPN_IR(64, 384, 48, 64, 1, true, false);
    else if (cin == 64) RPN_IR(64, 384, 48, 96, 1, false, false);
    else RPN_IR(96, 576, 48, 96, 1, true, false);
#undef RPN_IR
    return hipGetLastError();
}

}  // namespace rpn
