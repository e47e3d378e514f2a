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
// and is shared by every workgroup, so it lives in L2 / L1).  46 KB (stride 1, Cin = 96) .. 62 KB (stride 2) of LDS
// per workgroup: 2-3 workgroups per CU overlap one another's load / depthwise phases with MFMA work.
//
// STEM variant: the "expand" stage is Conv1 (3x3 stride 2, 3 -> 32 channels, K = 27 padded to 28) computed from an
// im2col tile gathered from the image, followed by expanded_conv's depthwise and its 32 -> 16 projection: stem +
// block 0 in one launch.
//
// Roofline: MFMA-bound on the exact-f32 MFMA (157.3 TFLOP/s); algorithmic flops per output pixel =
// 2*Cin*Cexp/s^2... (expand runs on the block's INPUT pixels) + 18*Cexp + 2*Cexp*Cout; the halo makes the expand stage
// compute (6*10)/(4*8) = 1.9x (stride 1) / (9*17)/(8*16) = 1.2x its algorithmic work.
#include "conv_kernels.h"

namespace rpn {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int IR_TH = 4, IR_TW = 8;           // output tile
constexpr int IR_THREADS = 256;

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
};

__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.0f), 6.0f); }

// KP: reduction length of the expand stage (Cin; STEM: 28); CE: expanded channels per chunk; S: depthwise stride.
template <int KP, int CEXP, int CE, int COUT, int S, bool RES, bool STEM>
__global__ void __launch_bounds__(IR_THREADS)
ir_block_kernel(IrArgs a)
{
    constexpr int SX = KP + 2;                               // Xs row stride: 2 * odd
    constexpr int IH = (IR_TH - 1) * S + 3, IW = (IR_TW - 1) * S + 3;
    constexpr int NH = IH * IW;                              // halo pixels
    constexpr int MB = (NH + 15) / 16;                       // M-blocks of the expand GEMM
    constexpr int MH = MB * 16;
    constexpr int MBW = (MB + 3) / 4;                        // M-blocks per wave
    constexpr int SE = (MH + 31) / 32 * 32 + 4;              // Es row stride = 4 mod 32
    constexpr int NBE = CE / 16;                             // N-blocks of the expand GEMM per chunk
    constexpr int NCHUNK = CEXP / CE;
    constexpr int SD = CE + 2;                               // Ds row stride: 2 * odd
    constexpr int COUTP = (COUT + 15) / 16 * 16;
    constexpr int NBO = COUTP / 16;                          // N-blocks of the projection
    constexpr int NJ = (NBO + 1) / 2;                        // ... per wave (waves split 2 M-blocks x 2 N-block parities)
    static_assert(KP % 4 == 0 && (SX / 2) % 2 == 1 && (SD / 2) % 2 == 1 && CEXP % CE == 0 && CE % 16 == 0, "layout");
    static_assert(!RES || (S == 1 && KP == COUT && !STEM), "residual needs stride 1 and Cin == Cout");

    __shared__ __attribute__((aligned(16))) float Xs[MH * SX];
    __shared__ __attribute__((aligned(16))) float Es[CE * SE];
    __shared__ __attribute__((aligned(16))) float Ds[IR_TH * IR_TW * SD];
    __shared__ float Wds[10 * CE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    int t = blockIdx.x;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int img = t / a.tiles_y;
    const int oy0 = ty * IR_TH, ox0 = tx * IR_TW;
    const int dpad = STEM ? 1 : a.pad;
    const int gy0 = oy0 * S - dpad, gx0 = ox0 * S - dpad;   // halo origin on the depthwise's input grid

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
        // im2col of Conv1: row m = pixel (gy, gx) of Conv1's output, column k = (r*3 + q)*3 + c <- image pixel
        // (2*gy - pad + r, 2*gx - pad + q), channel c; zero outside the image (ZeroPadding2D) and for k = 27
        const float *__restrict__ xin = a.x + (size_t)img * a.H * a.W * 3;
        for (int idx = tid; idx < MH * KP; idx += IR_THREADS) {
            const int m = idx / KP, k = idx - m * KP;
            const int hy = m / IW, hx = m - hy * IW;
            const int gy = gy0 + hy, gx = gx0 + hx;
            float v = 0.0f;
            if (m < NH && k < 27 && gy >= 0 && gy < a.DH && gx >= 0 && gx < a.DW) {
                const int r = k / 9, q = (k / 3) % 3, c = k % 3;
                const int iy = 2 * gy - a.pad + r, ix = 2 * gx - a.pad + q;
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = xin[((size_t)iy * a.W + ix) * 3 + c];
            }
            Xs[m * SX + k] = v;
        }
    }
    // which of this lane's expand outputs (M-block mbi of the wave, row 4*lk + i) are pixels inside the image
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
    __syncthreads();

    f32x4 pacc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) pacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int mbp = wave & 1, nbp = wave >> 1;             // projection: this wave's M-block and N-block parity

    for (int c = 0; c < NCHUNK; ++c) {
        // depthwise weights + bias of the chunk -> LDS (read as broadcasts below)
        for (int i = tid; i < 10 * CE; i += IR_THREADS) {
            const int r = i / CE, n = i - r * CE;
            Wds[i] = r < 9 ? a.wd[r * CEXP + c * CE + n] : a.bd[c * CE + n];
        }
        // ---- 2a. expand: (halo pixels x KP) * (KP x CE) ---------------------------------------------------------
        f32x4 eacc[MBW][NBE];
#pragma unroll
        for (int mbi = 0; mbi < MBW; ++mbi)
#pragma unroll
            for (int nb = 0; nb < NBE; ++nb) eacc[mbi][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *__restrict__ wec = a.we + (size_t)lk * CEXP + c * CE + lr;
#pragma unroll
        for (int kk = 0; kk < KP / 4; ++kk) {
            float bfr[NBE];
#pragma unroll
            for (int nb = 0; nb < NBE; ++nb) bfr[nb] = wec[(size_t)(4 * kk) * CEXP + nb * 16];
#pragma unroll
            for (int mbi = 0; mbi < MBW; ++mbi) {
                const int mb = wave + 4 * mbi;
                if (MB % 4 == 0 || mb < MB) {
                    const float af = Xs[(mb * 16 + lr) * SX + 4 * kk + lk];
#pragma unroll
                    for (int nb = 0; nb < NBE; ++nb)
                        eacc[mbi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bfr[nb], eacc[mbi][nb], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int mbi = 0; mbi < MBW; ++mbi) {
            const int mb = wave + 4 * mbi;
            if (MB % 4 == 0 || mb < MB) {
#pragma unroll
                for (int nb = 0; nb < NBE; ++nb) {
                    const int n = nb * 16 + lr;
                    const float bias = a.be[c * CE + n];
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        v[i] = ((vmask >> (mbi * 4 + i)) & 1u) ? relu6f(eacc[mbi][nb][i] + bias) : 0.0f;
                    *reinterpret_cast<f32x4 *>(&Es[n * SE + mb * 16 + 4 * lk]) = v;
                }
            }
        }
        __syncthreads();
        // ---- 2b. depthwise 3x3 on the chunk: lanes = the 32 output pixels, 2 channels per wave and step ----------
        {
            const int p = lane & 31, py = p >> 3, px = p & 7;
            const int base = (py * S) * IW + px * S;
#pragma unroll
            for (int it = 0; it < CE / 8; ++it) {
                const int n = wave * 2 + (lane >> 5) + 8 * it;
                const float *e = &Es[n * SE + base];
                float acc = Wds[9 * CE + n];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int q = 0; q < 3; ++q) acc = fmaf(e[r * IW + q], Wds[(r * 3 + q) * CE + n], acc);
                Ds[p * SD + n] = relu6f(acc);
            }
        }
        __syncthreads();
        // ---- 2c. projection: acc (32 px x COUT) += D (32 x CE) * Wp[chunk] --------------------------------------
        if (NBO >= 2 || nbp == 0) {
            const float *__restrict__ wpc = a.wp + (size_t)(c * CE + lk) * COUTP + lr;
#pragma unroll
            for (int kk = 0; kk < CE / 4; ++kk) {
                const float af = Ds[(mbp * 16 + lr) * SD + 4 * kk + lk];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int nb = nbp + 2 * j;
                    if (NBO % 2 == 0 || nb < NBO) {
                        const float bf = wpc[(size_t)(4 * kk) * COUTP + nb * 16];
                        pacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, pacc[j], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- 3. + bias (+ residual from the input tile) -> NHWC ----------------------------------------------------------
    if (NBO >= 2 || nbp == 0) {
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
#define RPN_IR(KP_, CEXP_, CE_, COUT_, S_, RES_, STEM_)                                                              \
    hipLaunchKernelGGL((ir_block_kernel<KP_, CEXP_, CE_, COUT_, S_, RES_, STEM_>), dim3((unsigned)nblocks),          \
                       dim3(IR_THREADS), 0, s, a)
    if (stem) {
        if (!(cin == 3 && cexp == 32 && cout == 16 && stride == 1 && !residual)) return hipErrorInvalidValue;
        RPN_IR(28, 32, 32, 16, 1, false, true);
    } else if (!ir_block_supported(cin, cexp, cout, stride, residual)) {
        return hipErrorInvalidValue;
    } else if (cin == 16) RPN_IR(16, 96, 48, 24, 2, false, false);
    else if (cin == 24 && stride == 1) RPN_IR(24, 144, 48, 24, 1, true, false);
    else if (cin == 24) RPN_IR(24, 144, 48, 32, 2, false, false);
    else if (cin == 32 && stride == 1) RPN_IR(32, 192, 48, 32, 1, true, false);
    else if (cin == 32) RPN_IR(32, 192, 48, 64, 2, false, false);
    else if (cin == 64 && cout == 64) RPN_IR(64, 384, 48, 64, 1, true, false);
    else if (cin == 64) RPN_IR(64, 384, 48, 96, 1, false, false);
    else RPN_IR(96, 576, 48, 96, 1, true, false);
#undef RPN_IR
    return hipGetLastError();
}

}  // namespace rpn
