// conv_split_kernels.hip -- 3x3 / stride-1 / 'same' convolution on the 16-bit matrix cores with
// float32-grade accuracy ("x3 split"), for gfx950 (MI355X).
//
// Why: the exact-f32 MFMA tops out at 157 TFLOP/s, i.e. ~1000 images/s for VGG16+RPN at 500x500
// (SURVEY.md H1); the 16-bit MFMA is 16x faster.  Every float32 operand x is carried as two
// 16-bit halves hi = rn16(x), lo = rn16(x - hi), and each product is formed as
//        a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (3 MFMAs, float32 accumulation)
// dropping only a_lo*b_lo.  With bfloat16 halves the product error is ~2^-16 relative, with
// float16 halves ~2^-21 (weights are pre-scaled by a power of two so their low halves stay
// normal); the error is measured against a CPU float64 reference in tests/, never assumed.
//
// Layouts ("SPLIT16"): activations [B][H][W][C/16][64 B], one 64-byte record per pixel and
// 16-channel slice = {hi[0:8], lo[0:8], hi[8:16], lo[8:16]} as four 16-byte pieces; weights
// [C/16][tap][cout_pad][64 B] with the same record per output channel.  A 16-byte piece is
// exactly one lane's A (or B) fragment of v_mfma_f32_32x32x16_{bf16,f16}.
//
// Kernel: implicit GEMM, M = TH x 32 output pixels of one image, N = 64 or 128 output
// channels, K walked as (16-channel slice) x (9 taps).  Per slice the (TH+2) x 34 input halo is
// staged in LDS ONCE and re-read by all 9 taps at shifted addresses (the im2col matrix never
// exists); per tap only the BN x 64 B weight tile is streamed.  Both are double-buffered through
// registers (global loads for step s+1 are issued before the MFMAs of step s), one barrier per
// tap.  LDS records are XOR-swizzled by (row >> 2) & 3 so that every ds_read_b128 fragment
// read is bank-conflict free.  Epilogue: bias + activation, then the tile is transposed
// through LDS so each lane stores whole 16-byte pieces (256 contiguous bytes per pixel).
//
// Roofline: MFMA-bound; 3 MFMAs per algorithmic product, so the algorithmic peak is
// (dense 16-bit MFMA peak, 2.5 PFLOP/s) / 3.
//
// Three kernel families share that design (DESIGN.md 4.1 has the measurements that chose between them):
//   conv3x3_split_kernel        32x32x16 MFMA, 16-channel slices, register-staged          (Cin = 64 -> 64: block1_conv2)
//   conv3x3_split16_kernel      16x16x32 MFMA, 32-channel slices, register-staged          (odd slice counts)
//   conv3x3_split16_dma_kernel  16x16x32 MFMA, LDS-DMA staging, one barrier per tap, persistent workgroups whose tap
//                               stream runs on across tiles                                 (11 of VGG16's 13 3x3 layers)
#include "conv_kernels.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

// The next slice's global loads are issued at the TOP of a barrier interval so that a whole interval of MFMAs hides
// their latency.  Left alone the scheduler sinks them to the end of the interval, right in front of the barrier whose
// far side stores them to LDS (measured in the ISA), exposing the full memory latency every interval: pin them.
#ifndef RPN_PIN
#define RPN_PIN 1
#endif
#if RPN_PIN
#define RPN_PIN_LOADS() __builtin_amdgcn_sched_barrier(0)
#else
#define RPN_PIN_LOADS() ((void)0)
#endif

// Debug build only (-DRPN_STAMP, scripts/stamp_probe.py): in-kernel cycle stamps of the 16x16x32 kernel's phases.
#ifdef RPN_STAMP
__device__ unsigned long long g_rpn_stamps[8192 * 32];
#define RPN_STAMP_AT(k)                                                                                    \
    do {                                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < 8192 && (k) < 32)                                             \
            g_rpn_stamps[blockIdx.x * 32 + (k)] = __builtin_readcyclecounter();                            \
    } while (0)
#define RPN_STAMP_VAL(k, v)                                                                                \
    do {                                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < 8192) g_rpn_stamps[blockIdx.x * 32 + (k)] = (v);             \
    } while (0)
extern "C" int rpn_debug_read_stamps(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rpn_stamps), (size_t)n * 8);
}
#else
#define RPN_STAMP_AT(k) ((void)0)
#define RPN_STAMP_VAL(k, v) ((void)0)
#endif

namespace rpn {

// The epilogue's staging area is private to a wave: lanes exchange data through it, other waves never touch it, and
// a wave's LDS instructions execute in order -- so the exchange needs a wave-level fence, not a workgroup barrier
// (which would make every wave wait for the slowest one twice per output row).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// (workgroups are 64 * NW threads; NW = 4 or 8)
constexpr int TWS = 32;                 // tile width in pixels = one MFMA M-block
constexpr int HW = TWS + 2;             // halo width
constexpr int kStagePad = 4;            // floats of padding per staged row (keeps 16-byte alignment)

template <bool F16> struct Half;
template <> struct Half<false> {
    using vec = bf16x8;
    using elem = __bf16;
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                       0, 0, 0);
    }
};
template <> struct Half<true> {
    using vec = f16x8;
    using elem = _Float16;
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0,
                                                      0, 0);
    }
};

// 8 floats -> one 16-byte piece: the hi halves, or the lo halves (x - hi)
// `status` (may be null): device word that receives RPN_STATUS_F16_RANGE when an activation does not fit float16
// (|x| > 65504, or non-finite): its hi half would become inf and every later layer silently garbage (ReLU turns the
// resulting NaNs into zeros).  Checked once per piece, by the lane that produces the hi halves; bf16 halves have
// float32's range.
template <bool F16>
__device__ __forceinline__ uint4 split_piece(const float (&x)[8], bool lo, unsigned *status = nullptr)
{
    using E = typename Half<F16>::elem;
    typename Half<F16>::vec v;
    if constexpr (F16) {
        if (status && !lo) {
            const float m = fmaxf(fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3]))),
                                  fmaxf(fmaxf(fabsf(x[4]), fabsf(x[5])), fmaxf(fabsf(x[6]), fabsf(x[7]))));
            float nan_probe = 0.0f;                           // fmaxf drops NaNs: x * 0 accumulates to NaN for any NaN / inf
#pragma unroll
            for (int k = 0; k < 8; ++k) nan_probe = fmaf(x[k], 0.0f, nan_probe);
            if (!(m <= 65504.0f) || nan_probe != 0.0f) atomicOr(status, 1u /* RPN_STATUS_F16_RANGE */);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const E h = (E)x[k];
        v[k] = lo ? (E)(x[k] - (float)h) : h;
    }
    return __builtin_bit_cast(uint4, v);
}

template <bool F16>
__device__ __forceinline__ float join(unsigned short hi, unsigned short lo)
{
    using E = typename Half<F16>::elem;
    return (float)__builtin_bit_cast(E, hi) + (float)__builtin_bit_cast(E, lo);
}

// v_mfma_f32_16x16x32: A row = lane & 15, k = 8 * (lane >> 4) .. + 7; B column = lane & 15, same k; C/D column = lane & 15,
// row = 4 * (lane >> 4) + reg
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c)
{
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// XOR swizzle of the eight 16-byte pieces of halo pixel hx (16x16x32 kernels): physical slot = logical piece ^ halo_swz(hx).
// A fragment read (ds_read_b128) is served in four groups of sixteen lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the
// same + 32 (MI355X_MICROARCH.md, LDS) -- i.e. sixteen consecutive pixels, eight of them with k-group g and eight with g ^ 1, and
// the tap's column shift s3 = 0, 1, 2 moves that window by a pixel.  With hx & 6 the sixteen pieces of every group fall on sixteen
// different 16-byte bank groups for all three shifts and both 16-pixel halves (exhaustive check and search:
// scripts/lds_swizzle_search.py).  (Rounds 1-5 used (hx >> 1) & 7, which is conflict-free for s3 = 0 only: the taps with
// s3 = 1, 2 paid 2-way conflicts on half of their groups -- the 24-31 % SQ_LDS_BANK_CONFLICT share of profiles/r05_f16x3_pmc.txt;
// round 6: 0.0-0.8 %, profiles/r06_f16x3_pmc.txt.)
__device__ __forceinline__ int halo_swz(int hx) { return hx & 6; }

__device__ __forceinline__ int xcd_remap_s(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

struct SplitConvArgs {
    const uint4 *x;       // SPLIT16 input  [B][H][W][Cin/16][4]
    const uint4 *w;       // split weights  [Cin/16][9][cout_pad][4]
    const float *bias;    // (Cout) float32, may be null
    void *out;            // SPLIT16 [B][H][W][Cout/16][4]  or float32 NHWC [B][H][W][Cout]
    int B, H, W, Cin, Cout, cout_pad;
    float out_scale;      // 2^-s when the weights were pre-scaled by 2^s
    int act, out_f32;
    unsigned *sched;      // persistent kernel: 17 zero-initialised counters (per XCD label: tile queue, exits; labels done), or null
    unsigned *status;     // float16 range flag of the owning model (split_piece), or null
    // B1 instantiation only (VGG16 block 1 in one launch: block1_conv1 computed into the halo tile, x unused):
    const float *img;     // float32 NHWC image (B, H, W, 3)
    const uint4 *w1;      // first-layer weights [64][8 pieces] (pack_weights_cin3_mfma_host)
    const float *b1;      // first-layer bias (64)
    float scale1;         // 2^-s of the pre-scaled first-layer weights
    // conv3x3_split16_kernel only: split-K over gridDim.y workgroups per tile.  Workgroup y accumulates the 32-channel
    // slices [y * chunks / gridDim.y, (y + 1) * chunks / gridDim.y) and writes its RAW partial sums (x out_scale; no bias,
    // no activation) as float32 NHWC to slab y of `out` (slab_floats apart); the consumer adds the slabs.
    long long slab_floats;
    // K TREE (rpn_conv, whose consumer -- the RPN head -- can add partial-sum slabs): the 32-channel slices are cut into FOUR
    // fixed leaves at slices k * slices / 4, k = 0 .. 4 (18 slices: 4 5 4 5), each leaf is one accumulation chain, and the layer's
    // value is (l0 + l1) + (l2 + l3) -- whoever adds: one workgroup inside its tile loop (conv3x3_split16_dma_kernel<.., 64,
    // true>, any batch), two workgroups writing l0 + l1 and l2 + l3 as slabs, or four writing one leaf each, the head adding
    // them in that order.  The same bits at every split factor, so the factor may follow the batch size.
    int ktree;
};

// TH: tile height (4 or 8 rows of 32 pixels); WN: waves along N (2 -> BN = 128, 1 -> BN = 64).
// One barrier interval ("step") = one 16-channel slice x one FILTER ROW (3 taps): 3 x MI x NI x 3 MFMAs per
// wave between barriers, and the LDS fragment reads of tap s+1 overlap the MFMAs of tap s.
// BBUF: weight-tile buffers in LDS (2 = double buffered, one barrier per step; 1 = single buffered with a
// second barrier, used where two buffers would not leave room for two workgroups per CU).
//
// B1 (VGG16 block 1 in ONE launch: block1_conv1 -> block1_conv2 -> block1_pool, models/rpn_vgg16.py:16): the input is
// the float32 image, and the halo tile of each 16-channel slice of block1_conv1's output is COMPUTED into LDS instead of
// being staged from HBM -- the 64-channel full-resolution tensor (512 MB at batch 8: written once, read 1.1 times) never
// exists.  Per tile the (TH+4) x 36 x 3 image patch is staged once; every wave builds the im2col operand (K = 27 -> 32)
// of its share of the 340 halo pixels once, as hi/lo fragments in registers (B operand of v_mfma_16x16x32, so that a
// lane's 4 results are 4 consecutive CHANNELS of one pixel = one 8-byte run of the pixel's LDS record), and per slice
// multiplies them with that slice's 16 x 32 weight fragment: 3 MFMAs + bias/ReLU/split + 2 ds_write_b64 per 16 pixels,
// two such blocks per barrier interval, written to the halo buffer the loop is not reading.  Halo pixels outside the
// image are zeros (block1_conv2 pads block1_conv1's OUTPUT).  +8 % MFMA work on block1_conv2 (halo recompute included).
template <int TH, int WN, int BBUF, int NW, bool F16, bool POOL, bool B1 = false>
__global__ void __launch_bounds__(64 * NW, NW / 2)
conv3x3_split_kernel(SplitConvArgs a, int tiles_x, int tiles_y, int n_tiles)
{
    constexpr int NT = 64 * NW;              // threads per workgroup (NW = 4 or 8 waves)
    constexpr int WM = NW / WN;              // waves along M
    constexpr int MI = TH / WM;              // 32-pixel rows per wave
    constexpr int NI = 2;                    // 32-channel blocks per wave
    constexpr int BN = WN * NI * 32;
    constexpr int HP = (TH + 2) * HW;        // halo pixels
    constexpr int A_PIECES = HP * 4;
    constexpr int A_ROUNDS = (A_PIECES + NT - 1) / NT;
    constexpr int A_RPS = (A_ROUNDS + 2) / 3;                         // halo rounds issued per step
    constexpr int B_PIECES = 3 * BN * 4;                              // one filter row of weight records
    constexpr int B_ROUNDS = B_PIECES / NT;
    static_assert(MI * WM == TH && B_PIECES % NT == 0, "tile shape");
    constexpr int STAGE_LD = 64 + kStagePad;                         // floats per staged row
    constexpr int ABUF = HP * 4 + 4;                                 // one halo buffer + a dummy slot for idle lanes
    constexpr int PW = TWS + 4;                                      // B1: image patch columns (halo of the halo)
    constexpr int PATCH_F = B1 ? (TH + 4) * PW * 3 : 0;              // B1: floats of the image patch
    static_assert(!B1 || (WN == 1 && POOL && BBUF == 2), "B1: the 64 -> 64 pooled instantiations only");
    constexpr int LDS_PIPE = 2 * ABUF + BBUF * B_PIECES;             // uint4 units
    constexpr int LDS_STAGE = (NW * 32 * STAGE_LD * 4 + 15) / 16;    // uint4 units (NW waves x 32 rows)
    constexpr int LDS_UINT4 = (LDS_PIPE > LDS_STAGE ? LDS_PIPE : LDS_STAGE) + (PATCH_F + 3) / 4;

    __shared__ uint4 lds[LDS_UINT4];
    // native vector type for everything staged: struct copies of HIP's uint4 lower to memcpy, which keeps the
    // staging registers in scratch memory across a barrier
    u32x4 *As = reinterpret_cast<u32x4 *>(lds);               // [2][ABUF]
    u32x4 *Bs = reinterpret_cast<u32x4 *>(lds) + 2 * ABUF;    // [BBUF][3][BN][4]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lm = lane & 31, kh = lane >> 5;

    // XCD-aware tile order (speed only): hardware deals workgroup ids round-robin over the 8 XCDs, so
    // blockIdx % 8 labels the XCD.  The 8 XCDs form an XN x XM grid; XCD (xn, xm) owns the N-tiles
    // nt = xn (mod XN) and the M-tiles mt = xm (mod XM), so the weight slice it streams (K x BN x |nt set|,
    // 1.2-2.4 MB for the 512-channel layers) stays resident in its private 4 MB L2 instead of all
    // 9.4 MB of weights passing through every L2.  Workgroups past the padded grid exit at once.
    int nt, mt;
    {
        const int m_tiles = tiles_x * tiles_y * a.B;
        const int XN = n_tiles >= 8 ? 8 : (n_tiles >= 4 ? 4 : (n_tiles >= 2 ? 2 : 1)), XM = 8 / XN;
        const int NTl = (n_tiles + XN - 1) / XN;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        nt = (slot % NTl) * XN + (xcd % XN);
        mt = (slot / NTl) * XM + (xcd / XN);
        if (nt >= n_tiles || mt >= m_tiles) return;
    }
    const int tx = mt % tiles_x;
    mt /= tiles_x;
    const int ty = mt % tiles_y;
    const int img = mt / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TWS, n0 = nt * BN;

    const int chunks = a.Cin >> 4;
    const int steps = chunks * 3;
    const size_t in_pix_stride = (size_t)chunks * 4;                     // uint4 per input pixel
    const uint4 *__restrict__ xin = B1 ? nullptr : a.x + (size_t)img * a.H * a.W * in_pix_stride;

    // ---- global -> register staging: every per-thread address is loop-invariant ----------------------
    // Halo pieces go through a raw buffer descriptor over THIS image: an out-of-image piece gets an offset
    // beyond num_records and the hardware range check returns zeros (no branch, no exec masking); the
    // 16-channel slice is selected by the scalar offset.  Weight pieces are plain global loads at
    // (uniform tile base) + (loop-invariant per-thread offset); cout_pad is a multiple of BN, no bounds check.
    constexpr int A_SLOTS = 3 * A_RPS;                                 // halo rounds per slice (some may be empty)
    constexpr unsigned kOob = 0x80000000u;                            // > any per-image tensor size (< 2 GiB)
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(xin), (short)0, (int)((size_t)a.H * a.W * in_pix_stride * 16), 0x00020000);
    unsigned a_goff[A_SLOTS];          // byte offset of this thread's piece of slice 0 (or kOob)
    int a_loff[A_SLOTS];               // uint4 index inside one halo buffer (dummy slot HP*4 when unused)
#pragma unroll
    for (int R = 0; R < (B1 ? 0 : A_SLOTS); ++R) {
        const int e = R * NT + tid;
        const int pix = e >> 2, pc = e & 3;
        const int hy = pix / HW, hx = pix - hy * HW;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        const bool piece = e < A_PIECES;
        const bool inimg = piece && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        a_loff[R] = piece ? (pix * 4 + (pc ^ ((hx >> 2) & 3))) : HP * 4;
        a_goff[R] = inimg ? (unsigned)((((size_t)iy * a.W + ix) * in_pix_stride + pc) * 16) : kOob;
    }
    u32x4 b_reg[B_ROUNDS];
    u32x4 a_reg[A_RPS];
    // (written as macros, not lambdas: a by-reference capture keeps the staging arrays in scratch memory)
#define RPN_LOAD_B(STEP)                                                                               \
    {                                                                                                  \
        const u32x4 *src_ = reinterpret_cast<const u32x4 *>(a.w) + ((size_t)(STEP) * 3 * a.cout_pad + n0) * 4;   \
        _Pragma("unroll") for (int i_ = 0; i_ < B_ROUNDS; ++i_) {                                      \
            const int e_ = tid + i_ * NT;                                                              \
            const int t_ = e_ / (BN * 4), rem_ = e_ - t_ * (BN * 4);          /* powers of two */      \
            b_reg[i_] = src_[(size_t)t_ * a.cout_pad * 4 + rem_];                                      \
        }                                                                                              \
    }
#define RPN_STORE_B(BUF)                                                                               \
    {                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < B_ROUNDS; ++i_) {                                      \
            const int e_ = tid + i_ * NT;                                                              \
            const int t_ = e_ / (BN * 4), rem_ = e_ - t_ * (BN * 4);                                   \
            const int n_ = rem_ >> 2, pc_ = rem_ & 3;                                                  \
            Bs[(((BUF) * 3 + t_) * BN + n_) * 4 + (pc_ ^ ((n_ >> 2) & 3))] = b_reg[i_];                \
        }                                                                                              \
    }
#define RPN_LOAD_A(CHUNK, R) \
    __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, a_goff[R], (CHUNK) * 64, 0))

    // ---- B1: first-layer state (all dead code otherwise) ------------------------------------------------------
    constexpr int NBLK = (HP + 15) / 16;                              // 16-pixel blocks of the halo tile
    constexpr int QB = B1 ? (NBLK + NW - 1) / NW : 1;                 // blocks per wave: block = wave + NW * q
    constexpr int QPS = (QB + 2) / 3;                                 // blocks computed per barrier interval
    float *patchf = reinterpret_cast<float *>(lds + (LDS_PIPE > LDS_STAGE ? LDS_PIPE : LDS_STAGE));
    const int lr = lane & 15, kg = lane >> 4;
    u32x4 phi[QB], plo[QB];            // im2col fragments (8 of the 32 k of pixel lr): hi / lo halves
    int pdst[QB];                      // byte offset of this lane's 8-byte run inside a halo buffer (no pixel: the dummy slot)
    float b1_max = 0.0f;               // largest first-layer activation this lane produced (float16 range check)
    unsigned pin = 0;                  // bit q: the pixel is inside the image
    u32x4 w1hi = {0, 0, 0, 0}, w1lo = {0, 0, 0, 0};
    f32x4 b1v = {0.f, 0.f, 0.f, 0.f};
    (void)patchf; (void)lr; (void)kg; (void)pin; (void)phi; (void)plo; (void)pdst; (void)w1hi; (void)w1lo; (void)b1v;
    (void)b1_max;
#define RPN_B1_LOADW(CHUNK)                                                                            \
    {                                                                                                  \
        const u32x4 *w1_ = reinterpret_cast<const u32x4 *>(a.w1) + ((CHUNK) * 16 + lr) * 8 + kg;       \
        w1hi = w1_[0];                                                                                 \
        w1lo = w1_[4];                                                                                 \
        b1v = *reinterpret_cast<const f32x4 *>(a.b1 + (CHUNK) * 16 + 4 * kg);                          \
    }
    // block Q of the slice whose weights are in w1hi / w1lo -> halo buffer BUF.  Branch-free, so that the scheduler may
    // place this work among the tap MFMAs of the interval: lanes without a pixel (the tail of the last block, the
    // blocks past it) store to the buffer's dummy slot, and the float16 range check is one max per block, tested once
    // at the end of the kernel.
#define RPN_B1_BLOCK(Q, BUF)                                                                           \
    {                                                                                                  \
        f32x4 c_ = {0.f, 0.f, 0.f, 0.f};                                                               \
        c_ = mfma16<F16>(w1lo, phi[Q], c_);                                                            \
        c_ = mfma16<F16>(w1hi, plo[Q], c_);                                                            \
        c_ = mfma16<F16>(w1hi, phi[Q], c_);                                                            \
        using E_ = typename Half<F16>::elem;                                                           \
        using h4_ = __attribute__((ext_vector_type(4))) E_;                                            \
        const bool in_ = (pin >> (Q)) & 1u;                                                            \
        h4_ hv_, lv_;                                                                                  \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                             \
            float v_ = fmaxf(c_[r_] * a.scale1 + b1v[r_], 0.0f);                                       \
            v_ = in_ ? v_ : 0.0f;                                                                      \
            b1_max = fmaxf(b1_max, v_);                                                                \
            const E_ h_ = (E_)v_;                                                                      \
            hv_[r_] = h_;                                                                              \
            lv_[r_] = (E_)(v_ - (float)h_);                                                            \
        }                                                                                              \
        char *d_ = reinterpret_cast<char *>(As + (BUF) * ABUF);                                        \
        *reinterpret_cast<uint2 *>(d_ + pdst[Q]) = __builtin_bit_cast(uint2, hv_);                     \
        *reinterpret_cast<uint2 *>(d_ + (pdst[Q] ^ 16)) = __builtin_bit_cast(uint2, lv_);              \
    }
    if constexpr (B1) {
        // image patch: rows oy0 - 2 .. oy0 + TH + 1, columns ox0 - 2 .. ox0 + 33, zero outside the image
        const float *ximg = a.img + (size_t)img * a.H * a.W * 3;
        for (int e = tid; e < PATCH_F; e += NT) {
            const int pr = e / (PW * 3), rem = e - pr * (PW * 3);
            const int iy = oy0 - 2 + pr, ix = ox0 - 2 + rem / 3;
            patchf[e] = (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? ximg[((size_t)iy * a.W + ix) * 3 + rem % 3] : 0.0f;
        }
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // fragment addresses: A depends on the lane and the tap column s only (column swizzle), B on the lane only
    int a_off[3];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) a_off[s3] = (lm + s3) * 4 + ((2 * kh) ^ (((lm + s3) >> 2) & 3));
    int b_off[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = wn * (NI * 32) + j * 32 + lm;
        b_off[j] = n * 4 + ((2 * kh) ^ ((n >> 2) & 3));
    }

    // ---- prologue: halo of slice 0, weights of step 0 -------------------------------------------
    if constexpr (B1) {
        RPN_LOAD_B(0);
        RPN_B1_LOADW(0);
        RPN_STORE_B(0);
        __syncthreads();                                   // the patch is complete
        // k = 8 kg + j = (r * 3 + s) * 3 + c  ->  offset inside the patch relative to the pixel's top-left input
        int koff[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * kg + j;
            koff[j] = k < 27 ? (k / 9) * (PW * 3) + (k % 9) : -1;
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            const int p = (wave + NW * q) * 16 + lr;
            const bool valid = p < HP;
            const int pp = valid ? p : HP - 1;
            const int hy = pp / HW, hx = pp - hy * HW;
            const int base = (hy * PW + hx) * 3;
            float xs[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xs[j] = koff[j] >= 0 ? patchf[base + koff[j]] : 0.0f;
            phi[q] = __builtin_bit_cast(u32x4, split_piece<F16>(xs, false));
            plo[q] = __builtin_bit_cast(u32x4, split_piece<F16>(xs, true));
            const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) pin |= 1u << q;
            // channels 4 kg .. 4 kg + 3 of the slice: hi piece 2 * (kg >> 1), its half (kg & 1); lo piece = hi piece ^ 1
            pdst[q] = valid ? (pp * 4 + ((2 * (kg >> 1)) ^ ((hx >> 2) & 3))) * 16 + (kg & 1) * 8 : HP * 4 * 16 + (kg & 1) * 8;
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) RPN_B1_BLOCK(q, 0);
    } else {
#pragma unroll
        for (int R = 0; R < A_SLOTS; ++R) As[a_loff[R]] = RPN_LOAD_A(0, R);
        RPN_LOAD_B(0);
        RPN_STORE_B(0);
    }
    __syncthreads();

    int abuf = 0, bbuf = 0;
#pragma unroll 1
    for (int chunk = 0; chunk < chunks; ++chunk) {
        // the prefetch past the last slice / step is clamped to the last one: its data lands in the idle
        // buffers and is never read, which keeps the loop free of branches
        const int next_chunk = chunk + 1 < chunks ? chunk + 1 : chunk;
#pragma unroll
        for (int row = 0; row < 3; ++row) {              // unrolled: halo slot indices are compile-time
            const int step = chunk * 3 + row;
            RPN_LOAD_B(step + 1 < steps ? step + 1 : step);
            if constexpr (B1) {
                if (row == 0) RPN_B1_LOADW(next_chunk);      // first-layer weights of the slice computed below
            } else {
#pragma unroll
                for (int q = 0; q < A_RPS; ++q) a_reg[q] = RPN_LOAD_A(next_chunk, row * A_RPS + q);
            }
            if constexpr (NW == 4) RPN_PIN_LOADS();          // (the 8-wave variant has 128 VGPRs: pinned, it spills)

            const u32x4 *arow = As + (abuf * ABUF + (wm * MI + row) * HW * 4);
#pragma unroll
            for (int s = 0; s < 3; ++s) {                    // the 3 taps of filter row `row`
                u32x4 ahi[MI], alo[MI], bhi[NI], blo[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {               // a_off[s] is lane-constant; i * HW * 4 is an immediate
                    ahi[i] = arow[i * HW * 4 + a_off[s]];
                    alo[i] = arow[i * HW * 4 + (a_off[s] ^ 1)];
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    bhi[j] = Bs[(bbuf * 3 + s) * BN * 4 + b_off[j]];
                    blo[j] = Bs[(bbuf * 3 + s) * BN * 4 + (b_off[j] ^ 1)];
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        acc[i][j] = Half<F16>::mfma(alo[i], bhi[j], acc[i][j]);
                        acc[i][j] = Half<F16>::mfma(ahi[i], blo[j], acc[i][j]);
                        acc[i][j] = Half<F16>::mfma(ahi[i], bhi[j], acc[i][j]);
                    }
            }

            if (BBUF == 1) __syncthreads();              // every wave is done reading the single weight buffer
            RPN_STORE_B(BBUF == 2 ? (bbuf ^ 1) : 0);
            if constexpr (B1) {
                // the next slice's halo tile: QPS blocks per interval (in the last slice: that slice's own tile once more,
                // into the idle buffer -- cheaper than a branch that would fence this work off from the MFMAs)
#pragma unroll
                for (int q = row * QPS; q < (row + 1) * QPS && q < QB; ++q) RPN_B1_BLOCK(q, abuf ^ 1);
            } else {
#pragma unroll
                for (int q = 0; q < A_RPS; ++q) As[(abuf ^ 1) * ABUF + a_loff[row * A_RPS + q]] = a_reg[q];
            }
            __syncthreads();
            if (BBUF == 2) bbuf ^= 1;
        }
        abuf ^= 1;
    }

#undef RPN_LOAD_B
#undef RPN_STORE_B
#undef RPN_LOAD_A
#undef RPN_B1_LOADW
#undef RPN_B1_BLOCK
    if constexpr (B1 && F16) {
        if (a.status && !(b1_max <= 65504.0f)) atomicOr(a.status, 1u /* RPN_STATUS_F16_RANGE */);
    }
    // ---- epilogue: scale + bias + activation (+ fused 2x2 max-pool), transpose through LDS, 16-byte stores --
    float *stage = reinterpret_cast<float *>(lds) + wave * (32 * STAGE_LD);
    float bias_v[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * (NI * 32) + j * 32 + lm;
        bias_v[j] = (a.bias && n < a.Cout) ? a.bias[n] : 0.0f;
    }
    // activation as a branch-free clamp: linear / relu / relu6 (the launcher rejects sigmoid)
    const float act_lo = a.act == ACT_LINEAR ? -INFINITY : 0.0f;
    const float act_hi = a.act == ACT_RELU6 ? 6.0f : INFINITY;
    const int cout_chunks = a.Cout >> 4;
    const int nbase = n0 + wn * (NI * 32);                             // first channel of this wave's 64
    // write NPX staged pixels x 64 channels of output row `oy` (output image OHo x OWo, first column oxb): one buffer per row
    // segment [oxb, OWo), a lane owns eight channels of a pixel and writes their hi and lo pieces (split16_epilogue has the why)
    const int pix_bytes = a.out_f32 ? a.Cout * 4 : cout_chunks * 64;
    auto store_stage = [&](int npx_log2, int oy, int oxb, int OHo, int OWo) {
        if (oy >= OHo) return;
        const long long row_off = (((long long)img * OHo + oy) * OWo + oxb) * (long long)pix_bytes;
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(a.out) + row_off, (short)0,
                                                                              (OWo - oxb) * pix_bytes, 0x00020000);
        if (a.out_f32) {
            const int rounds = (16 << npx_log2) >> 6;                  // (npx * 16 float4s) / 64 lanes
            const int px_l = lane >> 4, q_l = lane & 15;
            const int n = nbase + 4 * q_l;
            unsigned voff = n < a.Cout ? (unsigned)(px_l * pix_bytes + n * 4) : 0x80000000u;
#pragma unroll
            for (int rd = 0; rd < rounds; ++rd) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(&stage[(rd * 4 + px_l) * STAGE_LD + 4 * q_l]);
                __builtin_amdgcn_raw_buffer_store_b128(v, ors, voff, 0, 0);
                voff += (unsigned)(4 * pix_bytes);
            }
        } else {
            const int rounds2 = (8 << npx_log2) >> 6;                  // (npx * 8 lanes) / 64
            const int px2 = lane >> 3, q2 = lane & 7;
            const int cl = q2 >> 1, hf = q2 & 1;
            const int n = nbase + cl * 16;
            unsigned voff = n < a.Cout ? (unsigned)(px2 * pix_bytes + (n >> 4) * 64 + hf * 32) : 0x80000000u;
#pragma unroll
            for (int rd = 0; rd < rounds2; ++rd) {
                float xs[8];
                const float *src = &stage[(rd * 8 + px2) * STAGE_LD + cl * 16 + hf * 8];
                const float4 v0 = *reinterpret_cast<const float4 *>(src);
                const float4 v1 = *reinterpret_cast<const float4 *>(src + 4);
                xs[0] = v0.x; xs[1] = v0.y; xs[2] = v0.z; xs[3] = v0.w;
                xs[4] = v1.x; xs[5] = v1.y; xs[6] = v1.z; xs[7] = v1.w;
                const u32x4 hi = __builtin_bit_cast(u32x4, split_piece<F16>(xs, false, a.status));
                const u32x4 lo = __builtin_bit_cast(u32x4, split_piece<F16>(xs, true, nullptr));
                __builtin_amdgcn_raw_buffer_store_b128(hi, ors, voff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(lo, ors, voff + 16u, 0, 0);
                voff += (unsigned)(8 * pix_bytes);
            }
        }
    };

    if constexpr (POOL) {
        // MaxPooling2D(2,2) 'valid' fused: the 2x2 window of output pixel (y,x) is rows (2y, 2y+1) -- two
        // M-blocks of this wave -- and accumulator registers (2e', 2e'+1) of one lane, so the max is lane-local.
        // max commutes with the monotone bias + activation, which are applied once afterwards.
        static_assert(!POOL || (MI % 2 == 0), "pooling needs an even number of rows per wave");
        const int OHo = a.H >> 1, OWo = a.W >> 1;
#pragma unroll
        for (int ip = 0; ip < MI / 2; ++ip) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int e2 = 0; e2 < 8; ++e2) {
                    const float v0 = fmaxf(acc[2 * ip][j][2 * e2], acc[2 * ip][j][2 * e2 + 1]);
                    const float v1 = fmaxf(acc[2 * ip + 1][j][2 * e2], acc[2 * ip + 1][j][2 * e2 + 1]);
                    const int m2 = (e2 & 1) + 4 * (e2 >> 1) + 2 * kh;             // pooled column 0..15
                    stage[m2 * STAGE_LD + j * 32 + lm] = fminf(fmaxf(fmaxf(v0, v1) * a.out_scale + bias_v[j], act_lo), act_hi);
                }
            wave_sync();
            store_stage(4, (oy0 + wm * MI + 2 * ip) >> 1, ox0 >> 1, OHo, OWo);
            wave_sync();
        }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {             // fully unrolled: acc[] must be indexed statically
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = (e & 3) + 8 * (e >> 2) + 4 * kh;
                    stage[m * STAGE_LD + j * 32 + lm] = fminf(fmaxf(acc[i][j][e] * a.out_scale + bias_v[j], act_lo), act_hi);
                }
            wave_sync();
            store_stage(5, oy0 + wm * MI + i, ox0, a.H, a.W);
            wave_sync();
        }
    }
}

// ---- 16x16x32-MFMA variant -----------------------------------------------------------------------------
// Same algorithm on v_mfma_f32_16x16x32_{f16,bf16}: on MI355X the chip holds a higher clock on this MFMA shape
// than on 32x32x16 under load, so equal cycles per flop turn into more flops per second.  K = 32 per MFMA, so the
// K-slice is 32 channels: 8 x 32 px x 128 ch tile, 8 waves (512 threads, each 64 px x 64 ch = 4 x 4 MFMA tiles),
// one workgroup per CU (136 KB LDS: two halo buffers of 340 px x 128 B, one weight buffer of 3 taps x 128 x 128 B).
// LDS rows hold 8 pieces [hi k0-7, hi k8-15, hi k16-23, hi k24-31, lo ...] XOR-swizzled by halo_swz(column)
// (resp. (n >> 1) & 7 for the weight rows, whose reads are never shifted): every ds_read_b128 fragment read is conflict-free.  Weights use the "split32" packing
// [Cin/32][9][cout_pad][128 B] in that piece order; activations stay SPLIT16 in HBM (pieces re-ordered on staging).

// Timing experiment only (-DRPN_NOSTORE): every epilogue store of a lane goes to one 16-byte slot per lane, so the
// instruction stream is unchanged but no output traffic reaches HBM.
#ifdef RPN_NOSTORE
#define RPN_STORE_INDEX(i) ((i) & 63)
#else
#define RPN_STORE_INDEX(i) (i)
#endif

// Epilogue shared by the 16x16x32 kernels: scale + bias + activation (+ fused 2x2 max-pool), transposed through a
// wave-private LDS staging area, 16-byte stores (C/D of the 16x16 MFMA: column = lane & 15, row = 4 * (lane >> 4) + reg).
// The caller guarantees that no wave still reads (and no DMA still writes) the pipeline buffers the area overlays.
template <bool F16, bool POOL, int RW, int NW, int NJ = 4>
__device__ __forceinline__ void split16_epilogue(f32x4 (&acc)[RW * 2][NJ], float *lds_f, const SplitConvArgs &a, int img,
                                                 int oy0, int ox0, int n0, int wave, int wm, int wn, int lane,
                                                 int stamp_base = -1 /* -DRPN_STAMP builds: first of 4 stamp slots */)
{
    (void)stamp_base;
    // the lane index as an opaque value: the epilogue's lane arithmetic (e / PP, e % PP, staging addresses) is otherwise
    // loop-invariant in the persistent kernel's tile loop, gets hoisted in front of it and is carried through the tap loop
    // in registers the 256-VGPR kernel does not have (5 were spilled to scratch)
    asm volatile("" : "+v"(lane));
    constexpr int MT = RW * 2;
    constexpr int CW = 16 * NJ;                        // channels per wave
    constexpr int STAGE_LD = CW + kStagePad;
    constexpr int PP = CW / 4;                         // 16-byte pieces (or float4s) per pixel
    const int lr = lane & 15, kg = lane >> 4;
    // ---- epilogue (as in the 32x32 kernel; C/D of the 16x16 MFMA: column = lane & 15, row = 4 * (lane >> 4) + reg) --
    float *stage = lds_f + wave * (32 * STAGE_LD);
    float bias_v[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wn * CW + j * 16 + lr;
        bias_v[j] = (a.bias && n < a.Cout) ? a.bias[n] : 0.0f;
    }
    const float act_lo = a.act == ACT_LINEAR ? -INFINITY : 0.0f;
    const float act_hi = a.act == ACT_RELU6 ? 6.0f : INFINITY;
    const int cout_chunks = a.Cout >> 4;
    const int nbase = n0 + wn * CW;
    // Stores (round 6): one buffer per output ROW SEGMENT [oxb, OWo) of the image -- a pixel past the right edge is past the buffer's
    // end and the hardware drops it, a lane whose channels lie beyond Cout carries an out-of-range offset -- so a round is two LDS
    // reads, the hi / lo split and ONE buffer store whose 32-bit offset advances by a constant: no branch, no 64-bit index
    // arithmetic, no integer multiply.  (Rounds 1-5 recomputed (((img * OH + oy) * OW + ox) * chunks + n / 16) * 4 + pc in 64 bits
    // and branched on ox < OW && n < Cout in every round: ~100 v_mul_lo_u32 / v_lshl_add_u64 / v_mad_u64_u32 and 92 exec branches per
    // tile and thread -- a third of the epilogue's 2 600 vector instructions.)
    const int pix_bytes = a.out_f32 ? a.Cout * 4 : cout_chunks * 64;       // bytes per output pixel
    constexpr int PXR = 64 / PP;                                          // pixels per round of 64 lanes
    const int px_l = lane / PP, q_l = lane % PP;
    auto store_stage = [&](int npx_log2, int oy, int oxb, int OHo, int OWo) {
        if (oy >= OHo) return;
        const int rounds = (PP << npx_log2) >> 6;
        const long long row_off = (((long long)img * OHo + oy) * OWo + oxb) * (long long)pix_bytes;
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(a.out) + row_off, (short)0,
                                                                              (OWo - oxb) * pix_bytes, 0x00020000);
        if (a.out_f32) {
            const int n = nbase + 4 * q_l;
            unsigned voff = n < a.Cout ? (unsigned)(px_l * pix_bytes + n * 4) : 0x80000000u;
#pragma unroll
            for (int rd = 0; rd < rounds; ++rd) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(&stage[(rd * PXR + px_l) * STAGE_LD + 4 * q_l]);
#ifdef RPN_NOSTORE
                __builtin_amdgcn_raw_buffer_store_b128(v, ors, (unsigned)(lane * 16), 0, 0);
#else
                __builtin_amdgcn_raw_buffer_store_b128(v, ors, voff, 0, 0);
#endif
                voff += (unsigned)(PXR * pix_bytes);
            }
        } else {
            // a lane owns EIGHT channels of a pixel and writes their hi piece and their lo piece -- 32 contiguous bytes of the pixel's
            // SPLIT16 record -- from one read of the eight floats: 8 conversions to 16 bits, 8 back, 8 subtractions, 8 conversions
            // per 32 bytes.  (Rounds 1-5: a lane per 16-byte piece, hi or lo by lane parity -- both computed, one selected (8
            // v_cndmask), the same eight floats read by two lanes: 40 vector instructions and two LDS reads per 16 bytes.)
            constexpr int LPP = CW / 8;                                   // lanes per pixel: 8 | 4
            constexpr int PXR2 = 64 / LPP;                                // pixels per round: 8 | 16
            const int px2 = lane / LPP, q2 = lane % LPP;
            const int cl = q2 >> 1, hf = q2 & 1;                          // 16-channel slice of the wave's channels, its half
            const int n = nbase + cl * 16;
            unsigned voff = n < a.Cout ? (unsigned)(px2 * pix_bytes + (n >> 4) * 64 + hf * 32) : 0x80000000u;
            const int rounds2 = (1 << npx_log2) / PXR2;
#pragma unroll
            for (int rd = 0; rd < rounds2; ++rd) {
                float xs[8];
                const float *src = &stage[(rd * PXR2 + px2) * STAGE_LD + cl * 16 + hf * 8];
                const float4 v0 = *reinterpret_cast<const float4 *>(src);
                const float4 v1 = *reinterpret_cast<const float4 *>(src + 4);
                xs[0] = v0.x; xs[1] = v0.y; xs[2] = v0.z; xs[3] = v0.w;
                xs[4] = v1.x; xs[5] = v1.y; xs[6] = v1.z; xs[7] = v1.w;
                const u32x4 hi = __builtin_bit_cast(u32x4, split_piece<F16>(xs, false, a.status));
                const u32x4 lo = __builtin_bit_cast(u32x4, split_piece<F16>(xs, true, nullptr));
#ifdef RPN_NOSTORE
                __builtin_amdgcn_raw_buffer_store_b128(hi, ors, (unsigned)(lane * 32), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(lo, ors, (unsigned)(lane * 32 + 16), 0, 0);
#else
                __builtin_amdgcn_raw_buffer_store_b128(hi, ors, voff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(lo, ors, voff + 16u, 0, 0);
#endif
                voff += (unsigned)(PXR2 * pix_bytes);
            }
        }
    };

    if constexpr (POOL) {
        const int OHo = a.H >> 1, OWo = a.W >> 1;     // RW == 2: rows (2wm, 2wm+1) are one pooling row pair
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    const float v0 = fmaxf(acc[hf][j][2 * r2], acc[hf][j][2 * r2 + 1]);            // row i = 0
                    const float v1 = fmaxf(acc[(MT - 2) + hf][j][2 * r2], acc[(MT - 2) + hf][j][2 * r2 + 1]);    // row i = 1
                    const int px2 = 8 * hf + 2 * kg + r2;                                          // pooled column 0..15
                    stage[px2 * STAGE_LD + j * 16 + lr] = fminf(fmaxf(fmaxf(v0, v1) * a.out_scale + bias_v[j], act_lo), act_hi);
                }
        wave_sync();
        store_stage(4, (oy0 + 2 * wm) >> 1, ox0 >> 1, OHo, OWo);
    } else {
#pragma unroll
        for (int i = 0; i < RW; ++i) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int px = 16 * hf + 4 * kg + r;
                        stage[px * STAGE_LD + j * 16 + lr] =
                            fminf(fmaxf(acc[i * 2 + hf][j][r] * a.out_scale + bias_v[j], act_lo), act_hi);
                    }
            wave_sync();
            if (stamp_base >= 0) RPN_STAMP_AT(stamp_base + 2 * i);
            store_stage(5, oy0 + RW * wm + i, ox0, a.H, a.W);
            wave_sync();
            if (stamp_base >= 0) RPN_STAMP_AT(stamp_base + 2 * i + 1);
        }
    }
}

// TH: tile rows (8 | 4); WN: waves along N (2 -> 128 channels, 1 -> 64); NW: waves (8 | 4).  Each wave owns
// RW = TH / (NW / WN) rows of 32 pixels x 64 channels.  <8,2,8>: one 136 KB workgroup per CU (large layers);
// <4,1,4>: 77 KB, two workgroups per CU (small feature maps).
template <int TH, int WN, int NW, bool F16, bool POOL>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 2 : 2)
conv3x3_split16_kernel(SplitConvArgs a, int tiles_x, int tiles_y, int n_tiles)
{
    constexpr int NT = 64 * NW, BN = 64 * WN;
    constexpr int WM = NW / WN, RW = TH / WM;          // rows of 32 pixels per wave
    constexpr int MT = RW * 2;                         // 16-pixel MFMA tiles per wave along M
    constexpr int HP = (TH + 2) * HW;                 // halo pixels
    constexpr int PPP = 8;                            // 16-byte pieces per 32-channel row
    constexpr int ABUF = HP * PPP + 8;                // + dummy slot for idle lanes
    constexpr int A_PIECES = HP * PPP;
    constexpr int A_ROUNDS = (A_PIECES + NT - 1) / NT;
    constexpr int A_RPS = (A_ROUNDS + 2) / 3, A_SLOTS = 3 * A_RPS;
    constexpr int B_PIECES = 3 * BN * PPP;            // one filter row of 32-channel weight rows
    constexpr int B_ROUNDS = B_PIECES / NT;
    constexpr int STAGE_LD = 64 + kStagePad;
    constexpr int LDS_PIPE = 2 * ABUF + B_PIECES;
    constexpr int LDS_STAGE = (NW * 32 * STAGE_LD * 4 + 15) / 16;
    constexpr int LDS_UINT4 = LDS_PIPE > LDS_STAGE ? LDS_PIPE : LDS_STAGE;
    static_assert(RW * WM == TH && RW >= 1 && B_PIECES % NT == 0 && (!POOL || RW == 2), "tile shape");

    __shared__ uint4 lds[LDS_UINT4];
    u32x4 *As = reinterpret_cast<u32x4 *>(lds);
    u32x4 *Bs = reinterpret_cast<u32x4 *>(lds) + 2 * ABUF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;         // rows [RW*wm, +RW), channels [64wn, +64)
    const int lr = lane & 15, kg = lane >> 4;         // fragment row / k-group (8 channels) of this lane

    int nt, mt;
    {
        const int m_tiles = tiles_x * tiles_y * a.B;
        const int XN = n_tiles >= 8 ? 8 : (n_tiles >= 4 ? 4 : (n_tiles >= 2 ? 2 : 1)), XM = 8 / XN;
        const int NTl = (n_tiles + XN - 1) / XN;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        nt = (slot % NTl) * XN + (xcd % XN);
        mt = (slot / NTl) * XM + (xcd / XN);
        if (nt >= n_tiles || mt >= m_tiles) return;
    }
    const int tx = mt % tiles_x;
    mt /= tiles_x;
    const int ty = mt % tiles_y;
    const int img = mt / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TWS, n0 = nt * BN;
    RPN_STAMP_AT(0);
    RPN_STAMP_VAL(1, ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                         (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4));       // XCC_ID, HW_ID

    const int all_chunks = a.Cin >> 5;                // 32-channel slices
    int c_begin = (int)((long long)blockIdx.y * all_chunks / gridDim.y);        // split-K: this workgroup's slices
    int chunks = (int)((long long)(blockIdx.y + 1) * all_chunks / gridDim.y);   // (one past its last slice)
    int c_fold = -1;                                  // K tree, two leaves per workgroup: the first slice of the second leaf
    constexpr bool KT = TH == 4 && NW == 4 && !POOL;   // (the instantiation the K-tree launcher uses; the others have no registers for it)
    if (KT && a.ktree) {                              // (gridDim.y = 2 or 4: the launcher sends an unsplit tree layer elsewhere)
        const int per = 4 / (int)gridDim.y, l0 = (int)blockIdx.y * per;
        c_begin = ktree_cut(all_chunks, l0);
        chunks = ktree_cut(all_chunks, l0 + per);
        if (per == 2) c_fold = ktree_cut(all_chunks, l0 + 1);
    }
    const int steps = chunks * 3;
    const size_t in_pix_stride = (size_t)(a.Cin >> 4) * 4;
    const uint4 *__restrict__ xin = a.x + (size_t)img * a.H * a.W * in_pix_stride;

    constexpr unsigned kOob = 0x80000000u;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(xin), (short)0, (int)((size_t)a.H * a.W * in_pix_stride * 16), 0x00020000);
    unsigned a_goff[A_SLOTS];
    int a_loff[A_SLOTS];
#pragma unroll
    for (int R = 0; R < A_SLOTS; ++R) {
        const int e = R * NT + tid;
        const int pix = e >> 3, q = e & 7;            // q: piece of the pixel's 128-byte (2 x SPLIT16 record) slice
        const int hy = pix / HW, hx = pix - hy * HW;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        const bool piece = e < A_PIECES;
        const bool inimg = piece && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        const int kgq = (q >> 2) * 2 + ((q >> 1) & 1), lo = q & 1;     // SPLIT16 record order -> (k-group, lo)
        a_loff[R] = piece ? (pix * PPP + ((lo * 4 + kgq) ^ halo_swz(hx))) : HP * PPP;
        a_goff[R] = inimg ? (unsigned)((((size_t)iy * a.W + ix) * in_pix_stride + q) * 16) : kOob;
    }
    u32x4 b_reg[B_ROUNDS];
    u32x4 a_reg[A_RPS];
#define RPN16_LOAD_B(STEP)                                                                                   \
    {                                                                                                        \
        const u32x4 *src_ = reinterpret_cast<const u32x4 *>(a.w) + ((size_t)(STEP) * 3 * a.cout_pad + n0) * PPP; \
        _Pragma("unroll") for (int i_ = 0; i_ < B_ROUNDS; ++i_) {                                            \
            const int e_ = tid + i_ * NT;                                                                    \
            const int t_ = e_ / (BN * PPP), rem_ = e_ - t_ * (BN * PPP);                                     \
            b_reg[i_] = src_[(size_t)t_ * a.cout_pad * PPP + rem_];                                          \
        }                                                                                                    \
    }
#define RPN16_STORE_B()                                                                                      \
    {                                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < B_ROUNDS; ++i_) {                                            \
            const int e_ = tid + i_ * NT;                                                                    \
            const int t_ = e_ / (BN * PPP), rem_ = e_ - t_ * (BN * PPP);                                     \
            Bs[t_ * BN * PPP + rem_] = b_reg[i_];       /* the packed weights already are the swizzled image */ \
        }                                                                                                    \
    }
#define RPN16_LOAD_A(CHUNK, R) \
    __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, a_goff[R], (CHUNK) * 128, 0))

    f32x4 acc[MT][4];                                 // [M-tile = row i * 2 + half][N-tile j]
    f32x4 leaf0[KT ? MT : 1][KT ? 4 : 1];            // (K tree: the finished first leaf of this workgroup's pair)
    (void)leaf0;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (KT) leaf0[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    int a_off[3][2];                                  // [tap column s][16-px half]: lane-constant
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int hx = 16 * hf + lr + s3;
            a_off[s3][hf] = hx * PPP + (kg ^ halo_swz(hx));
        }
    int b_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = wn * 64 + j * 16 + lr;
        b_off[j] = n * PPP + (kg ^ ((n >> 1) & 7));
    }

#pragma unroll
    for (int R = 0; R < A_SLOTS; ++R) As[a_loff[R]] = RPN16_LOAD_A(c_begin, R);
    RPN16_LOAD_B(3 * c_begin);
    RPN16_STORE_B();
    __syncthreads();
    RPN_STAMP_AT(2);

    int abuf = 0;
#pragma unroll 1
    for (int chunk = c_begin; chunk < chunks; ++chunk) {
        const int next_chunk = chunk + 1 < chunks ? chunk + 1 : chunk;
#pragma unroll
        for (int row = 0; row < 3; ++row) {
            const int step = chunk * 3 + row;
            RPN16_LOAD_B(step + 1 < steps ? step + 1 : step);
#pragma unroll
            for (int q = 0; q < A_RPS; ++q) a_reg[q] = RPN16_LOAD_A(next_chunk, row * A_RPS + q);
            RPN_PIN_LOADS();

            const u32x4 *arow = As + (abuf * ABUF + (RW * wm + row) * HW * PPP);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                u32x4 ahi[MT], alo[MT], bhi[4], blo[4];
#pragma unroll
                for (int m = 0; m < MT; ++m) {            // m = i * 2 + half
                    const int idx = (m >> 1) * HW * PPP + a_off[s][m & 1];
                    ahi[m] = arow[idx];
                    alo[m] = arow[idx ^ 4];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bhi[j] = Bs[s * BN * PPP + b_off[j]];
                    blo[j] = Bs[s * BN * PPP + (b_off[j] ^ 4)];
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[m][j] = mfma16<F16>(alo[m], bhi[j], acc[m][j]);
                        acc[m][j] = mfma16<F16>(ahi[m], blo[j], acc[m][j]);
                        acc[m][j] = mfma16<F16>(ahi[m], bhi[j], acc[m][j]);
                    }
            }
            __syncthreads();                              // every wave is done reading the weight buffer
            RPN16_STORE_B();
#pragma unroll
            for (int q = 0; q < A_RPS; ++q) As[(abuf ^ 1) * ABUF + a_loff[row * A_RPS + q]] = a_reg[q];
            __syncthreads();
        }
        abuf ^= 1;
        RPN_STAMP_AT(4 + chunk);
        if constexpr (KT) {
            if (chunk + 1 == c_fold) {                // K tree: the first leaf is complete
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        leaf0[m][j] = acc[m][j];
                        acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
            }
        }
    }
    if constexpr (KT) {
        if (c_fold >= 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[m][j] = leaf0[m][j] + acc[m][j];
        }
    }
#undef RPN16_LOAD_B
#undef RPN16_STORE_B
#undef RPN16_LOAD_A

    if (gridDim.y > 1) {                              // split-K: raw partial sums to this workgroup's slab
        SplitConvArgs ar = a;
        ar.bias = nullptr;
        ar.act = ACT_LINEAR;
        ar.out_f32 = 1;
        ar.out = reinterpret_cast<float *>(a.out) + (size_t)blockIdx.y * a.slab_floats;
        split16_epilogue<F16, POOL, RW, NW>(acc, reinterpret_cast<float *>(lds), ar, img, oy0, ox0, n0, wave, wm, wn, lane);
    } else {
        split16_epilogue<F16, POOL, RW, NW>(acc, reinterpret_cast<float *>(lds), a, img, oy0, ox0, n0, wave, wm, wn, lane);
    }
    RPN_STAMP_AT(3);
}

// ---- 16x16x32-MFMA variant: persistent workgroups, LDS-DMA pipeline ----------------------------------------------
// Same tile (8 x 32 px x 128 ch, 8 waves), same LDS images and the same products accumulated in the same order as
// conv3x3_split16_kernel<8, 2, 8>, with the staging pipeline and the tile loop rebuilt from in-kernel cycle stamps
// (scripts/stamp_probe.py) of that kernel: 18.7k cycles per 32-channel slice against 13.8k of MFMA issue, and 13k
// cycles per tile of prologue + epilogue + dispatch gap that nothing overlapped (one workgroup per CU).
//   * `buffer_load ... lds` (global -> LDS, no registers, no ds_write): one barrier per TAP and no exposed store phase
//     (before: 62 KB of ds_write_b128 at ~79 B/clk/CU between the two barriers of every filter row).  The weights of
//     tap t+3 are DMA'd into a 3-slot ring while tap t computes; the halo tile of the next slice in 6 pieces per wave
//     during taps 0..5 of the current slice (other halo buffer).
//   * the MFMA fragments of tap t+1 are read into a second register set during tap t (one read behind each of the 16
//     chain-head MFMAs), so the matrix pipe does not wait for LDS behind a barrier.
//   * persistent workgroups (one per CU) walk the tiles of their XCD: the tap stream simply continues into the next
//     tile (its halo tile and first weights are DMA'd during the last slice of the current one), so there is no
//     prologue and no dispatch gap between tiles.
//   * the epilogue stages through halo buffer 1 + the LDS left over (dead until the next tile's first interval refills
//     them), one workgroup barrier per tile.  (Tried and dropped: swapped MFMA operands -- weights as A, pixels as B, so
//     that a lane's 4 accumulator registers are 4 consecutive CHANNELS of one pixel -- with 8-byte stores straight from
//     registers: 9.8k against 6.5k cycles per tile, the store issue became the limit.)
// Hazards, by the rule "read a DMA'd buffer one interval after the wait that retires it":
//   RAW  weights(t+2) (issued in interval t-1) are retired by the counted vmcnt at the end of interval t (which leaves
//        only interval t's own DMAs in flight), then the barrier; read in interval t+1.  Halo pieces likewise (issued in
//        intervals 0..5, first read in interval 8).
//   WAR  the ring slot of tap t+3 is the slot of tap t, whose fragments were read in interval t-1 and retired by the
//        lgkmcnt(0) in front of that interval's barrier.  The other halo buffer was last read in interval 7 of the
//        previous slice.
// (Tried: a 6-slot ring at 64 channels per tile -- weights four intervals ahead -- and the halo pieces two per tap behind
// the weights of taps 0..2 so that the in-order counter gives them two intervals to land: no gain on any layer, the DMA
// latency is already covered.)
// The weights in HBM are already the swizzled LDS image ("split32" packing), so their DMA is a linear copy; the halo
// tile's swizzle is applied on the per-lane SOURCE address (the LDS side of a DMA is lane-linear).  Out-of-image halo
// pixels are out-of-range buffer offsets: the DMA writes zeros for them (checked: scripts/micro/dma_oob.hip).
// Needs an even number of 32-channel slices (the body is unrolled over two slices = 18 taps so that register sets,
// halo buffers and ring slots are all compile-time) and an input tensor below 2 GiB (32-bit buffer offsets).
#ifndef RPN_DMA_SCHED
#define RPN_DMA_SCHED 1
#endif
#define RPN_LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

// Tile schedule of a persistent workgroup: blockIdx & 7 labels the XCD; an XCD owns n-tiles (xcd % XN) + XN * k and
// walks its slots.  next(): first real tile at or after `slot` (slots past the edge of the XN x XM ownership grid are
// skipped); -1: none.  Tiles are 8 rows x 32 pixels x bn channels.
struct TileWalk {
    int n_tiles, m_tiles, tiles_x, tiles_y, XN, XM, NTl, n_slots, xcd, slot_stride, bn;
    // tile of slot `slot` of this XCD label; false: the slot lies past the edge of the XN x XM ownership grid
    __device__ __forceinline__ bool decode(int slot, int &img, int &oy0, int &ox0, int &n0) const
    {
        const int nt = (slot % NTl) * XN + (xcd % XN);
        int mt = (slot / NTl) * XM + (xcd / XN);
        if (nt >= n_tiles || mt >= m_tiles) return false;
        const int tx = mt % tiles_x;
        mt /= tiles_x;
        const int ty = mt % tiles_y;
        img = mt / tiles_y;
        oy0 = ty * 8; ox0 = tx * TWS; n0 = nt * bn;
        return true;
    }
    // static schedule: first real tile at or after `slot`, stepping by the number of workgroups of this label; -1: none
    __device__ __forceinline__ int next(int slot, int &img, int &oy0, int &ox0, int &n0) const
    {
        for (; slot < n_slots; slot += slot_stride)
            if (decode(slot, img, oy0, ox0, n0)) return slot;
        return -1;
    }
    // dynamic schedule (one thread): given a ticket drawn from the label's queue, draw on until it names a real tile;
    // >= n_slots: the queue is exhausted
    // (the queue hands out the slots after the first `slot_stride`: those are every workgroup's statically assigned
    // first tile, so that a launch with one tile per workgroup cannot leave workgroups empty-handed)
    __device__ __forceinline__ int settle(int ticket, unsigned *queue) const
    {
        int img, oy0, ox0, n0, slot = ticket + slot_stride;
        while (slot < n_slots && !decode(slot, img, oy0, ox0, n0)) slot = (int)atomicAdd(queue, 1u) + slot_stride;
        return slot;
    }
};

// Byte offset (inside the SPLIT16 input tensor) of the 16-byte piece that lane `lane` of halo wave-instruction
// k = j * NW + wave DMAs: LDS piece e = 64 k + lane = halo pixel * 8 + physical slot; the LDS image's XOR swizzle is applied
// here, on the source side.  Out-of-image pixels and the padding lanes get an out-of-range offset (the DMA writes zeros).
template <int NW, int A_INSTR, int A_PIECES>
__device__ __forceinline__ unsigned halo_source_offset(int j, int wave, int lane, int im, int y0, int x0, int H, int W,
                                                       int in_pix_stride)
{
    int ln = lane;
    asm volatile("" : "+v"(ln));                      // opaque: keeps the lane-constant parts from being hoisted
    int k = j * NW + wave;                            // out of the tap loop into long-lived registers
    if (k >= A_INSTR) k -= A_INSTR;                   // surplus instructions of the last round re-fetch pieces 0, 1, ...
    const int e = k * 64 + ln;
    const int pix = e >> 3, ps = e & 7;
    const int hy = pix / HW, hx = pix - hy * HW;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool in = e < A_PIECES && iy >= 0 && iy < H && ix >= 0 && ix < W;
    const int lg = ps ^ halo_swz(hx);              // logical piece: lo * 4 + k-group
    const int q = (((lg & 3) >> 1) << 2) | ((lg & 1) << 1) | (lg >> 2);   // its place in the slice's SPLIT16 records
    return in ? (unsigned)((((im * H + iy) * W + ix) * in_pix_stride + q) * 16) : 0x80000000u;
}

template <bool F16, bool POOL, int BN, bool KTREE = false>
__global__ void __launch_bounds__(512, 2)
conv3x3_split16_dma_kernel(SplitConvArgs a, int tiles_x, int tiles_y, int n_tiles)
{
    static_assert(!KTREE || (BN == 64 && !POOL), "K tree: the 64-wide tile has the registers for two more accumulator sets");
    constexpr int TH = 8, NW = 8, WN = 2, RW = 2, MT = 4, PPP = 8;
    constexpr int NJ = BN / (16 * WN);                 // 16-channel MFMA tiles per wave: 4 (BN = 128) or 2 (BN = 64)
    constexpr int B_PER_WAVE = BN * PPP / 64 / NW;     // weight DMA instructions per wave and tap: 2 or 1
    constexpr int HP = (TH + 2) * HW;                  // 340 halo pixels
    constexpr int A_INSTR = (HP * PPP + 63) / 64;      // 43 wave-instructions (1 KB each) per halo tile
    constexpr int ABUF = A_INSTR * 64;                 // pieces per halo buffer (the tail of the last KB is padding)
    constexpr int A_PER_WAVE = (A_INSTR + NW - 1) / NW;   // 6 (the 5 surplus ones of the last round duplicate pieces 0..4)
    constexpr int BSLOT = BN * PPP;                    // pieces per ring slot (one tap: 16 KB)
    // LDS map (16-byte pieces): halo buffer 0 | weight ring | 1 KB scratch (tile queue) | halo buffer 1 | spare.  At the end of a tile halo
    // buffer 1 is dead (an even number of slices), so buffer 1 + spare is the epilogue's staging area: 8 waves x 32 px x
    // 68 floats = 69632 bytes = exactly what is left of the CU's 160 KB.
    constexpr int B_AT = ABUF, DUMP = B_AT + 3 * BSLOT, A1_AT = DUMP + 64;
    constexpr int STAGE_PIECES = NW * 32 * (16 * NJ + kStagePad) * 4 / 16;
    constexpr int LDS_UINT4 = A1_AT + (ABUF > STAGE_PIECES ? ABUF : STAGE_PIECES);
    static_assert(A_PER_WAVE <= 9 && RW * (NW / WN) == TH && MT == 2 * RW && LDS_UINT4 * 16 <= 160 * 1024 &&
                      (BN == 128 || BN == 64), "tile shape");

    __shared__ uint4 lds[LDS_UINT4];
    u32x4 *As = reinterpret_cast<u32x4 *>(lds);       // halo buffer b at As + b * A1_AT
    u32x4 *Bs = reinterpret_cast<u32x4 *>(lds) + B_AT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, kg = lane >> 4;

    // tile schedule: blockIdx & 7 labels the XCD; an XCD owns n-tiles (xcd % XN) + XN * k and walks its slots
    const int m_tiles = tiles_x * tiles_y * a.B;
    const int XN = n_tiles >= 8 ? 8 : (n_tiles >= 4 ? 4 : (n_tiles >= 2 ? 2 : 1)), XM = 8 / XN;
    const int NTl = (n_tiles + XN - 1) / XN, MTl = (m_tiles + XM - 1) / XM;
    const int xcd = blockIdx.x & 7;
    const int slot_stride = gridDim.x >> 3;
    const TileWalk walk{n_tiles, m_tiles, tiles_x, tiles_y, XN, XM, NTl, NTl * MTl, xcd, slot_stride, BN};
#define next_tile(SLOT, IMG, OY0, OX0, N0) walk.next((SLOT), (IMG), (OY0), (OX0), (N0))

    const int all_chunks = a.Cin >> 5;                // 32-channel slices (even)
    // K tree split two ways (gridDim.y == 2; launch_conv3x3_split16_ksplit on grids of 65 .. 128 tiles): workgroup y walks the
    // slices [cb, cb + chunks) = leaves 2y, 2y + 1 (ktree_cut's middle cut is even: whole slice pairs) and writes the RAW sum of
    // its two leaves to slab y (the host passes no bias, a linear activation, float32 output); the RPN head adds the slabs
    int cb = 0, chunks = all_chunks;
    if constexpr (KTREE) {
        if (gridDim.y == 2) {
            const int mid = ktree_cut(all_chunks, 2);
            cb = blockIdx.y ? mid : 0;
            chunks = blockIdx.y ? all_chunks - mid : mid;
            a.out = reinterpret_cast<float *>(a.out) + (size_t)blockIdx.y * a.slab_floats;
        }
    }
    const int total_taps = chunks * 9, tap0 = cb * 9;
    const int in_pix_stride = (a.Cin >> 4) * 4;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(a.x), (short)0, (int)((size_t)a.B * a.H * a.W * in_pix_stride * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(a.w), (short)0, (int)((size_t)all_chunks * 9 * a.cout_pad * 128), 0x00020000);
    const int w_tap_bytes = a.cout_pad * 128;

    // halo DMA: wave-instruction k = j * 8 + wave fills LDS pieces [64k, 64k + 64); piece e = pixel * 8 + physical slot.
    // The source offset is recomputed at each use (top of an interval, where register pressure is lowest; ~25 VALU
    // operations against the interval's 48 MFMAs) instead of living in 6 registers through the whole kernel.
#define halo_goff(J, IM, Y0, X0) halo_source_offset<NW, A_INSTR, HP * PPP>((J), wave, lane, (IM), (Y0), (X0), a.H, a.W, in_pix_stride)
    const unsigned b_voff = (unsigned)(((B_PER_WAVE * wave) * 64 + lane) * 16);   // this wave's share of a tap's weights

#define RPN_DMA_A(J, GOFF, SOFF, BUF)                                                                             \
    {                                                                                                             \
        const int k0_ = (J) * NW + wave, k_ = k0_ < A_INSTR ? k0_ : k0_ - A_INSTR;   /* surplus: a duplicate */         \
        u32x4 *dst_ = As + (BUF) * A1_AT + k_ * 64;                                                               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, RPN_LDS_PTR(dst_), 16, (GOFF), (SOFF), 0, 0);             \
    }
#define RPN_DMA_B(SOFF, SLOT)                                                                                     \
    {                                                                                                             \
        u32x4 *dst_ = Bs + (SLOT) * BSLOT + (B_PER_WAVE * wave) * 64;                                             \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, RPN_LDS_PTR(dst_), 16, b_voff, (SOFF), 0, 0);             \
        /* the instruction offset advances BOTH the global and the LDS address */                                \
        if constexpr (B_PER_WAVE == 2)                                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, RPN_LDS_PTR(dst_), 16, b_voff, (SOFF), 1024, 0);      \
    }

    int a_off[3][2];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int hx = 16 * hf + lr + s3;
            a_off[s3][hf] = hx * PPP + (kg ^ halo_swz(hx));
        }
    int b_off[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = wn * (16 * NJ) + j * 16 + lr;
        b_off[j] = n * PPP + (kg ^ ((n >> 1) & 7));
    }

    // Tile schedule.  Static: workgroup w of a label takes slots w, w + stride, ...  Dynamic (a.sched != null): only the
    // first tile is w; every further tile is drawn from a queue shared by the label's workgroups -- the ticket is drawn
    // (one atomic, not waited for) when a tile starts and turned into the next tile at taps 4..6, in time for the last
    // slice's look-ahead DMAs -- so that a CU that starts late or shares its time (another stream's kernel holding LDS:
    // the NMS, an RCCL kernel) takes fewer tiles instead of stalling a fixed share of the layer.
    const bool dyn = a.sched != nullptr;
    int *tile_q = reinterpret_cast<int *>(lds + DUMP);           // [tile parity]: the slot drawn for the next tile
    int img = 0, oy0 = 0, ox0 = 0, n0 = 0;
    int cur;
    if (dyn) {
        cur = blockIdx.x >> 3;
        if (!walk.decode(cur, img, oy0, ox0, n0)) {              // first slot on the ownership grid's padding (rare)
            if (tid == 0) tile_q[0] = walk.settle((int)atomicAdd(a.sched + xcd, 1u), a.sched + xcd);
            __syncthreads();
            cur = __builtin_amdgcn_readfirstlane(tile_q[0]);
            __syncthreads();
            if (cur >= walk.n_slots || !walk.decode(cur, img, oy0, ox0, n0)) cur = -1;
        }
    } else {
        cur = next_tile(blockIdx.x >> 3, img, oy0, ox0, n0);
    }
    // Every workgroup counts itself out, the last one re-arms the counters for the next launch on this stream.  Two
    // levels (per label, then one count of finished labels): 256 workgroups finishing together on ONE counter cost
    // ~6 us per launch in same-address atomics.
#define RPN_SCHED_EXIT()                                                                                          \
    if (dyn && tid == 0) {                                                                                        \
        __threadfence();                                                                                          \
        if (atomicAdd(a.sched + 8 + xcd, 1u) == (gridDim.x >> 3) - 1 && atomicAdd(a.sched + 16, 1u) == 7) {       \
            _Pragma("unroll") for (int i_ = 0; i_ < 17; ++i_) a.sched[i_] = 0u;                                   \
        }                                                                                                         \
    }
    if (cur < 0) {
        RPN_SCHED_EXIT();
        return;
    }
    RPN_STAMP_AT(0);
    RPN_STAMP_VAL(1, ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                         (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4));       // XCC_ID, HW_ID

    // Fragment registers.  The interval of a tap is worked in two halves (the wave's first / second pixel row) so that
    // only the weight fragments are double-buffered: fx0 / fx1 hold [hi, lo] of the two 16-pixel tiles of row 0 / 1.
    //   half 0: MFMAs of row 0 (fx0, fw[cur]); reads: fx1 of THIS tap
    //   half 1: MFMAs of row 1 (fx1, fw[cur]); reads: fw[next] and fx0 of the NEXT tap
    u32x4 fx0[4], fx1[4], fw[2][2 * NJ];                   // fx: [16-px half * 2 + (0 hi | 1 lo)];  fw: [set][j * 2 + (0 hi | 1 lo)]
#define RPN_X_ADDR(BUF, ROW, S, I, II)   /* row I of the wave, element II = half * 2 + lohi */                       \
    (As[(BUF) * A1_AT + (RW * wm + (ROW) + (I)) * HW * PPP + (a_off[S][(II) >> 1] ^ (((II) & 1) ? 4 : 0))])
#define RPN_W_ADDR(S, II)                /* element II = j * 2 + lohi */                                           \
    (Bs[(S) * BSLOT + (b_off[(II) >> 1] ^ (((II) & 1) ? 4 : 0))])

    // ---- prologue (first tile only): halo tile of slice 0, weights of taps 0..2, first fragments of tap 0 -------
    unsigned a_goff[A_PER_WAVE];                       // halo source offsets of the tile whose halo is being fetched
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) a_goff[j] = halo_goff(j, img, oy0, ox0);
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) RPN_DMA_A(j, a_goff[j], cb * 128, 0);
#pragma unroll
    for (int t = 0; t < 3; ++t) RPN_DMA_B((tap0 + t) * w_tap_bytes + n0 * (PPP * 16), t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 2 * NJ; ++i) fw[0][i] = RPN_W_ADDR(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) fx0[i] = RPN_X_ADDR(0, 0, 0, 0, i);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // ring slot 0 may now be overwritten (tap 3)

    int gt = 3;                                        // next tap to DMA, counted from the current tile's tap 0
    int tile_no = 0;
    for (;;) {
        int nimg = 0, noy0 = 0, nox0 = 0, nn0 = 0;
        int nxt = -1;
        unsigned ticket = 0;
        if (dyn) {
            if (tid == 0) ticket = atomicAdd(a.sched + xcd, 1u);     // for the next tile; resolved at taps 4..6 below
        } else {
            nxt = next_tile(cur + slot_stride, nimg, noy0, nox0, nn0);
        }
        f32x4 acc[MT][NJ];
        f32x4 t01[KTREE ? MT : 1][KTREE ? NJ : 1], t23[KTREE ? MT : 1][KTREE ? NJ : 1];     // K tree (SplitConvArgs::ktree): l0 + l1, l2
        (void)t01; (void)t23;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
        for (int pair = 0; pair < (chunks >> 1); ++pair) {
#pragma unroll
            for (int T = 0; T < 18; ++T) {            // tap T of this pair of slices; everything below is static in T
                const int c2 = T / 9, s9 = T % 9;
                const int CS = T & 1, CB = c2 & 1, CR = s9 / 3, CC = s9 % 3;                       // this tap
                const int NS = (T + 1) & 1, NB = ((T + 1) / 9) & 1, NR = ((T + 1) % 9) / 3, NC = ((T + 1) % 9) % 3;   // next
                {   // weights of tap t+3: this tile's, or the first taps of the next tile (or a harmless re-load)
                    int tap = gt, nb = n0;
                    if (gt >= total_taps) {
                        if (nxt >= 0) { tap = gt - total_taps; nb = nn0; }
                        else tap = total_taps - 1;
                    }
#ifndef RPN_EXP_NO_DMA
                    RPN_DMA_B((tap0 + tap) * w_tap_bytes + nb * (PPP * 16), s9 % 3);
#endif
                    ++gt;
                }
                if (dyn && pair == 0) {     // (nothing before tap 9 of the last slice needs to know the next tile)
                    if (T == 4 && tid == 0) tile_q[tile_no & 1] = walk.settle((int)ticket, a.sched + xcd);
                    if (T == 6) {           // two barriers later
                        const int sl = __builtin_amdgcn_readfirstlane(tile_q[tile_no & 1]);
                        nxt = (sl < walk.n_slots && walk.decode(sl, nimg, noy0, nox0, nn0)) ? sl : -1;
                    }
                }
                if (T == 9 && nxt >= 0 && pair == (chunks >> 1) - 1) {    // last slice of the tile: from here on the halo
#pragma unroll
                    for (int j = 0; j < A_PER_WAVE; ++j) a_goff[j] = halo_goff(j, nimg, noy0, nox0);   // of the NEXT tile
                }
                if (s9 < A_PER_WAVE) {      // halo piece s9 of the next slice: this tile's, or slice 0 of the next tile
                    const int nc = 2 * pair + c2 + 1;
#ifndef RPN_EXP_NO_DMA
                    RPN_DMA_A(s9, a_goff[s9], (cb + (nc < chunks ? nc : (nxt >= 0 ? 0 : chunks - 1))) * 128, (c2 + 1) & 1);
#endif
                }
                // ---- half 0: row 0.  Program order = intended issue order: a fragment read behind each of the first MFMAs
#pragma unroll
                for (int i = 0; i < 2 * NJ; ++i) {    // chain heads (x lo * w hi) of tiles (m = i / NJ, j = i % NJ)
#ifndef RPN_EXP_NO_LDSREAD
                    if (i < 4) fx1[i] = RPN_X_ADDR(CB, CR, CC, 1, i);
#endif
                    acc[i / NJ][i % NJ] = mfma16<F16>(fx0[2 * (i / NJ) + 1], fw[CS][2 * (i % NJ)], acc[i / NJ][i % NJ]);
                }
#pragma unroll
                for (int i = 0; i < 2 * NJ; ++i) {
                    acc[i / NJ][i % NJ] = mfma16<F16>(fx0[2 * (i / NJ)], fw[CS][2 * (i % NJ) + 1], acc[i / NJ][i % NJ]);   // x hi * w lo
                    acc[i / NJ][i % NJ] = mfma16<F16>(fx0[2 * (i / NJ)], fw[CS][2 * (i % NJ)], acc[i / NJ][i % NJ]);       // x hi * w hi
                }
#if RPN_DMA_SCHED
                if (s9 < A_PER_WAVE) __builtin_amdgcn_sched_group_barrier(0x020, B_PER_WAVE + 1, 0);
                else __builtin_amdgcn_sched_group_barrier(0x020, B_PER_WAVE, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 6 * NJ - 4, 0);
                __builtin_amdgcn_sched_barrier(0);
#endif
                // ---- half 1: row 1; the next tap's weights and row-0 pixels arrive meanwhile
#pragma unroll
                for (int i = 0; i < 2 * NJ; ++i) {
#ifndef RPN_EXP_NO_LDSREAD
                    fw[NS][i] = RPN_W_ADDR(NC, i);
#else
                    fw[NS][i] = fw[CS][i];
#endif
                    acc[2 + i / NJ][i % NJ] = mfma16<F16>(fx1[2 * (i / NJ) + 1], fw[CS][2 * (i % NJ)], acc[2 + i / NJ][i % NJ]);
                }
#pragma unroll
                for (int i = 0; i < 2 * NJ; ++i) {
#ifndef RPN_EXP_NO_LDSREAD
                    if (i < 4) fx0[i] = RPN_X_ADDR(NB, NR, NC, 0, i);
#endif
                    acc[2 + i / NJ][i % NJ] = mfma16<F16>(fx1[2 * (i / NJ)], fw[CS][2 * (i % NJ) + 1], acc[2 + i / NJ][i % NJ]);
                    acc[2 + i / NJ][i % NJ] = mfma16<F16>(fx1[2 * (i / NJ)], fw[CS][2 * (i % NJ)], acc[2 + i / NJ][i % NJ]);
                }
#if RPN_DMA_SCHED
#pragma unroll
                for (int i = 0; i < 2 * NJ; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
                if constexpr (4 * NJ - 8 > 0) __builtin_amdgcn_sched_group_barrier(0x008, 4 * NJ - 8, 0);
                __builtin_amdgcn_sched_barrier(0);     // (register-only MFMAs would otherwise sink below the barrier)
#endif
                // Leave exactly this interval's own DMAs in flight.  Not in a tile's first interval: what it needs (tap 2) was
                // retired by the vmcnt(0) in front of the previous tile's epilogue (or by the prologue), and the in-order
                // counter would otherwise make the wave wait for that epilogue's global stores -- all 256 workgroups
                // finish their tiles together, so those stores are a burst (stamps: 12.9k cycles of epilogue in steady
                // state against 6.5k on a workgroup's first tile).  The stores now have two intervals to drain.
#ifdef RPN_EXP_HALF_BARRIERS                /* timing experiment only (results are wrong): a barrier every second tap */
                if ((T & 1) == 0) continue;
#endif
                if (T == 0 && pair == 0) {
                } else if constexpr (B_PER_WAVE == 2) {
                    if (s9 < A_PER_WAVE) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                } else {
                    if (s9 < A_PER_WAVE) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#if RPN_DMA_SCHED
                __builtin_amdgcn_sched_barrier(0);
#endif
                if (s9 == 8 && tile_no == 1) RPN_STAMP_AT(4 + 2 * pair + c2);   // (second tile: steady state)
                if constexpr (KTREE) {      // a leaf ends behind slice k * slices / 4 (slices >= 8: every leaf has two or more)
                    if (s9 == 8) {          // (tap 8's MFMAs are all issued above; tap 9's fragments are in registers already)
                        const int done = cb + 2 * pair + c2 + 1;     // slices of the LAYER finished
                        const bool e0 = done == ktree_cut(all_chunks, 1), e1 = done == ktree_cut(all_chunks, 2),
                                   e2 = done == ktree_cut(all_chunks, 3);
                        if (e0 || e1 || e2) {
#pragma unroll
                            for (int m = 0; m < MT; ++m)
#pragma unroll
                                for (int j = 0; j < NJ; ++j) {
                                    if (e0) t01[m][j] = acc[m][j];
                                    else if (e1) t01[m][j] = t01[m][j] + acc[m][j];
                                    else t23[m][j] = acc[m][j];
                                    acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                                }
                        }
                    }
                }
            }
        }
        if constexpr (KTREE) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (gridDim.y == 2) acc[m][j] = blockIdx.y == 0 ? t01[m][j] : t23[m][j] + acc[m][j];   // l0 + l1 | l2 + l3
                    else acc[m][j] = t01[m][j] + (t23[m][j] + acc[m][j]);
                }
        }
        // The first fragments of the next tile's tap 0 are already in registers and its DMAs are in flight (halo buffer 0,
        // the weight ring): the epilogue stages through halo buffer 1 + the spare LDS, both dead until the next tile's
        // first interval starts to refill buffer 1 -- hence one workgroup barrier per tile.
        if (tile_no < 2) RPN_STAMP_AT(20 + 6 * tile_no);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last interval's weight DMAs (issued a whole interval ago)
        if (tile_no < 2) RPN_STAMP_AT(21 + 6 * tile_no);
#ifdef RPN_EXP_NO_EPILOGUE                  /* timing experiment only: the accumulators are kept alive, nothing is written */
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(acc[m][j]));
#elif defined(RPN_STAMP)
        split16_epilogue<F16, POOL, RW, NW, NJ>(acc, reinterpret_cast<float *>(lds + A1_AT), a, img, oy0, ox0, n0, wave, wm, wn, lane,
                                                tile_no < 2 ? 22 + 6 * tile_no : -1);
#else
        split16_epilogue<F16, POOL, RW, NW, NJ>(acc, reinterpret_cast<float *>(lds + A1_AT), a, img, oy0, ox0, n0, wave, wm, wn, lane);
#endif
        if (tile_no == 1) RPN_STAMP_AT(3);
        if (nxt < 0) break;
        __builtin_amdgcn_s_barrier();
        if (tile_no == 0) RPN_STAMP_AT(2);
        cur = nxt; img = nimg; oy0 = noy0; ox0 = nox0; n0 = nn0;
        gt -= total_taps;
        ++tile_no;
    }
#undef RPN_X_ADDR
#undef RPN_W_ADDR
#undef next_tile
#undef halo_goff
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail DMAs still target this workgroup's LDS
    RPN_SCHED_EXIT();
#undef RPN_SCHED_EXIT
#undef RPN_DMA_A
#undef RPN_DMA_B
}

// (LABORATORY BUILD ONLY, -DRPN_LAB: measured SLOWER than the 8-wave form and not part of the product library -- see the note at
// the end of this comment and NOTES.md.)
#ifdef RPN_LAB
// ---- 16x16x32-MFMA variant: persistent workgroups, LDS-DMA pipeline, ONE WAVE PER SIMD, epilogue under the next tile's taps ----
// conv3x3_split16_dma_kernel<.., 128> loses ~9 % of the VGG16 step to its epilogue (-DRPN_EXP_NO_EPILOGUE): at the end of a tile
// all eight waves stage and store 128 KB while the matrix pipes idle, and that store phase is bounded by store ISSUE on the CU
// (~14 B / clk with 16-byte stores), not by HBM and not by workgroups colliding (a staggered start changes nothing).  Hiding it
// needs an MFMA stream that stays resident while the stores drain, i.e. a second accumulator set: 64 more registers per wave
// that the 8-wave form (256 registers per wave at two waves per SIMD) does not have.  Here the same 8 x 32 px x 128 ch tile runs
// on FOUR waves (one per SIMD, up to 512 registers): a wave owns 4 rows x 32 px x 64 channels = 32 accumulator tiles
// (128 registers) and works a tap in four quarter-steps of 24 MFMAs (one pixel row each; row fragments double-buffered A / B,
// the next tap's weights and row 0 read during the fourth).  At the end of a tile the accumulators are COPIED to a second set
// and the tap stream continues into the next tile at once; the finished tile leaves in 8 micro-steps (4 with the fused pool),
// one per tap of the next tile's first slice: 16 px x 64 ch per wave -> + bias, activation -> a wave-private staging area that
// lives BEHIND halo buffer 1 (17 KB: nothing of the pipeline overlays it) -> hi / lo split -> four 16-byte stores per lane,
// scheduled among the first two quarter-steps' MFMAs.  The stores are issued IN FRONT of the interval's DMAs, so the interval's
// counted wait -- vmcnt(number of its own DMAs) -- retires them together with the previous interval's DMAs whatever their number.
// Same products, accumulated in the same order as the 8-wave form: the same bits.
// Static tile schedule only (RPN_S16_DYN keeps the 8-wave kernel).
// Measured (VGG16, batch 8, same device, RPN_S16_W4 = 0 | 1): parity tests green at the first run (same bits as the 8-wave form,
// repeats bit-identical), but 2 880 -> 2 660 images/s: block2_conv2 0.302 -> 0.337 ms, block4_conv2 0.283 -> 0.314.  With the
// epilogue left synchronous at the tile end (-DRPN_EXP_W4_SYNC_EPI) the same kernel runs 0.332 / 0.307 ms: the four-wave TAP LOOP by
// itself is 8-10 % slower than the eight-wave one -- with one wave per SIMD nothing covers a wave's own counted waits and the
// barrier skew of every tap -- and that is more than the epilogue it could hide.
template <bool F16, bool POOL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
conv3x3_split16_dma4_kernel(SplitConvArgs a, int tiles_x, int tiles_y, int n_tiles)
{
    constexpr int BN = 128, TH = 8, NW = 4, RW = 4, MT = 8, NJ = 4, PPP = 8;
    constexpr int B_PER_WAVE = BN * PPP / 64 / NW;     // 4 weight DMA instructions per wave and tap
    constexpr int HP = (TH + 2) * HW;                  // 340 halo pixels
    constexpr int A_INSTR = (HP * PPP + 63) / 64;      // 43 wave-instructions (1 KB each) per halo tile
    constexpr int ABUF = A_INSTR * 64;
    constexpr int A_PER_WAVE = 11;                     // (the one surplus instruction of the last round duplicates piece 0)
    constexpr int BSLOT = BN * PPP;
    constexpr int B_AT = ABUF, DUMP = B_AT + 3 * BSLOT, A1_AT = DUMP + 64;
    constexpr int STAGE_AT = A1_AT + ABUF;             // staging: 4 waves x 16 px x 68 floats, behind halo buffer 1
    constexpr int STAGE_LD = 16 * NJ + kStagePad;
    constexpr int STAGE_PIECES = NW * 16 * STAGE_LD * 4 / 16;
    constexpr int LDS_UINT4 = STAGE_AT + STAGE_PIECES;
    static_assert((A_INSTR + NW - 1) / NW == A_PER_WAVE && LDS_UINT4 * 16 <= 160 * 1024, "tile shape");

    __shared__ uint4 lds[LDS_UINT4];
    u32x4 *As = reinterpret_cast<u32x4 *>(lds);       // halo buffer b at As + b * A1_AT
    u32x4 *Bs = reinterpret_cast<u32x4 *>(lds) + B_AT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, kg = lane >> 4;

    const int m_tiles = tiles_x * tiles_y * a.B;
    const int XN = n_tiles >= 8 ? 8 : (n_tiles >= 4 ? 4 : (n_tiles >= 2 ? 2 : 1)), XM = 8 / XN;
    const int NTl = (n_tiles + XN - 1) / XN, MTl = (m_tiles + XM - 1) / XM;
    const int xcd = blockIdx.x & 7;
    const int slot_stride = gridDim.x >> 3;
    const TileWalk walk{n_tiles, m_tiles, tiles_x, tiles_y, XN, XM, NTl, NTl * MTl, xcd, slot_stride, BN};

    const int chunks = a.Cin >> 5;                    // 32-channel slices (even)
    const int total_taps = chunks * 9;
    const int in_pix_stride = (a.Cin >> 4) * 4;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(a.x), (short)0, (int)((size_t)a.B * a.H * a.W * in_pix_stride * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(a.w), (short)0, (int)((size_t)total_taps * a.cout_pad * 128), 0x00020000);
    const int w_tap_bytes = a.cout_pad * 128;

#define halo_goff4(J, IM, Y0, X0) halo_source_offset<NW, A_INSTR, HP * PPP>((J), wave, lane, (IM), (Y0), (X0), a.H, a.W, in_pix_stride)
    const unsigned b_voff = (unsigned)(((B_PER_WAVE * wave) * 64 + lane) * 16);
#define RPN4_DMA_A(J, GOFF, SOFF, BUF)                                                                            \
    {                                                                                                             \
        const int k0_ = (J) * NW + wave, k_ = k0_ < A_INSTR ? k0_ : k0_ - A_INSTR;                                \
        u32x4 *dst_ = As + (BUF) * A1_AT + k_ * 64;                                                               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, RPN_LDS_PTR(dst_), 16, (GOFF), (SOFF), 0, 0);             \
    }
#define RPN4_DMA_B(SOFF, SLOT)                                                                                    \
    {                                                                                                             \
        u32x4 *dst_ = Bs + (SLOT) * BSLOT + (B_PER_WAVE * wave) * 64;                                             \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, RPN_LDS_PTR(dst_), 16, b_voff, (SOFF), 0, 0);             \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, RPN_LDS_PTR(dst_), 16, b_voff, (SOFF), 1024, 0);          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, RPN_LDS_PTR(dst_), 16, b_voff, (SOFF), 2048, 0);          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, RPN_LDS_PTR(dst_), 16, b_voff, (SOFF), 3072, 0);          \
    }

    int a_off[3][2];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int hx = 16 * hf + lr + s3;
            a_off[s3][hf] = hx * PPP + (kg ^ halo_swz(hx));
        }
    int b_off[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = wn * (16 * NJ) + j * 16 + lr;
        b_off[j] = n * PPP + (kg ^ ((n >> 1) & 7));
    }

    int img = 0, oy0 = 0, ox0 = 0, n0 = 0;
    int cur = walk.next(blockIdx.x >> 3, img, oy0, ox0, n0);
    if (cur < 0) return;

    u32x4 fxA[4], fxB[4], fw[2][2 * NJ];              // fx: [16-px half * 2 + (0 hi | 1 lo)] of one pixel row;  fw: [set][j * 2 + (0 hi | 1 lo)]
#define RPN4_X_ADDR(BUF, ROW, S, I, II)                                                                          \
    (As[(BUF) * A1_AT + (RW * wm + (ROW) + (I)) * HW * PPP + (a_off[S][(II) >> 1] ^ (((II) & 1) ? 4 : 0))])
#define RPN4_W_ADDR(S, II) (Bs[(S) * BSLOT + (b_off[(II) >> 1] ^ (((II) & 1) ? 4 : 0))])

    // ---- prologue (first tile only): halo tile of slice 0, weights of taps 0..2, first fragments of tap 0 -------
    unsigned a_goff[A_PER_WAVE];
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) a_goff[j] = halo_goff4(j, img, oy0, ox0);
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) RPN4_DMA_A(j, a_goff[j], 0, 0);
#pragma unroll
    for (int t = 0; t < 3; ++t) RPN4_DMA_B(t * w_tap_bytes + n0 * (PPP * 16), t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 2 * NJ; ++i) fw[0][i] = RPN4_W_ADDR(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) fxA[i] = RPN4_X_ADDR(0, 0, 0, 0, i);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // ring slot 0 may now be overwritten (tap 3)

    // the finished tile whose epilogue is pending: its accumulators and where they go
    f32x4 done[MT][NJ];
    float d_bias[NJ];
    int d_img = 0, d_oy0 = 0, d_ox0 = 0, d_n0 = 0;
    bool pending = false;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NJ; ++j) done[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j) d_bias[j] = 0.0f;
    float *stage = reinterpret_cast<float *>(lds + STAGE_AT) + wave * (16 * STAGE_LD);
    const long long out_bytes = (long long)a.B * (POOL ? a.H >> 1 : a.H) * (POOL ? a.W >> 1 : a.W) * a.Cout * 4;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)out_bytes, 0x00020000);
    const float act_lo = a.act == ACT_LINEAR ? -INFINITY : 0.0f;
    const float act_hi = a.act == ACT_RELU6 ? 6.0f : INFINITY;
    const int cout_chunks = a.Cout >> 4;
    constexpr int PP = 16 * NJ / 4;                   // 16-byte pieces (or float4s) per pixel of the wave's 64 channels

    // micro-step STEP of the pending epilogue, part 1: 16 px x 64 ch (one row half; POOL: 8 pooled px of a row pair's half)
    // -> * scale + bias, activation -> the wave's staging area
#define RPN4_EPI_STAGE(STEP)                                                                                       \
    {                                                                                                              \
        if constexpr (POOL) {                                                                                      \
            const int st_ = (STEP) & 3, rp_ = st_ >> 1, h_ = st_ & 1;      /* (& 3: dead unrolled copies stay in range) */                                                      \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                         \
                _Pragma("unroll") for (int r2 = 0; r2 < 2; ++r2) {                                                 \
                    const float v0 = fmaxf(done[(2 * rp_) * 2 + h_][j][2 * r2], done[(2 * rp_) * 2 + h_][j][2 * r2 + 1]);            \
                    const float v1 = fmaxf(done[(2 * rp_ + 1) * 2 + h_][j][2 * r2], done[(2 * rp_ + 1) * 2 + h_][j][2 * r2 + 1]);    \
                    stage[(2 * kg + r2) * STAGE_LD + j * 16 + lr] =                                                \
                        fminf(fmaxf(fmaxf(v0, v1) * a.out_scale + d_bias[j], act_lo), act_hi);                     \
                }                                                                                                  \
        } else {                                                                                                   \
            const int st_ = (STEP) & 7, r_ = st_ >> 1, h_ = st_ & 1;                                                       \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                         \
                _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                      \
                    stage[(4 * kg + e) * STAGE_LD + j * 16 + lr] =                                                 \
                        fminf(fmaxf(done[r_ * 2 + h_][j][e] * a.out_scale + d_bias[j], act_lo), act_hi);           \
        }                                                                                                          \
        wave_sync();                                                                                               \
    }
    // ... part 2: staging area -> hi / lo split (or float32) -> 16-byte stores.  Through a raw buffer descriptor with 32-bit
    // byte offsets (the launcher checks that the output tensor stays below 2 GiB): pixels or channels beyond the tensor get an
    // out-of-range offset, which the hardware drops -- no branch, and no 64-bit address pair per lane and round for the register
    // allocator to carry across the tap loop (the first version spilled 42 of them).
#define RPN4_EPI_STORE(STEP)                                                                                       \
    {                                                                                                              \
        constexpr int NPX_ = POOL ? 8 : 16;                                                                        \
        const int st_ = (STEP) & (POOL ? 3 : 7), h_ = st_ & 1;                                                     \
        const int OHo_ = POOL ? a.H >> 1 : a.H, OWo_ = POOL ? a.W >> 1 : a.W;                                      \
        const int oy_ = POOL ? (d_oy0 + RW * wm + 2 * (st_ >> 1)) >> 1 : d_oy0 + RW * wm + (st_ >> 1);             \
        const int oxb_ = POOL ? (d_ox0 >> 1) + 8 * h_ : d_ox0 + 16 * h_;                                           \
        const int nbase_ = d_n0 + wn * (16 * NJ);                                                                  \
        const int row_ = (d_img * OHo_ + oy_) * OWo_;                                                              \
        int ln_ = lane;                                                                                            \
        asm volatile("" : "+v"(ln_));                                                                              \
        _Pragma("unroll") for (int rd = 0; rd < NPX_ * PP / 64; ++rd) {                                            \
            const int e = rd * 64 + ln_;                                                                           \
            const int px = e / PP, q = e % PP;                                                                     \
            const int ox = oxb_ + px;                                                                              \
            if (a.out_f32) {                                                                                       \
                const int n = nbase_ + 4 * q;                                                                      \
                const bool ok = oy_ < OHo_ && ox < OWo_ && n < a.Cout;                                             \
                const u32x4 v = *reinterpret_cast<const u32x4 *>(&stage[px * STAGE_LD + 4 * q]);                   \
                __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, ok ? ((row_ + ox) * a.Cout + n) * 4 : (int)0x80000000, 0, 0);   \
            } else {                                                                                               \
                const int cl = q >> 2, pc = q & 3;                                                                 \
                const int n = nbase_ + cl * 16;                                                                    \
                const bool ok = oy_ < OHo_ && ox < OWo_ && n < a.Cout;                                             \
                float xs[8];                                                                                       \
                const float *src = &stage[px * STAGE_LD + cl * 16 + (pc >> 1) * 8];                                \
                const float4 v0 = *reinterpret_cast<const float4 *>(src);                                          \
                const float4 v1 = *reinterpret_cast<const float4 *>(src + 4);                                      \
                xs[0] = v0.x; xs[1] = v0.y; xs[2] = v0.z; xs[3] = v0.w;                                            \
                xs[4] = v1.x; xs[5] = v1.y; xs[6] = v1.z; xs[7] = v1.w;                                            \
                const uint4 pv = split_piece<F16>(xs, (pc & 1) != 0, ok ? a.status : nullptr);                     \
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{pv.x, pv.y, pv.z, pv.w}, orsrc,                       \
                    ok ? (((row_ + ox) * cout_chunks + (n >> 4)) * 4 + pc) * 16 : (int)0x80000000, 0, 0);          \
            }                                                                                                      \
        }                                                                                                          \
        wave_sync();                                                                                               \
    }
    constexpr int EPI_STEPS = POOL ? 4 : 8;

    // one pixel row (quarter-step): 24 MFMAs of row Q on fragment set F, accumulating into acc[2 Q .. 2 Q + 1][*]
#define RPN4_ROW_MFMA(Q, F)                                                                                        \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 2 * NJ; ++i)                                                         \
            acc[2 * (Q) + i / NJ][i % NJ] = mfma16<F16>(F[2 * (i / NJ) + 1], fw[CS][2 * (i % NJ)], acc[2 * (Q) + i / NJ][i % NJ]);   \
        _Pragma("unroll") for (int i = 0; i < 2 * NJ; ++i) {                                                       \
            acc[2 * (Q) + i / NJ][i % NJ] = mfma16<F16>(F[2 * (i / NJ)], fw[CS][2 * (i % NJ) + 1], acc[2 * (Q) + i / NJ][i % NJ]);   \
            acc[2 * (Q) + i / NJ][i % NJ] = mfma16<F16>(F[2 * (i / NJ)], fw[CS][2 * (i % NJ)], acc[2 * (Q) + i / NJ][i % NJ]);       \
        }                                                                                                          \
    }

    int gt = 3;                                        // next tap to DMA, counted from the current tile's tap 0
    for (;;) {
        int nimg = 0, noy0 = 0, nox0 = 0, nn0 = 0;
        const int nxt = walk.next(cur + slot_stride, nimg, noy0, nox0, nn0);
        f32x4 acc[MT][NJ];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
        for (int pair = 0; pair < (chunks >> 1); ++pair) {
#ifdef RPN_EXP_W4_SYNC_EPI
            const bool epi = false;
#else
            const bool epi = pending && pair == 0;    // this slice carries the previous tile's epilogue
#endif
#pragma unroll
            for (int T = 0; T < 18; ++T) {            // tap T of this pair of slices; everything below is static in T
                const int c2 = T / 9, s9 = T % 9;
                const int CS = T & 1, CB = c2 & 1, CR = s9 / 3, CC = s9 % 3;                       // this tap
                const int NS = (T + 1) & 1, NB = ((T + 1) / 9) & 1, NR = ((T + 1) % 9) / 3, NC = ((T + 1) % 9) % 3;   // next
                const int nA = s9 < 5 ? 2 : (s9 == 5 ? 1 : 0);                                     // halo pieces issued in this tap
                if (T == 9 && nxt >= 0 && pair == (chunks >> 1) - 1) {    // last slice of the tile: from here on the halo
#pragma unroll
                    for (int j = 0; j < A_PER_WAVE; ++j) a_goff[j] = halo_goff4(j, nimg, noy0, nox0);   // of the NEXT tile
                }
                // ---- quarter 0: row 0 (fxA); reads row 1 -> fxB.  Epilogue micro-step T - 1, part 1, among its MFMAs --------
                if (T >= 1 && T <= EPI_STEPS && epi) RPN4_EPI_STAGE(T - 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) fxB[i] = RPN4_X_ADDR(CB, CR, CC, 1, i);
                RPN4_ROW_MFMA(0, fxA);
                __builtin_amdgcn_sched_barrier(0);
                // ---- quarter 1: row 1 (fxB); reads row 2 -> fxA.  Epilogue part 2: the stores, IN FRONT of this tap's DMAs ----
                if (T >= 1 && T <= EPI_STEPS && epi) RPN4_EPI_STORE(T - 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) fxA[i] = RPN4_X_ADDR(CB, CR, CC, 2, i);
                RPN4_ROW_MFMA(1, fxB);
                __builtin_amdgcn_sched_barrier(0);
                {   // weights of tap t+3: this tile's, or the first taps of the next tile (or a harmless re-load)
                    int tap = gt, nb = n0;
                    if (gt >= total_taps) {
                        if (nxt >= 0) { tap = gt - total_taps; nb = nn0; }
                        else tap = total_taps - 1;
                    }
                    RPN4_DMA_B(tap * w_tap_bytes + nb * (PPP * 16), s9 % 3);
                    ++gt;
                }
                if (nA > 0) {               // halo pieces of the next slice: this tile's, or slice 0 of the next tile
                    const int nc = 2 * pair + c2 + 1;
                    const int soff = nc < chunks ? nc * 128 : (nxt >= 0 ? 0 : (chunks - 1) * 128);
                    if (s9 < 5) {
                        RPN4_DMA_A(2 * s9, a_goff[2 * s9], soff, (c2 + 1) & 1);
                        RPN4_DMA_A(2 * s9 + 1, a_goff[2 * s9 + 1], soff, (c2 + 1) & 1);
                    } else {
                        RPN4_DMA_A(10, a_goff[10], soff, (c2 + 1) & 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- quarter 2: row 2 (fxA); reads row 3 -> fxB -------------------------------------------------------------
#pragma unroll
                for (int i = 0; i < 4; ++i) fxB[i] = RPN4_X_ADDR(CB, CR, CC, 3, i);
                RPN4_ROW_MFMA(2, fxA);
                __builtin_amdgcn_sched_barrier(0);
                // ---- quarter 3: row 3 (fxB); the next tap's weights and row 0 arrive meanwhile --------------------------------
#pragma unroll
                for (int i = 0; i < 2 * NJ; ++i) fw[NS][i] = RPN4_W_ADDR(NC, i);
#pragma unroll
                for (int i = 0; i < 4; ++i) fxA[i] = RPN4_X_ADDR(NB, NR, NC, 0, i);
                RPN4_ROW_MFMA(3, fxB);
                __builtin_amdgcn_sched_barrier(0);
                // leave exactly this interval's own DMAs in flight (the epilogue's stores were issued in front of them)
                if (nA == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (nA == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // a tile with ONE pair of slices (Cin = 64) only carries the epilogue's micro-steps that fit its 18 taps: all of them
        // (8 <= 17); the pending tile is finished before its registers are overwritten
        // ---- tile done: hand the accumulators over, go on with the next tile ------------------------------------------------
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < NJ; ++j) done[m][j] = acc[m][j];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + wn * (16 * NJ) + j * 16 + lr;
            d_bias[j] = (a.bias && n < a.Cout) ? a.bias[n] : 0.0f;
        }
        d_img = img; d_oy0 = oy0; d_ox0 = ox0; d_n0 = n0;
        pending = true;
#ifdef RPN_EXP_W4_SYNC_EPI
        RPN4_EPI_STAGE(0); RPN4_EPI_STORE(0);
        RPN4_EPI_STAGE(1); RPN4_EPI_STORE(1);
        RPN4_EPI_STAGE(2); RPN4_EPI_STORE(2);
        RPN4_EPI_STAGE(3); RPN4_EPI_STORE(3);
        if constexpr (!POOL) {
            RPN4_EPI_STAGE(4); RPN4_EPI_STORE(4);
            RPN4_EPI_STAGE(5); RPN4_EPI_STORE(5);
            RPN4_EPI_STAGE(6); RPN4_EPI_STORE(6);
            RPN4_EPI_STAGE(7); RPN4_EPI_STORE(7);
        }
        pending = false;
#endif
        if (nxt < 0) break;
        cur = nxt; img = nimg; oy0 = noy0; ox0 = nox0; n0 = nn0;
        gt -= total_taps;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail DMAs still target this workgroup's LDS
    // the last tile's epilogue, nothing left to hide it under
    if (pending) {
        RPN4_EPI_STAGE(0); RPN4_EPI_STORE(0);
        RPN4_EPI_STAGE(1); RPN4_EPI_STORE(1);
        RPN4_EPI_STAGE(2); RPN4_EPI_STORE(2);
        RPN4_EPI_STAGE(3); RPN4_EPI_STORE(3);
        if constexpr (!POOL) {
            RPN4_EPI_STAGE(4); RPN4_EPI_STORE(4);
            RPN4_EPI_STAGE(5); RPN4_EPI_STORE(5);
            RPN4_EPI_STAGE(6); RPN4_EPI_STORE(6);
            RPN4_EPI_STAGE(7); RPN4_EPI_STORE(7);
        }
    }
#undef RPN4_ROW_MFMA
#undef RPN4_EPI_STORE
#undef RPN4_EPI_STAGE
#undef RPN4_X_ADDR
#undef RPN4_W_ADDR
#undef RPN4_DMA_A
#undef RPN4_DMA_B
#undef halo_goff4
}
#endif  // RPN_LAB

static inline unsigned short f32_to_bf16_rne(float f);
static inline float bf16_to_f32(unsigned short h);

// ---- first layer on the matrix cores (split-precision modes) ---------------------------------------------------
// Cin = 3, 3x3: K = 27 is padded to 32 = ONE v_mfma_f32_16x16x32 step.  Workgroup = 8 x 32 output pixels x Cout
// (Cout = 16 * NTILES <= 64), 4 waves, each 2 rows.  The float32 input patch is staged in LDS; every lane gathers
// the 8 patch values of its (pixel, k-group) from LDS and splits them into hi / lo halves on the fly (the A
// fragment of the im2col matrix, never materialised); the pre-split weights are read from LDS once and stay in
// registers.  3 MFMAs per (16 px x 16 ch) tile.  ~4x fewer vector-ALU instructions per output than the direct
// kernel, so the layer becomes bound by its 4*Cout bytes per pixel of output.
template <bool F16, bool OUT_SPLIT, int STRIDE, int NTILES>
__global__ void __launch_bounds__(256)
conv_cin3_mfma_kernel(const float *__restrict__ x, const uint4 *__restrict__ w /* [Cout_pad][8 pieces] */,
                      const float *__restrict__ bias, void *__restrict__ out, int B, int H, int W, int OH, int OW,
                      int pad_t, int pad_l, int act, float out_scale, int tiles_x, int tiles_y, unsigned *status)
{
    constexpr int COUT = 16 * NTILES;
    constexpr int PR = 7 * STRIDE + 3, PC = 31 * STRIDE + 3;            // input patch rows / columns
    constexpr int PATCH = PR * PC * 3;
    constexpr int STAGE_LD = COUT + kStagePad;
    __shared__ __attribute__((aligned(16))) float patch[(PATCH + 3) / 4 * 4];
    __shared__ uint4 wl[COUT * 8];
    __shared__ __attribute__((aligned(16))) float stage_all[4 * 32 * STAGE_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, kg = lane >> 4;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int img = t / tiles_y;
    const int oy0 = ty * 8, ox0 = tx * 32;
    const int iy0 = oy0 * STRIDE - pad_t, ix0 = ox0 * STRIDE - pad_l;
    const float *ximg = x + (size_t)img * H * W * 3;

    for (int e = tid; e < PATCH; e += 256) {                             // patch rows are contiguous in the image
        const int pr = e / (PC * 3), rem = e - pr * (PC * 3);
        const int iy = iy0 + pr, ix = ix0 + rem / 3;
        patch[e] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? ximg[((size_t)iy * W + ix) * 3 + rem % 3] : 0.0f;
    }
    for (int e = tid; e < COUT * 8; e += 256) {
        const int n = e >> 3, pc = e & 7;
        reinterpret_cast<u32x4 *>(wl)[n * 8 + (pc ^ ((n >> 1) & 7))] = reinterpret_cast<const u32x4 *>(w)[e];
    }
    // k = 8 kg + j = (r * 3 + s) * 3 + c  ->  offset inside the patch relative to the pixel's top-left input
    int koff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * kg + j;
        koff[j] = k < 27 ? (k / 9) * (PC * 3) + (k % 9) : -1;
    }
    __syncthreads();

    u32x4 bhi[NTILES], blo[NTILES];
#pragma unroll
    for (int j = 0; j < NTILES; ++j) {
        const int n = j * 16 + lr;
        const int idx = n * 8 + (kg ^ ((n >> 1) & 7));
        bhi[j] = reinterpret_cast<const u32x4 *>(wl)[idx];
        blo[j] = reinterpret_cast<const u32x4 *>(wl)[idx ^ 4];
    }
    f32x4 acc[4][NTILES];                                                // [row i * 2 + half][N tile]
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int prow = (2 * wave + (m >> 1)) * STRIDE, pcol = (16 * (m & 1) + lr) * STRIDE;
        const int base = (prow * PC + pcol) * 3;
        float xs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xs[j] = koff[j] >= 0 ? patch[base + koff[j]] : 0.0f;
        const u32x4 ahi = __builtin_bit_cast(u32x4, split_piece<F16>(xs, false));
        const u32x4 alo = __builtin_bit_cast(u32x4, split_piece<F16>(xs, true));
#pragma unroll
        for (int j = 0; j < NTILES; ++j) {
            f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
            c = mfma16<F16>(alo, bhi[j], c);
            c = mfma16<F16>(ahi, blo[j], c);
            acc[m][j] = mfma16<F16>(ahi, bhi[j], c);
        }
    }

    float *stage = stage_all + wave * (32 * STAGE_LD);
    float bias_v[NTILES];
#pragma unroll
    for (int j = 0; j < NTILES; ++j) bias_v[j] = bias ? bias[j * 16 + lr] : 0.0f;
    const float act_lo = act == ACT_LINEAR ? -INFINITY : 0.0f;
    const float act_hi = act == ACT_RELU6 ? 6.0f : INFINITY;
    constexpr int PPX = COUT / 4;                                        // 16-byte pieces per pixel (both formats)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int j = 0; j < NTILES; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    stage[(16 * hf + 4 * kg + r) * STAGE_LD + j * 16 + lr] =
                        fminf(fmaxf(acc[i * 2 + hf][j][r] * out_scale + bias_v[j], act_lo), act_hi);
        __syncthreads();
        const int oy = oy0 + 2 * wave + i;
        if (oy < OH) {
            for (int e = lane; e < 32 * PPX; e += 64) {
                const int px = e / PPX, q = e - px * PPX;
                const int ox = ox0 + px;
                if (ox >= OW) continue;
                const size_t pix = ((size_t)img * OH + oy) * OW + ox;
                if constexpr (OUT_SPLIT) {
                    const int cl = q >> 2, pc = q & 3;
                    float xs[8];
                    const float *src = &stage[px * STAGE_LD + cl * 16 + (pc >> 1) * 8];
                    const float4 v0 = *reinterpret_cast<const float4 *>(src);
                    const float4 v1 = *reinterpret_cast<const float4 *>(src + 4);
                    xs[0] = v0.x; xs[1] = v0.y; xs[2] = v0.z; xs[3] = v0.w;
                    xs[4] = v1.x; xs[5] = v1.y; xs[6] = v1.z; xs[7] = v1.w;
                    reinterpret_cast<uint4 *>(out)[pix * PPX + q] = split_piece<F16>(xs, (pc & 1) != 0, status);
                } else {
                    reinterpret_cast<float4 *>(out)[pix * PPX + q] =
                        *reinterpret_cast<const float4 *>(&stage[px * STAGE_LD + 4 * q]);
                }
            }
        }
        __syncthreads();
    }
}

// weights of the MFMA first layer: HWIO (3,3,3,Cout) -> [cout_pad][8 pieces] (k = (r*3+s)*3+c, zero for k >= 27)
void pack_weights_cin3_mfma_host(const float *hwio, const float *scale, int Cout, int cout_pad, bool f16, int shift,
                                 unsigned short *dst /* [cout_pad][64] */)
{
    const float mul = ldexpf(1.0f, shift);
    memset(dst, 0, (size_t)cout_pad * 64 * sizeof(unsigned short));
    for (int k = 0; k < 27; ++k)
        for (int n = 0; n < Cout; ++n) {
            float v = hwio[(size_t)k * Cout + n];
            if (scale) v *= scale[n];
            v *= mul;
            unsigned short hi, lo;
            if (f16) {
                const _Float16 h = (_Float16)v;
                const _Float16 l = (_Float16)(v - (float)h);
                memcpy(&hi, &h, 2);
                memcpy(&lo, &l, 2);
            } else {
                hi = f32_to_bf16_rne(v);
                lo = f32_to_bf16_rne(v - bf16_to_f32(hi));
            }
            unsigned short *rec = dst + (size_t)n * 64;
            rec[(k >> 3) * 8 + (k & 7)] = hi;
            rec[(4 + (k >> 3)) * 8 + (k & 7)] = lo;
        }
}

hipError_t launch_conv_cin3_mfma(const float *x, const void *w, const float *bias, void *out, int B, int H, int W,
                                 int OH, int OW, int Cout, int stride, int pad_t, int pad_l, int act, float out_scale,
                                 int out_fmt, bool f16, hipStream_t s)
{
    if ((Cout != 32 && Cout != 64) || (stride != 1 && stride != 2) || act == ACT_SIGMOID) return hipErrorInvalidValue;
    const int tiles_x = (OW + 31) / 32, tiles_y = (OH + 7) / 8;
    const long long nblocks = (long long)tiles_x * tiles_y * B;
    if (nblocks <= 0 || nblocks > 0x7fffffffll) return hipErrorInvalidValue;
#define RPN_C3M(F16_, SPLIT_, STRIDE_, NT_)                                                                          \
    hipLaunchKernelGGL((conv_cin3_mfma_kernel<F16_, SPLIT_, STRIDE_, NT_>), dim3((unsigned)nblocks), dim3(256), 0, s, x, \
                       (const uint4 *)w, bias, out, B, H, W, OH, OW, pad_t, pad_l, act, out_scale, tiles_x, tiles_y, range_status())
#define RPN_C3M_FMT(F16_, STRIDE_, NT_) \
    { if (out_fmt) RPN_C3M(F16_, true, STRIDE_, NT_); else RPN_C3M(F16_, false, STRIDE_, NT_); }
#define RPN_C3M_ST(F16_, NT_) \
    { if (stride == 1) RPN_C3M_FMT(F16_, 1, NT_) else RPN_C3M_FMT(F16_, 2, NT_) }
    if (f16) { if (Cout == 64) RPN_C3M_ST(true, 4) else RPN_C3M_ST(true, 2) }
    else     { if (Cout == 64) RPN_C3M_ST(false, 4) else RPN_C3M_ST(false, 2) }
#undef RPN_C3M_ST
#undef RPN_C3M_FMT
#undef RPN_C3M
    return hipGetLastError();
}

// ---- float32 NHWC <-> SPLIT16 ---------------------------------------------------------------
template <bool F16>
__global__ void __launch_bounds__(256)
f32_to_split_kernel(const float *__restrict__ x, long long npieces, uint4 *__restrict__ out, unsigned *status)
{
    // piece p: pixel-major; 4 pieces per 16 channels; piece (g*2 + lo) covers channels 8g..8g+7
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npieces; p += (long long)gridDim.x * 256) {
        const long long rec = p >> 2;                 // (pixel, chunk) record
        const int pc = (int)(p & 3);
        const float *src = x + rec * 16 + (pc >> 1) * 8;
        const float4 v0 = *reinterpret_cast<const float4 *>(src);
        const float4 v1 = *reinterpret_cast<const float4 *>(src + 4);
        const float xs[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        out[p] = split_piece<F16>(xs, (pc & 1) != 0, status);
    }
}

template <bool F16>
__global__ void __launch_bounds__(256)
split_to_f32_kernel(const uint4 *__restrict__ x, long long nhalf, float *__restrict__ out)
{
    // one thread per 8 channels: pieces (2g, 2g+1) -> 8 floats
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < nhalf; p += (long long)gridDim.x * 256) {
        const uint4 hi = x[2 * p], lo = x[2 * p + 1];
        const unsigned h[4] = {hi.x, hi.y, hi.z, hi.w}, l[4] = {lo.x, lo.y, lo.z, lo.w};
        float o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[2 * k] = join<F16>((unsigned short)(h[k] & 0xffffu), (unsigned short)(l[k] & 0xffffu));
            o[2 * k + 1] = join<F16>((unsigned short)(h[k] >> 16), (unsigned short)(l[k] >> 16));
        }
        float4 *dst = reinterpret_cast<float4 *>(out + p * 8);
        dst[0] = make_float4(o[0], o[1], o[2], o[3]);
        dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
}

// MaxPooling2D(2,2) 'valid' on SPLIT16: compare hi+lo, keep the winner's (hi, lo) pair
template <bool F16>
__global__ void __launch_bounds__(256)
maxpool_split_kernel(const uint4 *__restrict__ x, int H, int W, int G, int OH, int OW, long long total,
                     uint4 *__restrict__ out)
{
    // G = C/8 groups of 8 channels per pixel; thread = one (pixel, group): reads pieces (2g, 2g+1)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % G);
        long long t = i / G;
        const int ox = (int)(t % OW);
        t /= OW;
        const int oy = (int)(t % OH);
        const long long b = t / OH;
        unsigned bh[4] = {0, 0, 0, 0}, bl[4] = {0, 0, 0, 0};
        float best[8];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const size_t pix = ((size_t)b * H + 2 * oy + dy) * W + 2 * ox + dx;
                const uint4 hi = x[(pix * G + g) * 2], lo = x[(pix * G + g) * 2 + 1];
                const unsigned h[4] = {hi.x, hi.y, hi.z, hi.w}, l[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned short hs = (unsigned short)((k & 1) ? (h[k >> 1] >> 16) : (h[k >> 1] & 0xffffu));
                    const unsigned short ls = (unsigned short)((k & 1) ? (l[k >> 1] >> 16) : (l[k >> 1] & 0xffffu));
                    const float v = join<F16>(hs, ls);
                    if ((dy == 0 && dx == 0) || v > best[k]) {
                        best[k] = v;
                        const unsigned sh = (k & 1) ? 16u : 0u;
                        const unsigned m = 0xffffu << sh;
                        bh[k >> 1] = (bh[k >> 1] & ~m) | ((unsigned)hs << sh);
                        bl[k >> 1] = (bl[k >> 1] & ~m) | ((unsigned)ls << sh);
                    }
                }
            }
        const size_t opix = ((size_t)b * OH + oy) * OW + ox;
        out[(opix * G + g) * 2] = make_uint4(bh[0], bh[1], bh[2], bh[3]);
        out[(opix * G + g) * 2 + 1] = make_uint4(bl[0], bl[1], bl[2], bl[3]);
    }
}

// ---- first layer: Cin = 3, 3x3, stride 1 or 2 (VGG16 block1_conv1, MobileNetV2 Conv1) -------------
// K = 27 is too short for the matrix cores to pay off (the gather dominates), so this is a direct
// convolution on the vector ALU: one thread = 2 horizontally adjacent output pixels x 16 output
// channels, weights broadcast from LDS (27 x Cout floats), 864 FMAs per thread.  The output is written
// either as float32 NHWC or directly as SPLIT16 records (one 64-byte record per thread and pixel), which
// removes the separate float32 -> SPLIT16 pass in front of the split-precision layers.
// HBM-bound in principle (12 B read + 4*Cout B written per pixel); VALU-bound in practice.
struct __attribute__((packed, aligned(4))) F4u {      // float4 at 4-byte alignment (still one global_load_dwordx4)
    float x, y, z, w;
};

// WCG (Cout <= 64): the channel group is WAVE-uniform (wave w of the block = group w % CG, a lane = a pixel pair), so the 16 weights of
// a tap are a scalar load (s_load_dwordx16 from `w`, scalar-cache hits) and feed the FMAs as SGPR operands, two channels per
// v_pk_fma_f32.  With a lane-dependent group every tap cost four 16-byte LDS reads per lane: 4 KB per wave for 32 FMAs -- the LDS
// pipe, shared by the CU's four SIMDs, was exactly as busy as the vector ALUs, and the layer ran at half its store rate.
template <bool F16, bool OUT_SPLIT, int STRIDE, bool WCG = false>
__global__ void __launch_bounds__(256)
conv_cin3_kernel(const float *__restrict__ x, const float *__restrict__ w /* (27, Cout) */,
                 const float *__restrict__ bias, void *__restrict__ out, int B, int H, int W, int OH, int OW,
                 int Cout, int stride, int pad_t, int pad_l, int act, long long npairs, unsigned *status)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];       // [27][Cout] + bias[Cout], then the stage
    const int CG = Cout >> 4;
    uint4 *stage = reinterpret_cast<uint4 *>(WCG ? wl : wl + 28 * Cout);   // [2 * 256 / CG pixels][4 * CG pieces] (WCG: no weights in LDS)
    if constexpr (!WCG) {
        for (int i = threadIdx.x; i < 27 * Cout; i += 256) wl[i] = w[i];
        for (int i = threadIdx.x; i < Cout; i += 256) wl[27 * Cout + i] = bias ? bias[i] : 0.0f;
        __syncthreads();
    }
    // index math in 32 bits with ONE division per block (the host checks npairs < 2^31): 64-bit div/mod per
    // thread and per stored piece used to double this kernel's VALU instruction count
    const int PW = (OW + 1) >> 1;                                     // pixel pairs per output row
    const int pairs_per_block = 256 / CG;
    const unsigned pair0 = blockIdx.x * (unsigned)pairs_per_block;
    const unsigned row0 = pair0 / (unsigned)PW;                       // flattened (img, oy) row of the block's first pair
    const int x0 = (int)(pair0 - row0 * (unsigned)PW);
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int cg = WCG ? wv % CG : (int)threadIdx.x % CG;
    const int pairl = WCG ? ((int)threadIdx.x & 63) + 64 * (wv / CG) : (int)threadIdx.x / CG;
    const bool live = pair0 + (unsigned)pairl < (unsigned)npairs;
    int px2 = x0 + pairl;
    unsigned rowf = row0;
    while (px2 >= PW) {                                               // at most pairs_per_block / PW + 1 turns
        px2 -= PW;
        ++rowf;
    }
    const int img = (int)(rowf / (unsigned)OH);
    const int oy = (int)(rowf - (unsigned)img * (unsigned)OH);
    const int ox = 2 * px2;
    (void)stride;

    using f2 = __attribute__((ext_vector_type(2))) float;
    f2 acc[2][8];                                                     // channel pairs: one v_pk_fma_f32 = two FMAs (same chains, same bits)
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        if constexpr (WCG) acc[0][n] = bias ? f2{bias[cg * 16 + 2 * n], bias[cg * 16 + 2 * n + 1]} : f2{0.0f, 0.0f};
        else acc[0][n] = f2{wl[27 * Cout + cg * 16 + 2 * n], wl[27 * Cout + cg * 16 + 2 * n + 1]};
        acc[1][n] = acc[0][n];
    }
    if (live) {
        // per filter row: the 2 output pixels read (STRIDE + 3) consecutive input pixels = 12 or 15 contiguous
        // floats; interior threads fetch them as 4 wide loads, border threads element by element with zero fill
        const float *ximg = x + (size_t)img * H * W * 3;
        const int c0 = ox * STRIDE - pad_l;                              // first input column of the window
        // WCG: the three rows' windows are requested together (one memory latency per thread instead of three; the rolled loop's
        // registers are not what limits residency there -- the staging LDS is)
        float win3[WCG ? 3 : 1][16];
        if constexpr (WCG) {
            const int iy0 = oy * STRIDE - pad_t;
            if (iy0 >= 0 && iy0 + 2 < H && c0 >= 0 && c0 * 3 + 16 <= W * 3) {      // ONE decision for the three rows: twelve loads in flight
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const F4u *src = reinterpret_cast<const F4u *>(ximg + ((size_t)(iy0 + r) * W + c0) * 3);
#pragma unroll
                    for (int v4 = 0; v4 < 4; ++v4) {
                        const F4u v = src[v4];
                        win3[r][4 * v4] = v.x; win3[r][4 * v4 + 1] = v.y; win3[r][4 * v4 + 2] = v.z; win3[r][4 * v4 + 3] = v.w;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int iy = iy0 + r;
                    const bool row_ok = iy >= 0 && iy < H;
#pragma unroll
                    for (int j = 0; j < 15; ++j) {
                        const int col = c0 + j / 3;
                        const bool v = row_ok && col >= 0 && col < W && j < 3 * (STRIDE + 3);
                        win3[r][j] = v ? ximg[((size_t)iy * W + col) * 3 + j % 3] : 0.0f;
                    }
                    win3[r][15] = 0.0f;
                }
            }
        }
#pragma unroll(WCG ? 3 : 1)
        for (int r = 0; r < 3; ++r) {                                    // (!WCG) rolled: keeps registers low
            const int iy = oy * STRIDE + r - pad_t;
            const bool row_ok = iy >= 0 && iy < H;
            float win[16];
            if constexpr (WCG) {
#pragma unroll
                for (int j = 0; j < 16; ++j) win[j] = win3[r][j];
            } else if (row_ok && c0 >= 0 && c0 * 3 + 16 <= W * 3) {
                const F4u *src = reinterpret_cast<const F4u *>(ximg + ((size_t)iy * W + c0) * 3);
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4) {
                    const F4u v = src[v4];
                    win[4 * v4] = v.x; win[4 * v4 + 1] = v.y; win[4 * v4 + 2] = v.z; win[4 * v4 + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 15; ++j) {
                    const int col = c0 + j / 3;
                    const bool v = row_ok && col >= 0 && col < W && j < 3 * (STRIDE + 3);
                    win[j] = v ? ximg[((size_t)iy * W + col) * 3 + j % 3] : 0.0f;
                }
                win[15] = 0.0f;
            }
#pragma unroll
            for (int sc = 0; sc < 9; ++sc) {                             // (tap column s, channel c) = (sc / 3, sc % 3)
                const float4 *wk = WCG ? reinterpret_cast<const float4 *>(w + (r * 9 + sc) * Cout + cg * 16)      // (uniform: scalar loads)
                                       : reinterpret_cast<const float4 *>(&wl[(r * 9 + sc) * Cout + cg * 16]);
                const float4 w0 = wk[0], w1 = wk[1], w2 = wk[2], w3 = wk[3];
                const f2 ws[8] = {{w0.x, w0.y}, {w0.z, w0.w}, {w1.x, w1.y}, {w1.z, w1.w},
                                  {w2.x, w2.y}, {w2.z, w2.w}, {w3.x, w3.y}, {w3.z, w3.w}};
                const float in0 = win[sc], in1 = win[sc + 3 * STRIDE];    // pixel p reads column p*STRIDE + s
                const f2 i0 = {in0, in0}, i1 = {in1, in1};
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    acc[0][n] = __builtin_elementwise_fma(i0, ws[n], acc[0][n]);
                    acc[1][n] = __builtin_elementwise_fma(i1, ws[n], acc[1][n]);
                }
            }
        }
    }
    // stage this thread's 2 x 4 pieces, block-local layout [pixel = 2*pairl + p][piece = 4*cg + k]
    // activation as a branch-free clamp (linear / relu / relu6 only; the host rejects sigmoid for this kernel)
    const float act_lo = act == ACT_LINEAR ? -INFINITY : 0.0f;
    const float act_hi = act == ACT_RELU6 ? 6.0f : INFINITY;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float v[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) v[n] = fminf(fmaxf(acc[p][n >> 1][n & 1], act_lo), act_hi);
        // WCG: a wave's lanes are pixel PAIRS, 2 * ppp pieces = a multiple of 512 bytes apart -- every lane of a store on the same
        // four banks; the piece index inside a pixel is XOR-swizzled by the pair index (undone by the block-linear reader below)
        uint4 *dst = stage + (2 * pairl + p) * (4 * CG);
        const int q0 = 4 * cg, sw = WCG ? (pairl & (4 * CG - 1) & 7) : 0;
        if constexpr (OUT_SPLIT) {
            const float lo8[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
            const float hi8[8] = {v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
            dst[(q0 + 0) ^ sw] = split_piece<F16>(lo8, false, status);
            dst[(q0 + 1) ^ sw] = split_piece<F16>(lo8, true);
            dst[(q0 + 2) ^ sw] = split_piece<F16>(hi8, false, status);
            dst[(q0 + 3) ^ sw] = split_piece<F16>(hi8, true);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                dst[(q0 + k) ^ sw] = make_uint4(__float_as_uint(v[4 * k]), __float_as_uint(v[4 * k + 1]),
                                                __float_as_uint(v[4 * k + 2]), __float_as_uint(v[4 * k + 3]));
        }
    }
    __syncthreads();
    // block-linear stores: consecutive lanes write consecutive 16-byte pieces (1 KB per wave instruction);
    // both layouts (SPLIT16 records, float32 NHWC) are 4*CG pieces per pixel in the same order
    const int ppp = 4 * CG;                                           // pieces per pixel
    const int total = 2 * pairs_per_block * ppp;                      // = 2048
    uint4 *gout = reinterpret_cast<uint4 *>(out);
    for (int e = threadIdx.x; e < total; e += 256) {
        const int pl = e / ppp, q = e - pl * ppp;
        if (pair0 + (unsigned)(pl >> 1) >= (unsigned)npairs) continue;
        int gx2 = x0 + (pl >> 1);
        unsigned grow = row0;
        while (gx2 >= PW) {
            gx2 -= PW;
            ++grow;
        }
        const int gox = 2 * gx2 + (pl & 1);
        if (gox >= OW) continue;
        gout[((size_t)grow * OW + gox) * ppp + q] = stage[WCG ? pl * ppp + (q ^ ((pl >> 1) & (ppp - 1) & 7)) : e];
    }
}

// ---- first layer of the float32 graphs on the float32 matrix pipe (round 6): Cin = 3, 3x3, stride 1, 'same', Cout = 64 --------
// VGG16 block1_conv1 in the f32 / f32w precisions.  The vector-ALU kernel above needs 432 packed FMAs + a staged store per thread and
// ran at 0.154 ms (batch 8, 500 x 500); a kernel that ONLY writes the layer's 512 MB takes 0.092 ms with 32 waves per CU and 0.12 ms
// with 8 (scripts/micro/store_rate.hip, store_overlap.hip).  This one: 0.139 ms.
// The 27 taps are ONE K = 28 reduction of v_mfma_f32_16x16x4_f32 (twice the vector ALU's FMA rate): the WEIGHTS are the A operand
// (rows = 16 output channels), a 16-pixel run of an output row is the B operand (columns), so that a lane's four accumulators are four
// CONSECUTIVE channels of one pixel -- a 16-byte store, four lanes complete 64 contiguous bytes, no transpose through LDS.  K index
// e = 9 r + 3 s + c (filter row, filter column, input channel) = the (27, Cout) weight order; for a fixed r the nine (s, c) values of a
// pixel are nine CONSECUTIVE floats of input row y + r - 1 starting at 3 (x - 1).  Bias first, then k ascending: the order of the
// vector-ALU kernel's chain.
// A first form read the B operand straight from global memory (one dword per lane and K step) and the weights per wave: 72 narrow
// load instructions per wave, and the texture-address path alone took 0.076 ms of its 0.166.
// So: PERSISTENT workgroups (the 28 + 16 weight / bias registers are loaded once per workgroup, not per tile), a tile = 8 output rows x
// 64 pixels (a wave = rows wv and wv + 4), its 10 x 66-pixel input patch staged through LDS by 8 coalesced dword loads per thread
// (requested one tile ahead, double-buffered: ONE barrier per tile), the B operand read from there (ds_read_b32, conflict-free: lanes
// 12 bytes apart).  The float32 MFMA executes on the vector ALU's lanes, so every other instruction of the tile loop is MFMA time
// as well: the tile's coordinates advance incrementally (two divisions per tile were ~90 scalar instructions), interior tiles
// have their tile offset in the loads' SCALAR offset (no per-lane address arithmetic), ReLU is one integer max.
// Reference op: Conv2D(64, (3, 3), padding="same", activation="relu") = VGG16 block1_conv1 behind /root/reference/rpn.py:26-33.
constexpr int kC3Rows = 8;                          // output rows per tile
constexpr int kC3PatchRow = 200;                    // floats per staged patch row (198 used).  (The B-operand reads are 2-way bank-conflicted by
                                                    // construction -- four K lanes per pixel on a 3-float pixel stride cannot tile 64 banks; rocprofv3: 45 % of
                                                    // this kernel's LDS cycles, which are < 1 % of its time.  A 201-float row changed nothing.)
constexpr int kC3Stage = ((kC3Rows + 2) * 198 + 255) / 256;            // staging loads per thread (8)
constexpr int kC3PatchFloats = (kC3Stage * 256 / 198 + 1) * kC3PatchRow;   // 10 rows + the row that takes the idle threads' writes
struct C3Tile {
    int img, by, bx;
};
template <bool RELU>
__global__ void __launch_bounds__(256, 2)       // (<= 256 registers: the accumulators stay in the VGPR file, no v_accvgpr copies)
conv_cin3_f32_mfma_kernel(const float *__restrict__ x, const float *__restrict__ w /* (27, 64) */, const float *__restrict__ bias,
                          float *__restrict__ out, int H, int W, int tiles_x, int tiles_y, int n_tiles)
{
    __shared__ float patch[2][kC3PatchFloats];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, kq = lane >> 4;
    constexpr unsigned kOob = 0x80000000u;
    float aw[7][4];                                                      // A: W[channel 16 nb + (lane & 15)][k = 4 kk + kq]
#pragma unroll
    for (int kk = 0; kk < 7; ++kk) {
        const int k = 4 * kk + kq;                                       // (k = 27, the padding: row 26 read, zero kept)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const float v = w[(k < 27 ? k : 26) * 64 + nb * 16 + px];
            aw[kk][nb] = k < 27 ? v : 0.0f;
        }
    }
    f32x4 bv[4];                                                         // the lane's channels 16 nb + 4 kq + (0..3)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) bv[nb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (bias) {                                                          // (the caller's pointer: 4-byte alignment only)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[nb][i] = bias[nb * 16 + 4 * kq + i];
    }
    int rd[7];                                                           // B: patch[row wv + 4 h + r][3 (16 mb + px) + t] of k = 9 r + t, floats
#pragma unroll
    for (int kk = 0; kk < 7; ++kk) {
        const int e = 4 * kk + kq < 27 ? 4 * kk + kq : 26, r = e / 9, t = e - 9 * r;   // (k = 27: any address, its weight is 0 -- but finite data: row 26's)
        rd[kk] = (wv + r) * kC3PatchRow + 3 * px + t;
    }
    // staging: element i = tid + 256 it of the 10 x 198 patch (row r = i / 198, float j of the row: column x0 - 1 + j / 3)
    int st_lds[kC3Stage];
    unsigned st_off[kC3Stage];                                           // byte offset from the patch's first float; out of range for the idle threads
#pragma unroll
    for (int it = 0; it < kC3Stage; ++it) {
        const int i = tid + 256 * it, r = i / 198, j = i - 198 * r;
        st_lds[it] = r * kC3PatchRow + j;
        st_off[it] = r < kC3Rows + 2 ? (unsigned)((r * W * 3 + j) * 4) : kOob;
    }
    const int tiles_xy = tiles_x * tiles_y;
    // tile t = (img, by, bx); a workgroup's tiles are gridDim.x apart: the step as (d_img, d_by, d_bx), added with carries
    const int g = (int)gridDim.x, d_img = g / tiles_xy, d_by = (g - d_img * tiles_xy) / tiles_x, d_bx = g - d_img * tiles_xy - d_by * tiles_x;
    auto advance = [&](C3Tile t) {
        t.bx += d_bx;
        if (t.bx >= tiles_x) {
            t.bx -= tiles_x;
            ++t.by;
        }
        t.by += d_by;
        if (t.by >= tiles_y) {
            t.by -= tiles_y;
            ++t.img;
        }
        t.img += d_img;
        return t;
    };
    float pre[kC3Stage];
    auto request = [&](C3Tile t) {                                       // the tile's patch -> registers (zero outside the image)
        const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x) + (size_t)t.img * H * W * 3, (short)0,
                                                                               H * W * 12, 0x00020000);
        if (t.by >= 1 && t.by * kC3Rows + kC3Rows + 1 <= H && t.bx >= 1 && t.bx * 192 + 195 <= 3 * W) {   // (uniform) all inside the image:
            const int tbase = (((t.by * kC3Rows - 1) * W * 3) + t.bx * 192 - 3) * 4;   // no vector ALU work -- the tile in the SCALAR offset
#pragma unroll
            for (int it = 0; it < kC3Stage; ++it)
                pre[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rs, st_off[it], tbase, 0));
        } else {
#pragma unroll
            for (int it = 0; it < kC3Stage; ++it) {
                const int i = tid + 256 * it, r = i / 198, j = i - 198 * r;
                const int iy = t.by * kC3Rows - 1 + r, fx = t.bx * 192 - 3 + j;
                const bool ok = r < kC3Rows + 2 && iy >= 0 && iy < H && fx >= 0 && fx < 3 * W;
                pre[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rs, ok ? (unsigned)((iy * W * 3 + fx) * 4) : kOob, 0, 0));
            }
        }
    };
    auto deposit = [&](int buf) {
#pragma unroll
        for (int it = 0; it < kC3Stage; ++it) patch[buf][st_lds[it]] = pre[it];
    };
    const unsigned st_out = (unsigned)((px * 64 + 4 * kq) * 4);          // the lane's bytes from the tile row's first output
    int tile = blockIdx.x, buf = 0;
    C3Tile cur;
    cur.img = tile / tiles_xy;
    cur.by = (tile - cur.img * tiles_xy) / tiles_x;
    cur.bx = tile - cur.img * tiles_xy - cur.by * tiles_x;
    if (tile < n_tiles) {
        request(cur);
        deposit(0);
    }
    __syncthreads();
    // The loop body is STRAIGHT-LINE (a row past the image's end computes on zeros and stores out of range; the last tile requests an
    // out-of-range patch): the vector-memory counter is in order, so "the next patch has arrived" is `s_waitcnt vmcnt(32)` -- the 32
    // stores issued behind the request may still be in flight -- and hipcc can only count that far when no branch lies between.  With the
    // stores in a branch the wait became vmcnt(4): every tile waited for its own stores' acknowledgements.
    for (; tile < n_tiles; tile += g, buf ^= 1) {
        const bool more = tile + g < n_tiles;
        C3Tile nxt = advance(cur);
        if (!more) nxt.by = -0x100000;                                   // (every row out of the image: eight loads that fetch nothing)
        const int x0 = cur.bx * 64;
        const bool whole = x0 + 64 <= W;
        const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)cur.img * H * W * 64, (short)0, H * W * 256,
                                                                                0x00020000);
        float pin[4][7];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int kk = 0; kk < 7; ++kk) pin[mb][kk] = patch[buf][rd[kk] + mb * 48];
        request(nxt);                                                    // (in flight across the tile's MFMAs)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int y = cur.by * kC3Rows + wv + 4 * h;
            const bool row = y < H;                                      // (uniform)
            const int tb = (y * W + x0) * 256;                           // (scalar) the tile row's first output, bytes into the image
            if (h == 1) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int kk = 0; kk < 7; ++kk) pin[mb][kk] = patch[buf][4 * kC3PatchRow + rd[kk] + mb * 48];
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                f32x4 acc[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[nb] = bv[nb];
#pragma unroll
                for (int kk = 0; kk < 7; ++kk)
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[kk][nb], pin[mb][kk], acc[nb], 0, 0, 0);
                if constexpr (RELU) {
                    // max(x, 0) on the BITS (v_max_i32: one instruction; fmaxf is a canonicalising v_max v, v, v + the max).  Negative floats
                    // are negative integers -> +0; -0.0 -> +0; a NaN with the sign bit clear stays a NaN (as in the reference's ReLU), with
                    // it set -> 0.
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float f = acc[nb][i];                  // (a VALUE: __builtin_bit_cast of the vector-element lvalue read element 0 four times)
                            acc[nb][i] = __int_as_float(__builtin_elementwise_max(__float_as_int(f), 0));
                        }
                }
                // The tile offset goes into the VECTOR offset (one v_add per 16 pixels), the scalar offset is the constant 0.  Behind a
                // 16-byte store whose scalar offset is a REGISTER hipcc leaves NO wait state before a vector instruction that overwrites the
                // store's data registers (its hazard table knows the hazard for a constant offset only), and gfx950 needs one:
                // `buffer_store_dwordx4 v[18:21], v26, s[12:15], s54 offen` + `v_add_u32 v18, 0x1000, v120` stored the LDS address in
                // element 0 of lanes 12-15 of every 16, differently from run to run (scripts/micro/store_hazard.hip measures it: 0.5 % of
                // the stores with no wait state, none with one; dword stores are not affected).  scripts/isa_store_hazard.py scans the
                // library's assembly for the pattern.
                const unsigned vo = (row && (whole || x0 + 16 * mb + px < W)) ? st_out + (unsigned)(tb + mb * 4096) : kOob;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[nb]), out_rs, vo + (unsigned)(nb * 64), 0, 0);
            }
        }
        deposit(buf ^ 1);                                                // (the other buffer: every wave left it at the last barrier)
        cur = nxt;
        __syncthreads();
    }
}

bool conv_cin3_uses_f32_mfma(int B, int H, int W, int OH, int OW, int Cout, int stride, int pad_t, int pad_l, int act, int out_fmt)
{
    // (laboratory knob 0 = the vector-ALU kernel; same contract)
    static const int f32_mfma = RPN_LAB_KNOB("RPN_CIN3_F32_MFMA", 1);
    return f32_mfma && out_fmt == 0 && Cout == 64 && stride == 1 && pad_t == 1 && pad_l == 1 && OH == H && OW == W &&
           (act == ACT_RELU || act == ACT_LINEAR) && (long long)H * W * 256 < 0x7fffffffll && B > 0 && H > 0 && W > 0 &&
           (long long)((W + 63) / 64) * ((H + 7) / 8) * B < 0x7fffffffll;            // (the tile count: 32-bit in the kernel)
}

// w: (27, Cout) float32 = HWIO flattened (BatchNorm scale already folded); out_fmt: 0 float32 NHWC, 1 SPLIT16
hipError_t launch_conv_cin3(const float *x, const float *w, const float *bias, void *out, int B, int H, int W,
                            int OH, int OW, int Cout, int stride, int pad_t, int pad_l, int act, int out_fmt, bool f16,
                            hipStream_t s)
{
    if (Cout % 16 != 0 || Cout > 256 || 256 % (Cout / 16) != 0 || act == ACT_SIGMOID) return hipErrorInvalidValue;
    if (conv_cin3_uses_f32_mfma(B, H, W, OH, OW, Cout, stride, pad_t, pad_l, act, out_fmt)) {
        const int tiles_x = (W + 63) / 64, tiles_y = (H + kC3Rows - 1) / kC3Rows;
        const long long n_tiles = (long long)tiles_x * tiles_y * B;
        if (n_tiles >= 0x7fffffffll) return hipErrorInvalidValue;
        int dev = 0, n_cus = 0;                                          // persistent: 3 workgroups per CU (<= 168 registers) of the CURRENT device
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            n_cus < 1)
            n_cus = 256;
        const dim3 grid((unsigned)(n_tiles < 3ll * n_cus ? n_tiles : 3ll * n_cus));
        if (act == ACT_RELU)
            hipLaunchKernelGGL(conv_cin3_f32_mfma_kernel<true>, grid, dim3(256), 0, s, x, w, bias, static_cast<float *>(out), H, W,
                               tiles_x, tiles_y, (int)n_tiles);
        else
            hipLaunchKernelGGL(conv_cin3_f32_mfma_kernel<false>, grid, dim3(256), 0, s, x, w, bias, static_cast<float *>(out), H, W,
                               tiles_x, tiles_y, (int)n_tiles);
        return hipGetLastError();
    }
    const long long npairs = (long long)B * OH * ((OW + 1) / 2);
    const long long threads = npairs * (Cout / 16);
    if (threads <= 0) return hipSuccess;
    if (npairs >= 0x7fffffffll || threads / 256 >= 0x7fffffffll) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    const bool wcg = Cout <= 64 && 4 % (Cout / 16) == 0;            // 1, 2 or 4 channel groups: one per wave (conv_cin3_kernel)
    const size_t lds = (wcg ? 0 : (size_t)28 * Cout * sizeof(float)) + (size_t)2048 * 16;   // (weights + bias) + staged records
#define RPN_CIN3(F16_, SPLIT_, STRIDE_)                                                                       \
    do {                                                                                                      \
        if (wcg)                                                                                              \
            hipLaunchKernelGGL((conv_cin3_kernel<F16_, SPLIT_, STRIDE_, true>), dim3(grid), dim3(256), lds, s, x, w, bias, out, B, \
                               H, W, OH, OW, Cout, stride, pad_t, pad_l, act, npairs, range_status());         \
        else                                                                                                  \
            hipLaunchKernelGGL((conv_cin3_kernel<F16_, SPLIT_, STRIDE_, false>), dim3(grid), dim3(256), lds, s, x, w, bias, out, B, \
                               H, W, OH, OW, Cout, stride, pad_t, pad_l, act, npairs, range_status());         \
    } while (0)
    if (stride != 1 && stride != 2) return hipErrorInvalidValue;
    if (out_fmt == 0) {
        if (stride == 1) RPN_CIN3(false, false, 1); else RPN_CIN3(false, false, 2);
    } else if (f16) {
        if (stride == 1) RPN_CIN3(true, true, 1); else RPN_CIN3(true, true, 2);
    } else {
        if (stride == 1) RPN_CIN3(false, true, 1); else RPN_CIN3(false, true, 2);
    }
#undef RPN_CIN3
    return hipGetLastError();
}

// ---- host side -------------------------------------------------------------------------------
// Device word that the kernels launched from this thread flag float16 range violations into (rpn_model_forward sets
// it to the model's status word around its launches; null = no reporting).
static thread_local unsigned *t_range_status = nullptr;
void set_range_status(unsigned *p) { t_range_status = p; }
unsigned *range_status() { return t_range_status; }

static inline unsigned short f32_to_bf16_rne(float f)
{
    unsigned u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float bf16_to_f32(unsigned short h)
{
    const unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int split_weight_shift(const float *hwio, size_t count, bool f16)
{
    if (!f16) return 0;
    float mx = 0.0f;
    for (size_t i = 0; i < count; ++i) {
        const float v = hwio[i] < 0 ? -hwio[i] : hwio[i];
        if (v > mx) mx = v;
    }
    if (!(mx > 0.0f)) return 0;
    int e = 0;
    (void)frexpf(mx, &e);                       // mx = m * 2^e, m in [0.5, 1)
    int s = 12 - e;                             // largest weight lands in [2^11, 2^12): far from fp16 overflow
    if (s < -100) s = -100;                     // negative: weights beyond the float16 range are scaled DOWN (their hi
    if (s > 24) s = 24;                         // halves would be inf), the epilogue's 2^-s restores the magnitude
    return s;
}

// HWIO (3,3,Cin,Cout) float32 (already multiplied by `scale[n]` if given) -> split records
void pack_weights_split_host(const float *hwio, const float *scale, int Cin, int Cout, int cout_pad, bool f16,
                             int shift, unsigned short *dst /* [Cin/16][9][cout_pad][32] */)
{
    const int chunks = Cin / 16;
    const float mul = ldexpf(1.0f, shift);
    memset(dst, 0, (size_t)9 * chunks * cout_pad * 32 * sizeof(unsigned short));
    for (int t = 0; t < 9; ++t)
        for (int c = 0; c < Cin; ++c)
            for (int n = 0; n < Cout; ++n) {
                float v = hwio[((size_t)t * Cin + c) * Cout + n];
                if (scale) v *= scale[n];
                v *= mul;
                unsigned short hi, lo;
                if (f16) {
                    const _Float16 h = (_Float16)v;
                    const _Float16 l = (_Float16)(v - (float)h);
                    memcpy(&hi, &h, 2);
                    memcpy(&lo, &l, 2);
                } else {
                    hi = f32_to_bf16_rne(v);
                    lo = f32_to_bf16_rne(v - bf16_to_f32(hi));
                }
                const int chunk = c >> 4, g = (c >> 3) & 1, k = c & 7;
                unsigned short *rec = dst + (((size_t)chunk * 9 + t) * cout_pad + n) * 32;
                rec[(g * 2 + 0) * 8 + k] = hi;
                rec[(g * 2 + 1) * 8 + k] = lo;
            }
}

// "split32" packing for the 16x16x32-MFMA kernel: [Cin/32][9][cout_pad][8 pieces = hi k0-7, hi k8-15, hi k16-23,
// hi k24-31, lo k0-7, ...] (128 bytes per output channel, tap and 32-channel slice)
void pack_weights_split32_host(const float *hwio, const float *scale, int Cin, int Cout, int cout_pad, bool f16,
                               int shift, unsigned short *dst /* [Cin/32][9][cout_pad][64] */)
{
    const int chunks = Cin / 32;
    const float mul = ldexpf(1.0f, shift);
    memset(dst, 0, (size_t)9 * chunks * cout_pad * 64 * sizeof(unsigned short));
    for (int t = 0; t < 9; ++t)
        for (int c = 0; c < Cin; ++c)
            for (int n = 0; n < Cout; ++n) {
                float v = hwio[((size_t)t * Cin + c) * Cout + n];
                if (scale) v *= scale[n];
                v *= mul;
                unsigned short hi, lo;
                if (f16) {
                    const _Float16 h = (_Float16)v;
                    const _Float16 l = (_Float16)(v - (float)h);
                    memcpy(&hi, &h, 2);
                    memcpy(&lo, &l, 2);
                } else {
                    hi = f32_to_bf16_rne(v);
                    lo = f32_to_bf16_rne(v - bf16_to_f32(hi));
                }
                const int chunk = c >> 5, kgq = (c >> 3) & 3, k = c & 7;
                unsigned short *rec = dst + (((size_t)chunk * 9 + t) * cout_pad + n) * 64;
                const int swz = (n >> 1) & 7;                 // the kernels' LDS image: piece p of channel n at p ^ swz
                rec[(kgq ^ swz) * 8 + k] = hi;
                rec[((4 + kgq) ^ swz) * 8 + k] = lo;
            }
}

static int grid_cap(long long items)
{
    long long g = (items + 255) / 256;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;
    return (int)g;
}

hipError_t launch_f32_to_split(const float *x, long long npix, int C, bool f16, void *out, hipStream_t s)
{
    if (C % 16 != 0) return hipErrorInvalidValue;
    const long long pieces = npix * (C / 16) * 4;
    if (pieces == 0) return hipSuccess;
    if (f16) hipLaunchKernelGGL(f32_to_split_kernel<true>, dim3(grid_cap(pieces)), dim3(256), 0, s, x, pieces, (uint4 *)out, range_status());
    else hipLaunchKernelGGL(f32_to_split_kernel<false>, dim3(grid_cap(pieces)), dim3(256), 0, s, x, pieces, (uint4 *)out, range_status());
    return hipGetLastError();
}

hipError_t launch_split_to_f32(const void *x, long long npix, int C, bool f16, float *out, hipStream_t s)
{
    if (C % 16 != 0) return hipErrorInvalidValue;
    const long long halves = npix * (C / 8);
    if (halves == 0) return hipSuccess;
    if (f16) hipLaunchKernelGGL(split_to_f32_kernel<true>, dim3(grid_cap(halves)), dim3(256), 0, s, (const uint4 *)x, halves, out);
    else hipLaunchKernelGGL(split_to_f32_kernel<false>, dim3(grid_cap(halves)), dim3(256), 0, s, (const uint4 *)x, halves, out);
    return hipGetLastError();
}

hipError_t launch_maxpool_split(const void *x, int B, int H, int W, int C, bool f16, void *out, hipStream_t s)
{
    if (C % 16 != 0) return hipErrorInvalidValue;
    const int OH = H / 2, OW = W / 2, G = C / 8;
    const long long total = (long long)B * OH * OW * G;
    if (total == 0) return hipSuccess;
    if (f16) hipLaunchKernelGGL(maxpool_split_kernel<true>, dim3(grid_cap(total)), dim3(256), 0, s, (const uint4 *)x, H, W, G, OH, OW, total, (uint4 *)out);
    else hipLaunchKernelGGL(maxpool_split_kernel<false>, dim3(grid_cap(total)), dim3(256), 0, s, (const uint4 *)x, H, W, G, OH, OW, total, (uint4 *)out);
    return hipGetLastError();
}

template <int TH, int WN, int BBUF, bool F16, bool POOL, int NW = 4>
static hipError_t launch_split_variant(const SplitConvArgs &a, hipStream_t s)
{
    constexpr int BN = (NW == 8 ? WN * 64 : WN * 64);
    const int tiles_x = (a.W + TWS - 1) / TWS, tiles_y = (a.H + TH - 1) / TH;
    const int n_tiles = (a.Cout + BN - 1) / BN;
    // grid padded to 8 x ceil(n_tiles / XN) x ceil(m_tiles / XM): see the XCD-aware tile order in the kernel
    const long long m_tiles = (long long)tiles_x * tiles_y * a.B;
    const int XN = n_tiles >= 8 ? 8 : (n_tiles >= 4 ? 4 : (n_tiles >= 2 ? 2 : 1)), XM = 8 / XN;
    const long long nblocks = 8ll * ((n_tiles + XN - 1) / XN) * ((m_tiles + XM - 1) / XM);
    if (m_tiles <= 0 || nblocks > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv3x3_split_kernel<TH, WN, BBUF, NW, F16, POOL>), dim3((unsigned)nblocks), dim3(64 * NW), 0,
                       s, a, tiles_x, tiles_y, n_tiles);
    return hipGetLastError();
}

// laboratory knobs (-DRPN_LAB builds only; speed only, never results): RPN_SPLIT_TILE = 82 | 81 | 42 | 41 forces a tile shape,
// RPN_SPLIT_BBUF = 1 | 2 the number of weight buffers

template <bool F16, bool POOL>
static hipError_t launch_split_tiles(const SplitConvArgs &a, hipStream_t s)
{
    // tile choice: 64-wide N tiles for Cout <= 64; shorter / narrower tiles when the grid would not put
    // two workgroups on each of the 256 CUs
    static const int force_tile = RPN_LAB_KNOB("RPN_SPLIT_TILE", 0), force_bbuf = RPN_LAB_KNOB("RPN_SPLIT_BBUF", 0);
    const long long mt8 = (long long)((a.W + 31) / 32) * ((a.H + 7) / 8) * a.B;
    const long long mt4 = (long long)((a.W + 31) / 32) * ((a.H + 3) / 4) * a.B;
    const int nt128 = (a.Cout + 127) / 128;
    int tile;
    if (a.Cout <= 64) tile = 81;
    else if (mt8 * nt128 >= 512) tile = 82;
    else if (mt4 * nt128 >= 512 || POOL) tile = 42;
    else tile = 41;
    if (force_tile && a.Cout > 64 && !(POOL && force_tile == 41)) tile = force_tile;
    // default buffers: two wherever LDS still admits two workgroups per CU (<= 80 KB)
    int bbuf = (tile == 82) ? 1 : 2;
    if (force_bbuf) bbuf = force_bbuf;
    if (tile == 82 && bbuf == 2) bbuf = 1;                   // 92 KB: one workgroup per CU, not offered
    static const int force_waves = RPN_LAB_KNOB("RPN_SPLIT_WAVES", 0);        // 8: 512-thread workgroups on the 8x32 x 128 tile
    if (tile == 82 && force_waves == 8) return launch_split_variant<8, 2, 1, F16, POOL, 8>(a, s);
    // experiment (RPN_SPLIT_SMALL=2): 128 px x 128 ch tiles shared by 8 waves for small feature maps.  Measured on
    // 31x31x512, batch 8: 0.137 ms vs 0.126 ms for the default 4-wave 128 x 64 tiles, so it stays off.
    static const int small_mode = RPN_LAB_KNOB("RPN_SPLIT_SMALL", 0);
    if constexpr (!POOL) {
        if (tile == 41 && small_mode == 2 && mt4 * nt128 >= 256) return launch_split_variant<4, 2, 2, F16, false, 8>(a, s);
    }
    switch (tile * 10 + bbuf) {
        case 821: return launch_split_variant<8, 2, 1, F16, POOL>(a, s);
        case 811: return launch_split_variant<8, 1, 1, F16, POOL>(a, s);
        case 812: return launch_split_variant<8, 1, 2, F16, POOL>(a, s);
        case 421: return launch_split_variant<4, 2, 1, F16, POOL>(a, s);
        case 422: return launch_split_variant<4, 2, 2, F16, POOL>(a, s);
        case 411: if constexpr (!POOL) return launch_split_variant<4, 1, 1, F16, false>(a, s); else break;
        case 412: if constexpr (!POOL) return launch_split_variant<4, 1, 2, F16, false>(a, s); else break;
        default: break;
    }
    return hipErrorInvalidValue;
}

// 16x16x32-MFMA kernel (weights in "split32" packing).  Needs Cin % 32 == 0 and cout_pad % 128 == 0.
// Tile-queue counters of the persistent kernel's dynamic schedule: 17 unsigned per stream (launches on one stream are
// serialised and every launch leaves them zero), allocated and zeroed at the stream's first launch.  Null = static tile
// schedule: the default.  RPN_S16_DYN=1 turns the dynamic schedule on (not under stream capture at a stream's first
// launch, where nothing may be allocated).  Measured on VGG16, batch 8: alone on the GPU the dynamic schedule costs 2 %
// (2760 vs 2814 images/s: a device-scope atomic takes microseconds on this multi-XCD part, and the in-order vmcnt makes
// the drawing wave wait for it one interval later); it pays when another stream's kernel holds CUs during a persistent
// layer (block1_conv2 on this kernel beside the overlapped NMS: 2800 vs 2737 images/s with the static schedule).
static unsigned *sched_counters(hipStream_t s)
{
    static const int dyn = RPN_KNOB("RPN_S16_DYN", 0);
    if (!dyn) return nullptr;
    static std::mutex mu;
    static std::unordered_map<hipStream_t, unsigned *> pool;
    std::lock_guard<std::mutex> lock(mu);
    auto it = pool.find(s);
    if (it != pool.end()) return it->second;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
    unsigned *p = nullptr;
    if (hipMalloc(&p, 128) != hipSuccess || hipMemset(p, 0, 128) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    pool[s] = p;
    return p;
}

const char *conv3x3_split16_variant(int B, int H, int W, int Cin, int Cout, int cout_pad, bool pool)
{
    if (Cin % 32 != 0 || Cout % 16 != 0 || cout_pad % 64 != 0 || cout_pad < Cout) return nullptr;
    static const int dma_mode = RPN_LAB_KNOB("RPN_S16_DMA", 1);
    const long long big_blocks = (long long)((W + 31) / 32) * ((H + 7) / 8) * B * ((Cout + 127) / 128);
    const bool dma_ok = dma_mode && Cin % 64 == 0 && (long long)B * H * W * Cin * 4 < 0x7fffffffll &&
                        (long long)(Cin / 32) * 9 * cout_pad * 128 < 0x7fffffffll;
    // a grid of at most half the CUs (batch-1 feature maps: the layer's time is ONE tile's walk through K): the 4 x 32 x 64
    // register-staged tiles give twice the workgroups (measured at 32 x 32 x 576 -> 512, batch 1: 0.067 vs 0.076 ms)
    const long long blocks64 = (long long)((W + 31) / 32) * ((H + 7) / 8) * B * ((Cout + 63) / 64);
    if (dma_ok && !pool && cout_pad % 128 == 0 && blocks64 <= 128) return "reg,64";
    if (dma_ok) return (cout_pad % 128 != 0 || big_blocks < 256) ? "dma,64" : "dma,128";
    if (cout_pad % 128 != 0) return nullptr;
    return (big_blocks < 256 && !pool) ? "reg,64" : "reg,128";
}

hipError_t launch_conv3x3_split16(const void *x, const void *w, const float *bias, void *out, int B, int H, int W,
                                  int Cin, int Cout, int cout_pad, float out_scale, int act, bool out_f32, bool f16,
                                  bool pool, hipStream_t s, bool ktree)
{
    if (Cin % 32 != 0 || Cout % 16 != 0 || cout_pad % 64 != 0 || act == ACT_SIGMOID) return hipErrorInvalidValue;
    if (ktree && (pool || !conv3x3_split16_ktree_ok(B, H, W, Cin, Cout, cout_pad))) return hipErrorInvalidValue;
    SplitConvArgs a{};
    a.status = range_status();
    a.x = (const uint4 *)x; a.w = (const uint4 *)w; a.bias = bias; a.out = out;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.cout_pad = cout_pad;
    a.out_scale = out_scale; a.act = act; a.out_f32 = out_f32 ? 1 : 0;
    const char *variant = ktree ? "dma,64" : conv3x3_split16_variant(B, H, W, Cin, Cout, cout_pad, pool);
    if (!variant) return hipErrorInvalidValue;
    a.ktree = ktree ? 1 : 0;
    const long long big_blocks = (long long)((W + 31) / 32) * ((H + 7) / 8) * B * ((Cout + 127) / 128);
    if (variant[0] == 'd') {
        // Persistent LDS-DMA kernel: 8 x 32 px tiles, 128 channels wide -- or 64 wide where that is all there is
        // (Cout <= 64) or where 128-wide tiles would leave CUs without a tile (the 31 x 31 layers).  One workgroup per
        // CU (8 per XCD label at least), each walks the slots of its XCD label.
        const int BN = variant[4] == '6' ? 64 : 128;
        const int tiles_x = (W + TWS - 1) / TWS, tiles_y = (H + 7) / 8;
        const int n_tiles = (Cout + BN - 1) / BN;
        const long long m_tiles = (long long)tiles_x * tiles_y * B;
        const int XN = n_tiles >= 8 ? 8 : (n_tiles >= 4 ? 4 : (n_tiles >= 2 ? 2 : 1)), XM = 8 / XN;
        const long long slots = (long long)((n_tiles + XN - 1) / XN) * ((m_tiles + XM - 1) / XM);
        if (m_tiles <= 0 || slots > 0x0fffffffll) return hipErrorInvalidValue;
        static const int n_cus = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
                n = 256;
            return n;
        }();
        const unsigned pgrid = 8u * (unsigned)(slots < n_cus / 8 ? slots : n_cus / 8);
        a.sched = sched_counters(s);
#define RPN_L16D(F16_, POOL_, BN_)                                                                                   \
    hipLaunchKernelGGL((conv3x3_split16_dma_kernel<F16_, POOL_, BN_>), dim3(pgrid), dim3(512), 0, s, a, tiles_x, tiles_y, \
                       n_tiles)
#define RPN_L16D_BN(F16_, POOL_)                                                                                     \
    if (BN == 128) RPN_L16D(F16_, POOL_, 128); else RPN_L16D(F16_, POOL_, 64)
#ifdef RPN_LAB
        // 128-wide tiles: the one-wave-per-SIMD kernel whose epilogue runs under the next tile's taps (static schedule only;
        // slower than this one: laboratory builds only, RPN_S16_W4=1)
        static const int w4 = RPN_LAB_KNOB("RPN_S16_W4", 0);
        const long long out_bytes4 = (long long)B * (pool ? H >> 1 : H) * (pool ? W >> 1 : W) * Cout * 4;
        // (w4 = 1: every 128-wide layer; w4 > 1: only the layers with Cin == w4 -- round 5: block2_conv1 alone, K = 576, where the
        // epilogue is 15 % of a tile's time)
        if (BN == 128 && (w4 == 1 || w4 == Cin) && !a.sched && out_bytes4 < 0x7fffffffll) {
#define RPN_L16D4(F16_, POOL_)                                                                                       \
    hipLaunchKernelGGL((conv3x3_split16_dma4_kernel<F16_, POOL_>), dim3(pgrid), dim3(256), 0, s, a, tiles_x, tiles_y, n_tiles)
            if (f16) { if (pool) { RPN_L16D4(true, true); } else { RPN_L16D4(true, false); } }
            else { if (pool) { RPN_L16D4(false, true); } else { RPN_L16D4(false, false); } }
#undef RPN_L16D4
            return hipGetLastError();
        }
#endif
        if (ktree) {                        // the whole K tree inside the tile loop (SplitConvArgs::ktree)
            if (f16) hipLaunchKernelGGL((conv3x3_split16_dma_kernel<true, false, 64, true>), dim3(pgrid), dim3(512), 0, s, a, tiles_x, tiles_y, n_tiles);
            else hipLaunchKernelGGL((conv3x3_split16_dma_kernel<false, false, 64, true>), dim3(pgrid), dim3(512), 0, s, a, tiles_x, tiles_y, n_tiles);
        } else if (f16) { if (pool) { RPN_L16D_BN(true, true); } else { RPN_L16D_BN(true, false); } }
        else { if (pool) { RPN_L16D_BN(false, true); } else { RPN_L16D_BN(false, false); } }
#undef RPN_L16D_BN
#undef RPN_L16D
        return hipGetLastError();
    }
    // Register-staged kernels (an odd number of 32-channel slices, or RPN_S16_DMA=0).  Large grids: 8 x 32 px x 128 ch
    // tiles, one 8-wave workgroup per CU; small feature maps (no pooling there): 4 x 32 px x 64 ch tiles, 4 waves, two
    // workgroups per CU.
    const bool small = big_blocks < 256 && !pool;
    const int TH = small ? 4 : 8, BN = small ? 64 : 128;
    const int tiles_x = (W + TWS - 1) / TWS, tiles_y = (H + TH - 1) / TH;
    const int n_tiles = (Cout + BN - 1) / BN;
    const long long m_tiles = (long long)tiles_x * tiles_y * B;
    const int XN = n_tiles >= 8 ? 8 : (n_tiles >= 4 ? 4 : (n_tiles >= 2 ? 2 : 1)), XM = 8 / XN;
    const long long nblocks = 8ll * ((n_tiles + XN - 1) / XN) * ((m_tiles + XM - 1) / XM);
    if (m_tiles <= 0 || nblocks > 0x7fffffffll) return hipErrorInvalidValue;
#define RPN_L16(TH_, WN_, NW_, F16_, POOL_)                                                                          \
    hipLaunchKernelGGL((conv3x3_split16_kernel<TH_, WN_, NW_, F16_, POOL_>), dim3((unsigned)nblocks), dim3(64 * NW_), 0, s, \
                       a, tiles_x, tiles_y, n_tiles)
    if (small) {
        if (f16) RPN_L16(4, 1, 4, true, false); else RPN_L16(4, 1, 4, false, false);
    } else if (f16) {
        if (pool) RPN_L16(8, 2, 8, true, true); else RPN_L16(8, 2, 8, true, false);
    } else {
        if (pool) RPN_L16(8, 2, 8, false, true); else RPN_L16(8, 2, 8, false, false);
    }
#undef RPN_L16
    return hipGetLastError();
}

// Split-K factor for a 3x3 layer whose only consumer can add partial-sum slabs (rpn_conv -> the RPN head): grids of at
// most a quarter / half of the CUs (batch-1 feature maps) are cut 4 / 2 ways along K so that the layer's time is no longer
// one workgroup's walk through all of K.  1: no split.
// Whether a 3x3 layer feeding the slab-adding head runs as a K TREE (SplitConvArgs::ktree) -- at EVERY batch size up to
// B, so that the split factor may follow the grid size without changing bits: needs the persistent 64-wide kernel for the
// unsplit case (and its 2 GiB / even-slice limits) and at least one slice pair per leaf.  RPN_KSPLIT=0: one accumulation
// chain at every batch size (the layer's time at batch 1 is then one workgroup's walk through all of K).
bool conv3x3_split16_ktree_ok(int B, int H, int W, int Cin, int Cout, int cout_pad)
{
    static const int on = RPN_KNOB("RPN_KSPLIT", 1);
    return on && Cin % 64 == 0 && Cin >= 256 && Cout % 16 == 0 && cout_pad % 128 == 0 && cout_pad >= Cout &&
           (long long)B * H * W * Cin * 4 < 0x7fffffffll && (long long)(Cin / 32) * 9 * cout_pad * 128 < 0x7fffffffll;
}

bool conv3x3_split16_ksplit_dma(int B, int H, int W, int Cin, int Cout, int cout_pad)
{
    static const int on = RPN_LAB_KNOB("RPN_KSPLIT_DMA", 1);
    const char *v = conv3x3_split16_variant(B, H, W, Cin, Cout, cout_pad, false);
    if (!on || !v || strcmp(v, "reg,64") != 0) return false;           // (the small-grid case of a DMA-capable layer)
    const long long blocks64 = (long long)((W + 31) / 32) * ((H + 7) / 8) * B * ((Cout + 63) / 64);
    const int chunks = Cin / 32, mid = ktree_cut(chunks, 2);
    // the launcher's one-round grid: 2 x 8 x slots workgroups with slots <= 16 (n_tiles / m_tiles rounded up to the XCD split --
    // the same arithmetic as launch_conv3x3_split16_ksplit, so the predicate never says yes to a grid the launcher refuses)
    const int n_tiles = (Cout + 63) / 64;
    const long long m_tiles = (long long)((W + 31) / 32) * ((H + 7) / 8) * B;
    const int XN = n_tiles >= 8 ? 8 : (n_tiles >= 4 ? 4 : (n_tiles >= 2 ? 2 : 1)), XM = 8 / XN;
    const long long slots = (long long)((n_tiles + XN - 1) / XN) * ((m_tiles + XM - 1) / XM);
    return blocks64 > 64 && blocks64 <= 128 && slots <= 16 && mid >= 2 && (chunks - mid) >= 2 && (chunks - mid) % 2 == 0;
}

int conv3x3_split16_ksplit(int B, int H, int W, int Cin, int Cout, int cout_pad)
{
    if (conv3x3_split16_ksplit_dma(B, H, W, Cin, Cout, cout_pad)) return 2;
    // (a K-tree layer only -- the caller checks conv3x3_split16_ktree_ok at its largest batch)  MobileNetV2 500 x 500, one
    // image: 0.292 -> 0.259 ms per step; 1024 x 1024: 0.395 -> 0.377.
    const char *v = conv3x3_split16_variant(B, H, W, Cin, Cout, cout_pad, false);
    if (!v || strcmp(v, "reg,64") != 0) return 1;
    const long long blocks = (long long)((W + 31) / 32) * ((H + 3) / 4) * B * ((Cout + 63) / 64);
    const int chunks = Cin / 32;
    (void)chunks;
    return blocks <= 128 ? 4 : (blocks <= 256 ? 2 : 1);           // (2 workgroups of this kernel fit a CU: 512 slots)
}

// The split-K form of launch_conv3x3_split16 (4 x 32 x 64 register-staged tiles): out = ksplit float32 NHWC slabs of
// (B,H,W,Cout), slab_floats apart, holding RAW partial sums (no bias, no activation).
hipError_t launch_conv3x3_split16_ksplit(const void *x, const void *w, float *out, long long slab_floats, int B, int H, int W,
                                         int Cin, int Cout, int cout_pad, float out_scale, bool f16, int ksplit, hipStream_t s)
{
    if ((ksplit != 2 && ksplit != 4) || !conv3x3_split16_ktree_ok(B, H, W, Cin, Cout, cout_pad)) return hipErrorInvalidValue;
    SplitConvArgs a{};
    a.status = nullptr;
    a.ktree = 1;
    a.x = (const uint4 *)x; a.w = (const uint4 *)w; a.bias = nullptr; a.out = out;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.cout_pad = cout_pad;
    a.out_scale = out_scale; a.act = ACT_LINEAR; a.out_f32 = 1; a.slab_floats = slab_floats;
    if (ksplit == 2 && conv3x3_split16_ksplit_dma(B, H, W, Cin, Cout, cout_pad)) {
        // the persistent LDS-DMA kernel, two workgroups per 8 x 32 x 64 tile (static schedule: the two halves of a tile must not
        // share a tile queue); every workgroup has exactly one tile on these grids
        const int tiles_x = (W + TWS - 1) / TWS, tiles_y = (H + 7) / 8, n_tiles = (Cout + 63) / 64;
        const long long m_tiles = (long long)tiles_x * tiles_y * B;
        const int XN = n_tiles >= 8 ? 8 : (n_tiles >= 4 ? 4 : (n_tiles >= 2 ? 2 : 1)), XM = 8 / XN;
        const long long slots = (long long)((n_tiles + XN - 1) / XN) * ((m_tiles + XM - 1) / XM);
        if (m_tiles <= 0 || slots > 16) return hipErrorInvalidValue;      // 2 x 8 x slots workgroups: one tile each, one round
        const unsigned pgrid = 8u * (unsigned)slots;
        a.sched = nullptr;
        if (f16) hipLaunchKernelGGL((conv3x3_split16_dma_kernel<true, false, 64, true>), dim3(pgrid, 2), dim3(512), 0, s, a, tiles_x, tiles_y, n_tiles);
        else hipLaunchKernelGGL((conv3x3_split16_dma_kernel<false, false, 64, true>), dim3(pgrid, 2), dim3(512), 0, s, a, tiles_x, tiles_y, n_tiles);
        return hipGetLastError();
    }
    const int tiles_x = (W + TWS - 1) / TWS, tiles_y = (H + 3) / 4;
    const int n_tiles = (Cout + 63) / 64;
    const long long m_tiles = (long long)tiles_x * tiles_y * B;
    const int XN = n_tiles >= 8 ? 8 : (n_tiles >= 4 ? 4 : (n_tiles >= 2 ? 2 : 1)), XM = 8 / XN;
    const long long nblocks = 8ll * ((n_tiles + XN - 1) / XN) * ((m_tiles + XM - 1) / XM);
    if (m_tiles <= 0 || nblocks > 0x7fffffffll) return hipErrorInvalidValue;
    const dim3 grid((unsigned)nblocks, (unsigned)ksplit);
    if (f16) hipLaunchKernelGGL((conv3x3_split16_kernel<4, 1, 4, true, false>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, n_tiles);
    else hipLaunchKernelGGL((conv3x3_split16_kernel<4, 1, 4, false, false>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, n_tiles);
    return hipGetLastError();
}

// 3x3 stride-1 'same' conv on SPLIT16 input (optionally followed by a fused 2x2 'valid' max-pool).
// x: SPLIT16 (B,H,W,Cin), w: split records, out: SPLIT16 or f32 NHWC of (B,H,W,Cout) or, pooled, (B,H/2,W/2,Cout).
hipError_t launch_conv3x3_split(const void *x, const void *w, const float *bias, void *out, int B, int H, int W,
                                int Cin, int Cout, int cout_pad, float out_scale, int act, bool out_f32, bool f16,
                                bool pool, hipStream_t s, bool ktree)
{
    if (ktree) return hipErrorInvalidValue;           // (the parameter only keeps the two launchers' signatures equal)
    if (Cin % 16 != 0 || Cout % 16 != 0 || cout_pad % 64 != 0 || act == ACT_SIGMOID) return hipErrorInvalidValue;
    SplitConvArgs a{};
    a.status = range_status();
    a.x = (const uint4 *)x; a.w = (const uint4 *)w; a.bias = bias; a.out = out;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.cout_pad = cout_pad;
    a.out_scale = out_scale; a.act = act; a.out_f32 = out_f32 ? 1 : 0;
    if (pool) return f16 ? launch_split_tiles<true, true>(a, s) : launch_split_tiles<false, true>(a, s);
    return f16 ? launch_split_tiles<true, false>(a, s) : launch_split_tiles<false, false>(a, s);
}

// VGG16 block 1 in one launch: block1_conv1 (3 -> 64) + ReLU -> block1_conv2 (64 -> 64) + ReLU -> 2x2 max-pool
// (models/rpn_vgg16.py:16, keras-applications VGG16).  img: float32 NHWC (B,H,W,3); w1: pack_weights_cin3_mfma_host
// records of block1_conv1; w2: pack_weights_split_host records of block1_conv2; out: SPLIT16 (B,H/2,W/2,64).
hipError_t launch_vgg_block1(const float *img, const void *w1, const float *b1, float scale1, const void *w2,
                             const float *b2, float scale2, void *out, int B, int H, int W, bool f16, hipStream_t s)
{
    if (B < 1 || H < 2 || W < 2) return hipErrorInvalidValue;
    SplitConvArgs a{};
    a.status = range_status();
    a.x = nullptr; a.w = (const uint4 *)w2; a.bias = b2; a.out = out;
    a.B = B; a.H = H; a.W = W; a.Cin = 64; a.Cout = 64; a.cout_pad = 64;
    a.out_scale = scale2; a.act = ACT_RELU; a.out_f32 = 0;
    a.img = img; a.w1 = (const uint4 *)w1; a.b1 = b1; a.scale1 = scale1;
    const int tiles_x = (W + TWS - 1) / TWS, tiles_y = (H + 7) / 8;
    const long long m_tiles = (long long)tiles_x * tiles_y * B;
    const long long nblocks = 8ll * ((m_tiles + 7) / 8);           // n_tiles = 1: XN = 1, XM = 8 (the kernel's tile order)
    if (nblocks > 0x7fffffffll) return hipErrorInvalidValue;
    if (f16)
        hipLaunchKernelGGL((conv3x3_split_kernel<8, 1, 2, 4, true, true, true>), dim3((unsigned)nblocks), dim3(256), 0, s, a,
                           tiles_x, tiles_y, 1);
    else
        hipLaunchKernelGGL((conv3x3_split_kernel<8, 1, 2, 4, false, true, true>), dim3((unsigned)nblocks), dim3(256), 0, s, a,
                           tiles_x, tiles_y, 1);
    return hipGetLastError();
}

}  // namespace rpn
