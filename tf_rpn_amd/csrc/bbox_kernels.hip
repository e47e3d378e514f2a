// bbox_kernels.hip -- anchor grid, delta decode / encode and the pairwise IoU map for gfx950.
//
// All four are HBM-streaming kernels (no reuse, no MFMA): one 16-byte vector access per
// lane per operand, grid capped at ~2048 workgroups with a grid-stride loop.
//   anchors : 16*A bytes written once per process            (utils/bbox_utils.py:23-46)
//   decode  : 16 B read (deltas) + 16 B written per box, anchors are batch-shared and stay
//             in L2                                           (utils/bbox_utils.py:72-96)
//   iou map : 4*B*A*G bytes written, inputs negligible        (utils/bbox_utils.py:126-150)
// Compiled with -ffp-contract=off (see bbox_core.h).
#include "bbox_core.h"
#include "rpn_common.h"

#include <cmath>
#include <cstdlib>

namespace rpn {

constexpr int kThreads = 256;
constexpr int kMaxGrid = 2048;      // 256 CUs x 8 workgroups
constexpr int kMaxBase = 64;        // anchors per cell supported by the by-value table

struct BaseAnchors {
    float v[kMaxBase * 4];
};

// generate_anchors: flat index (y*F + x)*K + k; grid coords are computed in float64 and
// cast to float32 exactly like `tf.range(0,F) / F + stride / 2` (utils/bbox_utils.py:36).
__global__ void __launch_bounds__(kThreads)
anchors_kernel(BaseAnchors base, int F, int K, double half_stride, float *__restrict__ out, int A)
{
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < A; i += gridDim.x * kThreads) {
        const int k = i % K;
        const int cell = i / K;
        const int x = cell % F;
        const int y = cell / F;
        const float gy = (float)((double)y / (double)F + half_stride);
        const float gx = (float)((double)x / (double)F + half_stride);
        Box a;
        a.y1 = clip01(base.v[4 * k + 0] + gy);      // :44 add, :46 clip
        a.x1 = clip01(base.v[4 * k + 1] + gx);
        a.y2 = clip01(base.v[4 * k + 2] + gy);
        a.x2 = clip01(base.v[4 * k + 3] + gx);
        store_box(out + 4 * (size_t)i, a);
    }
}

struct Variances {
    float v[4];
    int enabled;
};

// grid.y = image; 32-bit indices inside an image (64-bit div/mod per element costs more than the box math)
__global__ void __launch_bounds__(kThreads)
decode_kernel(const float *__restrict__ anchors, int anchors_batched, const float *__restrict__ deltas,
              Variances var, int A, float *__restrict__ out)
{
    const size_t img_off = (size_t)blockIdx.y * A;
    for (int a = blockIdx.x * kThreads + threadIdx.x; a < A; a += gridDim.x * kThreads) {
        const size_t i = img_off + a;
        const Box an = load_box(anchors + 4 * (anchors_batched ? i : (size_t)a));
        const float4 d = *reinterpret_cast<const float4 *>(deltas + 4 * i);
        float dy = d.x, dx = d.y, dh = d.z, dw = d.w;
        if (var.enabled) {                      // predictor.py:55
            dy = dy * var.v[0];
            dx = dx * var.v[1];
            dh = dh * var.v[2];
            dw = dw * var.v[3];
        }
        store_box(out + 4 * i, decode_box(an, dy, dx, dh, dw));
    }
}

__global__ void __launch_bounds__(kThreads)
encode_kernel(const float *__restrict__ bboxes, int bboxes_batched, const float *__restrict__ gt, int A,
              float *__restrict__ out)
{
    const size_t img_off = (size_t)blockIdx.y * A;
    for (int a = blockIdx.x * kThreads + threadIdx.x; a < A; a += gridDim.x * kThreads) {
        const size_t i = img_off + a;
        const float4 d = encode_box(load_box(bboxes + 4 * (bboxes_batched ? i : (size_t)a)), load_box(gt + 4 * i));
        *reinterpret_cast<float4 *>(out + 4 * i) = d;
    }
}

// IoU map: grid.y = image; each lane produces 4 consecutive floats of that image's (A,G) slab and stores them
// with one 16-byte store (the slab base is 16-byte aligned when A*G % 4 == 0; otherwise scalar stores).  All
// index math is 32-bit with one division per lane.  The (anchor, gt) operands are a few hundred KB and are
// served by L1/L2; the kernel is bound by the 4*B*A*G output bytes.
__global__ void __launch_bounds__(kThreads)
iou_map_kernel(const float *__restrict__ bboxes, int bboxes_batched, int A, const float *__restrict__ gt,
               int G, int vec_ok, float *__restrict__ out)
{
    const int b = blockIdx.y;
    const int per_img = A * G;
    const float *__restrict__ gtb = gt + 4 * (size_t)b * G;
    const float *__restrict__ bbb = bboxes + (bboxes_batched ? 4 * (size_t)b * A : 0);
    float *__restrict__ outb = out + (size_t)b * per_img;
    const int nvec = (per_img + 3) >> 2;
    for (int v = blockIdx.x * kThreads + threadIdx.x; v < nvec; v += gridDim.x * kThreads) {
        const int e0 = v << 2;
        int a = e0 / G;
        int g = e0 - a * G;
        Box bb = load_box(bbb + 4 * (size_t)a);
        float barea = box_area_plain(bb);                       // :139
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (e0 + j < per_img) {
                const Box gg = load_box(gtb + 4 * g);
                r[j] = iou_map_pair(bb, barea, gg, box_area_plain(gg));
                if (++g == G) {
                    g = 0;
                    ++a;
                    if (j < 3 && e0 + j + 1 < per_img) {
                        bb = load_box(bbb + 4 * (size_t)a);
                        barea = box_area_plain(bb);
                    }
                }
            } else {
                r[j] = 0.0f;
            }
        }
        if (vec_ok && e0 + 3 < per_img) {
            *reinterpret_cast<float4 *>(outb + e0) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
            for (int j = 0; j < 4 && e0 + j < per_img; ++j) outb[e0 + j] = r[j];
        }
    }
}

// IoU map, chunked form (the one that runs at BASELINE sizes).  The (B, A, G) output is one flat array; a workgroup
// owns one 16-byte-aligned run of kIouChunk floats of it (image slabs of A*G floats need not be 16-byte aligned --
// VGG16's 8649 * 42 is not a multiple of 4 -- so a run may straddle two images; it never spans three because
// kIouChunk <= A*G is required).  The gt boxes and areas of the (up to) two images are staged in LDS once per
// workgroup; a lane produces 4 consecutive floats per iteration (one 16-byte store, consecutive lanes -> consecutive
// 16 bytes) and walks its (image, anchor, gt) position incrementally: two integer divisions per lane per launch
// instead of one per vector, no gt-area recomputation, no data-dependent reload branch (both candidate anchor rows of
// a vector are loaded -- L1 hits -- and selected per float).  Operation order per pair is iou_map_pair's
// (utils/bbox_utils.py:138-150): bit-exact with the generic kernel.  HBM-write-bound: 4*B*A*G bytes;
// NT = nontemporal stores (the map is never re-read by this kernel).
using f32x4_t = __attribute__((ext_vector_type(4))) float;
constexpr int kIouChunk = 4096;

template <bool NT, int EXP = 0>
__global__ void __launch_bounds__(kThreads)
iou_map_chunk_kernel(const float *__restrict__ bboxes, int bboxes_batched, int A, const float *__restrict__ gt, int G,
                     int B, long long total, float *__restrict__ out)
{
    extern __shared__ float4 iou_lds[];                       // 2 x G gt boxes, then 2 x G gt areas
    float4 *gbox = iou_lds;
    float *garea = reinterpret_cast<float *>(iou_lds + 2 * G);
    const long long f0 = (long long)blockIdx.x * kIouChunk;   // first float of this workgroup's run
    const int per_img = A * G;
    const int b0 = (int)(f0 / per_img);                       // uniform: scalar division
    for (int i = threadIdx.x; i < 2 * G; i += kThreads) {
        const int bi = min(b0 + (i >= G ? 1 : 0), B - 1), g = i >= G ? i - G : i;
        const Box gg = load_box(gt + 4 * ((size_t)bi * G + g));
        gbox[i] = make_float4(gg.y1, gg.x1, gg.y2, gg.x2);
        garea[i] = box_area_plain(gg);                        // :138
    }
    __syncthreads();
    const long long left = total - f0;
    const int nvec = (int)((left < kIouChunk ? left : kIouChunk) >> 2);          // whole vectors (the tail is scalar)
    int v = threadIdx.x;
    const int r0 = (int)(f0 - (long long)b0 * per_img) + 4 * v;   // offset inside image b0 (may run into image b0 + 1)
    int sel = r0 >= per_img ? 1 : 0;                          // which of the two staged images
    int r = r0 - sel * per_img;
    int a = r / G;                                            // the only per-lane integer division
    int g = r - a * G;
    const int step_a = (4 * kThreads) / G, step_g = (4 * kThreads) - step_a * G;
    const size_t bstride = bboxes_batched ? 4 * (size_t)A : 0;
    for (; v < nvec; v += kThreads) {
        // the vector's first float is (image b0 + sel, anchor a, gt g); a later float may belong to the next anchor,
        // which may be anchor 0 of the next image
        const bool wrap = a + 1 == A;
        const int a1 = wrap ? 0 : a + 1, sel1 = wrap ? min(sel + 1, 1) : sel;
        const Box bx0 = load_box(bboxes + bstride * min(b0 + sel, B - 1) + 4 * (size_t)a);
        const Box bx1 = load_box(bboxes + bstride * min(b0 + sel1, B - 1) + 4 * (size_t)a1);
        const float area0 = box_area_plain(bx0), area1 = box_area_plain(bx1);    // :139
        float rr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool next = g + j >= G;                     // G >= 4: at most one anchor boundary inside a vector
            const int gj = (next ? g + j - G : g + j) + (next ? sel1 : sel) * G;
            const float4 q = gbox[gj];
            const Box gg{q.x, q.y, q.z, q.w};
            const Box bb{next ? bx1.y1 : bx0.y1, next ? bx1.x1 : bx0.x1, next ? bx1.y2 : bx0.y2, next ? bx1.x2 : bx0.x2};
            rr[j] = EXP == 2 ? bx0.y1 : iou_map_pair(bb, next ? area1 : area0, gg, garea[gj]);
            if (EXP == 1) rr[j] = rr[j] * 0.0f + bb.y1 * gg.x1;
        }
        f32x4_t val = {rr[0], rr[1], rr[2], rr[3]};
        f32x4_t *dst = reinterpret_cast<f32x4_t *>(out + f0 + 4 * (long long)v);
        if constexpr (NT) __builtin_nontemporal_store(val, dst);
        else *dst = val;
        a += step_a;
        g += step_g;
        if (g >= G) {
            g -= G;
            ++a;
        }
        if (a >= A) {
            a -= A;
            sel = 1;
        }
    }
    // scalar tail of the whole map (total % 4 floats), done by the last workgroup
    if (left <= kIouChunk && threadIdx.x < (int)(left & 3)) {
        const long long e = f0 + (left & ~3ll) + threadIdx.x;
        const int bi = (int)(e / per_img), rem = (int)(e - (long long)bi * per_img);
        const int ai = rem / G, gi = rem - ai * G;
        const Box bb = load_box(bboxes + bstride * bi + 4 * (size_t)ai);
        const Box gg = load_box(gt + 4 * ((size_t)bi * G + gi));
        out[e] = iou_map_pair(bb, box_area_plain(bb), gg, box_area_plain(gg));
    }
}

// IoU map, row form (small, conflict-friendly G: the BASELINE case G = 42).  The chunked kernel above is bound by its
// per-element bookkeeping (position walk, operand selects) and by LDS bank conflicts on the gt reads (lanes 4 gt apart);
// measured 35 us at B=64, A=8649, G=42 against 16 us for the same store pattern without the arithmetic.  Here a lane owns
// one ANCHOR (box and area in registers) and the wave walks the G gt boxes together: the gt box is wave-uniform (scalar
// loads), so an element costs the ~22 instructions of iou_map_pair and nothing else.  The 64 x G results of a wave are
// one contiguous run of the output; they are written into a wave-private LDS tile at [lane*G + g] (+ the run's
// misalignment q, so that LDS and global 16-byte boundaries coincide) and streamed out with 16-byte nontemporal stores.
// LDS write stride G words: gcd(G, 32)-way bank conflicts -- the launcher picks this kernel only for gcd <= 2 (free).
// FAST: the arithmetic of TWO gt boxes per trip on the packed-float32 pipe (v_pk_add / v_pk_mul / v_pk_fma_f32: two IEEE
// operations per issue slot), and the IEEE quotient without its range scaffolding: hipcc lowers `a / b` to v_div_scale x2,
// v_rcp, 4 fma + 1 mul, v_div_fmas, v_div_fixup; when neither operand needs rescaling (v_div_scale returns them
// unchanged) v_div_fmas IS an fma and v_div_fixup returns its input, so the seven arithmetic instructions alone give the
// same bits.  That holds while a == 0 or 2^-60 <= a <= 2^60 and 2^-60 <= b <= 2^60 (quotient and every intermediate far
// from the subnormal and overflow ranges, no NaN / inf / negative operand): each lane tracks the unsigned minimum and
// maximum of the operands' bit patterns (3 instructions per pair, v_min3_u32 / v_max3_u32), and a wave in which any lane
// ever left that range recomputes its rows with the plain `/` (never on real boxes; covered by the adversarial test).
// Per (anchor, gt) pair ~17 issue slots instead of ~26; gt areas come from one v_readlane each (lane g holds gt g's).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// v_max_f32 / v_min_f32 of a per-lane value and a wave-uniform one (SGPR operand).  fmaxf / fminf would first
// canonicalise the loaded gt coordinates (one extra v_max_f32 x, x each: 8 of them per gt pair); the bare instruction
// differs only in how a signalling NaN is passed on, and any NaN operand sends the wave to the plain-divide path.
__device__ __forceinline__ float vmax_vs(float v, float s)
{
    float r;
    asm("v_max_f32 %0, %2, %1" : "=v"(r) : "v"(v), "s"(s));
    return r;
}
__device__ __forceinline__ float vmin_vs(float v, float s)
{
    float r;
    asm("v_min_f32 %0, %2, %1" : "=v"(r) : "v"(v), "s"(s));
    return r;
}

__device__ __forceinline__ f32x2_t fma2(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }

// a / b for two pairs, correctly rounded under the range condition above (the instruction sequence of the IEEE divide)
__device__ __forceinline__ f32x2_t div2_in_range(f32x2_t a, f32x2_t b)
{
    const f32x2_t y0 = {__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y)};
    const f32x2_t nb = -b, one = {1.0f, 1.0f};
    const f32x2_t e0 = fma2(nb, y0, one);
    const f32x2_t y1 = fma2(e0, y0, y0);
    const f32x2_t q0 = a * y1;
    const f32x2_t r0 = fma2(nb, q0, a);
    const f32x2_t q1 = fma2(r0, y1, q0);
    const f32x2_t r1 = fma2(nb, q1, a);
    return fma2(r1, y1, q1);
}

// one gt pair on the packed pipe; r = the two quotients, (mn2, mx2) the range trackers
#define RPN_IOU_PAIR(G0, G1, GA0, GA1, R, MN, MX)                                                             \
    {                                                                                                  \
        const f32x2_t lo0_ = {vmax_vv(bb.x1, (G0)[1]), vmax_vv(bb.y1, (G0)[0])};     /* :141-142 */    \
        const f32x2_t hi0_ = {vmin_vv(bb.x2, (G0)[3]), vmin_vv(bb.y2, (G0)[2])};     /* :143-144 */    \
        const f32x2_t lo1_ = {vmax_vv(bb.x1, (G1)[1]), vmax_vv(bb.y1, (G1)[0])};                       \
        const f32x2_t hi1_ = {vmin_vv(bb.x2, (G1)[3]), vmin_vv(bb.y2, (G1)[2])};                       \
        const f32x2_t d0_ = hi0_ - lo0_, d1_ = hi1_ - lo1_;                           /* (dx, dy) */    \
        const f32x2_t mxs_ = {fmaxf(d0_.x, 0.0f), fmaxf(d1_.x, 0.0f)};                                 \
        const f32x2_t mys_ = {fmaxf(d0_.y, 0.0f), fmaxf(d1_.y, 0.0f)};                                 \
        const f32x2_t inter_ = mxs_ * mys_;                                           /* :146 */        \
        const f32x2_t ga_ = {(GA0), (GA1)};                                                            \
        const f32x2_t uni_ = (barea2 + ga_) - inter_;                                 /* :148 */        \
        (R) = div2_in_range(inter_, uni_);                                            /* :150 */        \
        const u32x2_t ua_ = __builtin_bit_cast(u32x2_t, inter_), ub_ = __builtin_bit_cast(u32x2_t, uni_);   \
        MN = __builtin_elementwise_min(MN, __builtin_elementwise_min(ua_ - 1u, ub_));     /* a == 0 wraps: ignored */ \
        MX = __builtin_elementwise_max(MX, __builtin_elementwise_max(ua_, ub_));                       \
    }

// Per-wave LDS: the result tile (64 * G + 4 floats, rounded up to 4) followed by the image's gt boxes (GP x 4 floats)
// and their areas (GP floats), GP = G rounded up to 4.  The gt operands are read back with one address for the whole
// wave (LDS broadcast): they arrive in vector registers without occupying a vector-ALU issue slot, several reads in
// flight, where wave-uniform scalar loads made the wave wait out the scalar-cache latency on every trip
// (rocprofv3: waves parked 35 % of their 15 k cycles in s_waitcnt, 12 waves per CU -- the kernel was bound by
// per-wave latency x occupancy, not by its stores).
__host__ __device__ inline int iou_rows_wave_floats(int G, int rows = 64) { return ((rows * G + 4 + 3) & ~3) + 5 * ((G + 3) & ~3); }

// RPN_IOU_PERSIST=1 (off by default: measured slower, 28.2 vs 26.6 us at C3) makes the waves walk several tiles.
// ROWS = 64: a lane owns one anchor and walks all G gt boxes.  ROWS = 32 (round 4): a wave's tile is 32 anchors, lane
// = (anchor lane & 31, half lane >> 5) and each half walks its own run of the gt boxes ([0, Gh) | [Gh, G), Gh = G / 2 rounded
// down to a multiple of 4; the two halves read their operands at two LDS addresses per wave, still a broadcast per 16-lane
// read group).  Half the wave-private LDS tile (6.3 KB at G = 42: 24 instead of 12 resident waves per CU by LDS) and half the
// per-wave latency: the kernel was bound by per-wave latency x occupancy (rocprofv3, round 2), not by its stores.
// (Round 5 built runs of 2 / 4 CONSECUTIVE tiles per wave -- the image's gt staged once, the next tile's anchor requested under the
// current tile's arithmetic -- parity-green and slower, 22.7 -> 26.3 / 30.7 us: the kernel lives from many short independent waves.
// Removed in round 6; NOTES.md.)
template <bool NT, int WAVES, bool FAST, int ROWS = 64>
__global__ void __launch_bounds__(64 * WAVES)
iou_map_rows_kernel(const float *__restrict__ bboxes, int bboxes_batched, int A, const float *__restrict__ gt, int G,
                    float *__restrict__ out, int tiles_per_img, int n_tiles)
{
    static_assert(ROWS == 64 || ROWS == 32, "tile");
    extern __shared__ __attribute__((aligned(16))) float iou_tile[];       // WAVES x iou_rows_wave_floats(G, ROWS)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int GP = (G + 3) & ~3;
    float *wt = iou_tile + wave * iou_rows_wave_floats(G, ROWS);
    f32x4_t *gbox_t = reinterpret_cast<f32x4_t *>(wt + ((ROWS * G + 4 + 3) & ~3));     // [GP]
    float *garea_t = reinterpret_cast<float *>(gbox_t + GP);                            // [GP]
    const int al = ROWS == 64 ? lane : (lane & 31);           // this lane's anchor inside the tile
    const int Gh = ROWS == 64 ? G : ((G >> 1) & ~3);          // the gt run of this lane: [g_lo, g_hi)
    const int g_lo = (ROWS == 64 || lane < 32) ? 0 : Gh, g_hi = (ROWS == 64 || lane >= 32) ? G : Gh;
    // one tile per trip, tiles gridDim.x * WAVES apart (one trip unless RPN_IOU_PERSIST)
    for (int tile = blockIdx.x * WAVES + wave; tile < n_tiles; tile += gridDim.x * WAVES) {
    const int b = tile / tiles_per_img;
    const int a_base = (tile - b * tiles_per_img) * ROWS;
    const int rows = min(ROWS, A - a_base);
    const int a = min(a_base + al, A - 1);                    // idle lanes recompute the last anchor, results unused
    const float *__restrict__ gtb = gt + 4 * (size_t)b * G;
    const Box bb = load_box(bboxes + (bboxes_batched ? 4 * (size_t)b * A : 0) + 4 * (size_t)a);
    if constexpr (FAST) {                                     // lane g stages gt box g and its area (G <= 64)
        if (lane < GP) {
            const Box gl = load_box(gtb + 4 * min(lane, G - 1));
            gbox_t[lane] = f32x4_t{gl.y1, gl.x1, gl.y2, gl.x2};
            garea_t[lane] = box_area_plain(gl);
        }
    }
    const float barea = box_area_plain(bb);                   // :139
    const long long s = ((long long)b * A + a_base) * G;      // first float of this wave's run
    const int q = (int)(s & 3);
    float *row = wt + q + al * G;
    bool exact_needed = !FAST;
    if constexpr (FAST) {
        constexpr unsigned kLo = 0x21800000u, kHi = 0x5D800000u;            // bits of 2^-60, 2^60
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        u32x2_t mn2 = {0xFFFFFFFFu, 0xFFFFFFFFu}, mx2 = {0u, 0u}, mn2b = mn2, mx2b = mx2;     // one tracker pair per chain
        const f32x2_t barea2 = {barea, barea};
        // 4 gt boxes (two packed pairs = two independent dependency chains) per trip; the next trip's operands are
        // requested before this trip's arithmetic
        const int G4 = g_lo + ((g_hi - g_lo) & ~3);           // whole trips of this lane's run ([g_lo, g_hi), g_lo a multiple of 4)
        f32x4_t c0 = gbox_t[g_lo], c1 = gbox_t[g_lo + 1], c2 = gbox_t[g_lo + 2], c3 = gbox_t[g_lo + 3];
        f32x4_t ca = *reinterpret_cast<const f32x4_t *>(garea_t + g_lo);
        for (int g = g_lo; g < G4; g += 4) {
            const int gn = g + 4 < GP ? g + 4 : g;            // (GP is a multiple of 4: in bounds; clamped on the last trip)
            const f32x4_t n0 = gbox_t[gn], n1 = gbox_t[gn + 1], n2 = gbox_t[gn + 2], n3 = gbox_t[gn + 3];
            const f32x4_t na = *reinterpret_cast<const f32x4_t *>(garea_t + gn);
            f32x2_t r01, r23;
            RPN_IOU_PAIR(c0, c1, ca[0], ca[1], r01, mn2, mx2);
            RPN_IOU_PAIR(c2, c3, ca[2], ca[3], r23, mn2b, mx2b);
            row[g] = r01.x;
            row[g + 1] = r01.y;
            row[g + 2] = r23.x;
            row[g + 3] = r23.y;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3; ca = na;
        }
        if (G4 + 2 <= g_hi) {                                  // c0.. hold gt G4 .. G4+3 (staged rows beyond G repeat G-1)
            f32x2_t r01;
            RPN_IOU_PAIR(c0, c1, ca[0], ca[1], r01, mn2, mx2);
            row[G4] = r01.x;
            row[G4 + 1] = r01.y;
            if (G4 + 3 == g_hi) {                              // odd run: the last gt box alone, plain divide
                const Box gg = load_box(gtb + 4 * (g_hi - 1));
                row[g_hi - 1] = iou_map_pair(bb, barea, gg, box_area_plain(gg));
            }
        } else if (G4 < g_hi) {
            const Box gg = load_box(gtb + 4 * (g_hi - 1));
            row[g_hi - 1] = iou_map_pair(bb, barea, gg, box_area_plain(gg));
        }
        const unsigned mn = min(min(mn2.x, mn2.y), min(mn2b.x, mn2b.y)), mx = max(max(mx2.x, mx2.y), max(mx2b.x, mx2b.y));
        exact_needed = __any(mn < kLo - 1u || mx > kHi) != 0;
    }
    if (exact_needed) {
        // 4 independent pairs per trip: one pair is a ~20-deep dependent chain (max/min, products, an IEEE divide) and a
        // wave alone issues a dependent instruction only every 4-8 cycles
        int g = g_lo;
        for (; g + 4 <= g_hi; g += 4) {
            float r4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const Box gg = load_box(gtb + 4 * (g + j));       // wave-uniform address (per half)
                r4[j] = iou_map_pair(bb, barea, gg, box_area_plain(gg));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) row[g + j] = r4[j];
        }
        for (; g < g_hi; ++g) {
            const Box gg = load_box(gtb + 4 * g);
            row[g] = iou_map_pair(bb, barea, gg, box_area_plain(gg));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n = rows * G;                                   // valid floats: LDS [q, q + n)
    float *gbase = out + (s - q);                             // 16-byte aligned; LDS index i <-> gbase[i]
    const int v0 = (q + 3) >> 2, v1 = (q + n) >> 2;           // whole vectors [v0, v1)
    int v = v0 + lane;
    for (; v + 192 < v1; v += 256) {                          // 4 reads in flight, then their 4 stores
        f32x4_t val[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) val[j] = *reinterpret_cast<const f32x4_t *>(wt + 4 * (v + 64 * j));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4_t *dst = reinterpret_cast<f32x4_t *>(gbase + 4 * (v + 64 * j));
            if constexpr (NT) __builtin_nontemporal_store(val[j], dst);
            else *dst = val[j];
        }
    }
    for (; v < v1; v += 64) {
        const f32x4_t val = *reinterpret_cast<const f32x4_t *>(wt + 4 * v);
        f32x4_t *dst = reinterpret_cast<f32x4_t *>(gbase + 4 * v);
        if constexpr (NT) __builtin_nontemporal_store(val, dst);
        else *dst = val;
    }
    // head [q, 4*v0) and tail [max(4*v1, q), q + n): at most 3 floats each
    const int head_end = min(4 * v0, q + n);
    if (lane < head_end - q) gbase[q + lane] = wt[q + lane];
    const int tail0 = max(4 * v1, head_end);
    if (lane < q + n - tail0) gbase[tail0 + lane] = wt[tail0 + lane];
    // (the next tile's LDS writes stay behind these reads: a wave's LDS operations execute in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
#undef RPN_IOU_PAIR

// normalize_bboxes / denormalize_bboxes (utils/bbox_utils.py:152-182): per-coordinate divide / multiply by the
// image height (y) or width (x); denormalize rounds half-to-even like tf.round (rintf in the default mode)
__global__ void __launch_bounds__(kThreads)
scale_boxes_kernel(const float *__restrict__ in, float height, float width, int denorm, long long nboxes,
                   float *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < nboxes;
         i += (long long)gridDim.x * kThreads) {
        const Box b = load_box(in + 4 * i);
        Box o;
        if (denorm) {
            o.y1 = rintf(b.y1 * height);
            o.x1 = rintf(b.x1 * width);
            o.y2 = rintf(b.y2 * height);
            o.x2 = rintf(b.x2 * width);
        } else {
            o.y1 = b.y1 / height;
            o.x1 = b.x1 / width;
            o.y2 = b.y2 / height;
            o.x2 = b.x2 / width;
        }
        store_box(out + 4 * i, o);
    }
}

// Input preprocessing (utils/data_utils.py:25-28): uint8 HWC -> float32 in [0,1] (one multiply by 1/255, as
// tf.image.convert_image_dtype), bilinear resize with half-pixel centres and no antialias (TF 2.x
// tf.image.resize; op order of resize_bilinear_op.cc), optional tf.image.flip_left_right.  One thread per output
// pixel (3 channels); HBM-bound: ~4 source pixels x 3 B read (L2-served) + 12 B written per output pixel.
__global__ void __launch_bounds__(kThreads)
preprocess_kernel(const unsigned char *__restrict__ img, int H, int W, int OH, int OW, int flip,
                  float *__restrict__ out)
{
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    const float k255 = (float)(1.0 / 255.0);
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < OH * OW; i += gridDim.x * kThreads) {
        const int oy = i / OW, oxo = i - oy * OW;
        const int ox = flip ? (OW - 1 - oxo) : oxo;               // flip_left_right of the resized image
        const float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
        const float fyf = floorf(fy), fxf = floorf(fx);
        const int y0 = max((int)fyf, 0), y1 = min((int)ceilf(fy), H - 1);
        const int x0 = max((int)fxf, 0), x1 = min((int)ceilf(fx), W - 1);
        const float ly = fy - fyf, lx = fx - fxf;
        const unsigned char *p00 = img + ((size_t)y0 * W + x0) * 3, *p01 = img + ((size_t)y0 * W + x1) * 3;
        const unsigned char *p10 = img + ((size_t)y1 * W + x0) * 3, *p11 = img + ((size_t)y1 * W + x1) * 3;
        float *o = out + (size_t)i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float tl = (float)p00[c] * k255, tr = (float)p01[c] * k255;
            const float bl = (float)p10[c] * k255, br = (float)p11[c] * k255;
            const float top = tl + (tr - tl) * lx;
            const float bottom = bl + (br - bl) * lx;
            o[c] = top + (bottom - top) * ly;
        }
    }
}

static int grid_for(long long work_items)
{
    long long g = (work_items + kThreads - 1) / kThreads;
    if (g < 1) g = 1;
    if (g > kMaxGrid) g = kMaxGrid;
    return (int)g;
}

}  // namespace rpn

using namespace rpn;

extern "C" int rpn_generate_anchors(double img_size, int F, const double *ratios, int nr, const double *scales,
                                    int ns, float *d_anchors, void *stream)
{
    RPN_REQUIRE(ratios && scales && d_anchors, "rpn_generate_anchors: null pointer");
    RPN_REQUIRE(F > 0 && nr > 0 && ns > 0 && img_size > 0, "rpn_generate_anchors: non-positive size");
    const int K = nr * ns;
    RPN_REQUIRE(K <= kMaxBase, "rpn_generate_anchors: %d anchors per cell > %d supported", K, kMaxBase);
    RPN_REQUIRE((long long)F * F * K < (1ll << 31), "rpn_generate_anchors: too many anchors");
    RPN_REQUIRE_DEVICE();
    // generate_base_anchors (utils/bbox_utils.py:3-21): python-double scale arithmetic, then
    // float32 sqrt of the float32-converted argument, float32 multiply by float32(ratio).
    BaseAnchors base{};
    int k = 0;
    for (int si = 0; si < ns; ++si) {
        const double scale = scales[si] / img_size;                     // :16
        for (int ri = 0; ri < nr; ++ri, ++k) {
            volatile float arg = (float)(scale * scale / ratios[ri]);   // :18 double -> f32
            volatile float w = sqrtf(arg);                              //     f32 sqrt (correctly rounded)
            volatile float h = w * (float)ratios[ri];                   // :19
            base.v[4 * k + 0] = -h / 2.0f;                              // :20
            base.v[4 * k + 1] = -w / 2.0f;
            base.v[4 * k + 2] = h / 2.0f;
            base.v[4 * k + 3] = w / 2.0f;
        }
    }
    const double half_stride = (1.0 / (double)F) / 2.0;                 // :35-36
    const int A = F * F * K;
    hipLaunchKernelGGL(anchors_kernel, dim3(grid_for(A)), dim3(kThreads), 0, as_stream(stream), base, F, K,
                       half_stride, d_anchors, A);
    RPN_CHECK_LAUNCH();
    return RPN_OK;
}

extern "C" int rpn_decode(const float *d_anchors, int anchors_batched, const float *d_deltas,
                          const float *variances, int B, int A, float *d_boxes, void *stream)
{
    RPN_REQUIRE(B >= 0 && A >= 0, "rpn_decode: negative size");
    if ((long long)B * A == 0) return RPN_OK;
    RPN_REQUIRE(d_anchors && d_deltas && d_boxes, "rpn_decode: null pointer");
    RPN_REQUIRE_DEVICE();
    Variances var{};
    if (variances) {
        for (int i = 0; i < 4; ++i) var.v[i] = variances[i];
        var.enabled = 1;
    }
    RPN_REQUIRE(B <= 65535, "rpn_decode: batch %d > 65535", B);
    hipLaunchKernelGGL(decode_kernel, dim3(grid_for(A), B), dim3(kThreads), 0, as_stream(stream), d_anchors,
                       anchors_batched, d_deltas, var, A, d_boxes);
    RPN_CHECK_LAUNCH();
    return RPN_OK;
}

extern "C" int rpn_encode(const float *d_bboxes, int bboxes_batched, const float *d_gt, int B, int A,
                          float *d_deltas, void *stream)
{
    RPN_REQUIRE(B >= 0 && A >= 0, "rpn_encode: negative size");
    if ((long long)B * A == 0) return RPN_OK;
    RPN_REQUIRE(d_bboxes && d_gt && d_deltas, "rpn_encode: null pointer");
    RPN_REQUIRE_DEVICE();
    RPN_REQUIRE(B <= 65535, "rpn_encode: batch %d > 65535", B);
    hipLaunchKernelGGL(encode_kernel, dim3(grid_for(A), B), dim3(kThreads), 0, as_stream(stream), d_bboxes,
                       bboxes_batched, d_gt, A, d_deltas);
    RPN_CHECK_LAUNCH();
    return RPN_OK;
}

static bool nt_rows()
{
    static const int nt = RPN_LAB_KNOB("RPN_IOU_NT", 1);
    return nt != 0;
}

extern "C" int rpn_iou_map(const float *d_bboxes, int bboxes_batched, int A, const float *d_gt, int B, int G,
                           float *d_iou, void *stream)
{
    RPN_REQUIRE(B >= 0 && A >= 0 && G >= 0, "rpn_iou_map: negative size");
    const long long total = (long long)B * A * G;
    if (total == 0) return RPN_OK;
    RPN_REQUIRE(d_bboxes && d_gt && d_iou, "rpn_iou_map: null pointer");
    RPN_REQUIRE_DEVICE();
    RPN_REQUIRE((long long)A * G < (1ll << 31) && B <= 65535, "rpn_iou_map: A*G or B too large");
    const int per_img = A * G;
    static const int chunked = RPN_LAB_KNOB("RPN_IOU_CHUNKED", 1);
    static const int nt = RPN_LAB_KNOB("RPN_IOU_NT", 1);
    static const int rowsk = RPN_LAB_KNOB("RPN_IOU_ROWS", 1);
    const int g32 = (G % 4 == 0) ? 4 : (G % 2 == 0 ? 2 : 1);          // >= gcd(G, 32) capped at 4
    static const int half_rows = RPN_LAB_KNOB("RPN_IOU_HALF", 0);   // 32-anchor tiles, the gt run split over the wave's halves
    const bool half = half_rows && G >= 16;
    const int tile_rows = half ? 32 : 64;
    const size_t wave_lds = (size_t)iou_rows_wave_floats(G, tile_rows) * sizeof(float);
    if (rowsk && G >= 1 && g32 <= 2 && (size_t)iou_rows_wave_floats(G) * sizeof(float) <= 12 * 1024 && A >= 64) {
        // one tile (64 anchors x G) per wave, 4 waves per workgroup (single-wave workgroups measured 30.4 vs 26.5 us at C3).
        // RPN_IOU_PERSIST=1: as many workgroups as stay resident, every wave walking the same number of tiles
        const int tiles_per_img = (A + tile_rows - 1) / tile_rows;
        const long long n_tiles = (long long)tiles_per_img * B;
        RPN_REQUIRE(n_tiles <= 0x7fffffffll, "rpn_iou_map: too many tiles");
        static const int persist = RPN_LAB_KNOB("RPN_IOU_PERSIST", 0);
        static const int fast = RPN_LAB_KNOB("RPN_IOU_FAST", 1);
        static const int wv = RPN_LAB_KNOB("RPN_IOU_WAVES", 4);     // waves per workgroup: 4 | 2
        const int W = wv == 2 ? 2 : 4;
        long long wgs = (n_tiles + W - 1) / W;
        if (persist) {
            const long long resident = 256ll * (long long)((160 * 1024) / (W * wave_lds));
            if (wgs > resident) {
                const long long rounds = (n_tiles + W * resident - 1) / (W * resident);
                wgs = ((n_tiles + rounds - 1) / rounds + W - 1) / W;
            }
        }
        const dim3 grid((unsigned)wgs);
        const int nt_i = (int)n_tiles;
#define RPN_IOU_ROWS(NT_, W_, FAST_, ...)                                                                             \
    hipLaunchKernelGGL((iou_map_rows_kernel<NT_, W_, FAST_, ##__VA_ARGS__>), grid, dim3(64 * W_), W_ * wave_lds, as_stream(stream), d_bboxes, \
                       bboxes_batched, A, d_gt, G, d_iou, tiles_per_img, nt_i)
        if (half && nt_rows() && fast && W == 4) RPN_IOU_ROWS(true, 4, true, 32);
        else if (half) return rpn::fail(RPN_ERR_INVALID, "rpn_iou_map: RPN_IOU_HALF needs the default row-kernel settings");
        else if (!nt_rows()) RPN_IOU_ROWS(false, 4, true);
        else if (!fast) RPN_IOU_ROWS(true, 4, false);
        else if (W == 2) RPN_IOU_ROWS(true, 2, true);
        else RPN_IOU_ROWS(true, 4, true);
#undef RPN_IOU_ROWS
    } else if (chunked && G >= 4 && G <= 2048 && per_img >= kIouChunk && total < (1ll << 40)) {
        const long long chunks = (total + kIouChunk - 1) / kIouChunk;
        RPN_REQUIRE(chunks <= 0x7fffffffll, "rpn_iou_map: too many chunks");
        const size_t lds = (size_t)G * 40;
#ifdef RPN_LAB      /* timing experiment: the map then holds coordinates, not IoUs */
        static const int iexp = RPN_LAB_KNOB("RPN_IOU_EXP", 0);
        if (iexp == 2)
            hipLaunchKernelGGL((iou_map_chunk_kernel<true, 2>), dim3((unsigned)chunks), dim3(kThreads), lds, as_stream(stream),
                               d_bboxes, bboxes_batched, A, d_gt, G, B, total, d_iou);
        else
#endif
        if (nt)
            hipLaunchKernelGGL(iou_map_chunk_kernel<true>, dim3((unsigned)chunks), dim3(kThreads), lds, as_stream(stream),
                               d_bboxes, bboxes_batched, A, d_gt, G, B, total, d_iou);
        else
            hipLaunchKernelGGL(iou_map_chunk_kernel<false>, dim3((unsigned)chunks), dim3(kThreads), lds, as_stream(stream),
                               d_bboxes, bboxes_batched, A, d_gt, G, B, total, d_iou);
    } else {
        hipLaunchKernelGGL(iou_map_kernel, dim3(grid_for((per_img + 3) / 4), B), dim3(kThreads), 0, as_stream(stream),
                           d_bboxes, bboxes_batched, A, d_gt, G, (per_img % 4 == 0) ? 1 : 0, d_iou);
    }
    RPN_CHECK_LAUNCH();
    return RPN_OK;
}

extern "C" int rpn_scale_boxes(const float *d_boxes, long long nboxes, float height, float width, int denormalize,
                               float *d_out, void *stream)
{
    RPN_REQUIRE(nboxes >= 0, "rpn_scale_boxes: negative size");
    if (nboxes == 0) return RPN_OK;
    RPN_REQUIRE(d_boxes && d_out, "rpn_scale_boxes: null pointer");
    RPN_REQUIRE_DEVICE();
    hipLaunchKernelGGL(scale_boxes_kernel, dim3(grid_for(nboxes)), dim3(kThreads), 0, as_stream(stream), d_boxes,
                       height, width, denormalize ? 1 : 0, nboxes, d_out);
    RPN_CHECK_LAUNCH();
    return RPN_OK;
}

extern "C" int rpn_preprocess_image(const unsigned char *d_img_u8, int H, int W, int out_h, int out_w, int flip,
                                    float *d_out, void *stream)
{
    RPN_REQUIRE(d_img_u8 && d_out, "rpn_preprocess_image: null pointer");
    RPN_REQUIRE(H >= 1 && W >= 1 && out_h >= 1 && out_w >= 1 && (long long)out_h * out_w < (1ll << 31) &&
                    (long long)H * W < (1ll << 31),
                "rpn_preprocess_image: bad size");
    RPN_REQUIRE_DEVICE();
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for((long long)out_h * out_w)), dim3(kThreads), 0,
                       as_stream(stream), d_img_u8, H, W, out_h, out_w, flip ? 1 : 0, d_out);
    RPN_CHECK_LAUNCH();
    return RPN_OK;
}
