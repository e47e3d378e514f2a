// target_kernels.hip -- RPN training-target assignment (utils/train_utils.py:84-144,
// calculate_rpn_actual_outputs + randomly_select_xyz_mask :50-65) for gfx950.
//
// The reference materialises the (B,A,G) IoU map (93 MB at B=64, A=8649, G=42) only to reduce it three
// ways; here the map never exists:
//   iou_argmax_kernel : per (image, anchor) the IoU row is evaluated in registers against the image's gt boxes
//                       (LDS), keeping max / first-argmax over gt; the per-gt argmax over anchors is a 64-bit
//                       atomicMax of (orderable(iou) << 32 | ~anchor) after a wave-level max (first anchor wins).
//   target_select_kernel : one workgroup per image: positive mask (IoU > 0.7 or best anchor of a valid gt),
//                       exact top-K by random priority (radix select on (random << 32 | ~index): the reference's
//                       double argsort, ties to the lower index), negatives (IoU < 0.3, not positive) likewise,
//                       then labels {1, 0, -1} and the encoded deltas / variances of the positives.
// The two tf.random.uniform draws are explicit inputs (TF's generator cannot be reproduced).
// HBM traffic: 16*A + 16*B*G read, 28*B*A written/re-read in the workspace, 20*B*A written as outputs.
// Compiled with -ffp-contract=off (same arithmetic as generate_iou_map / get_deltas_from_bboxes).
#include "bbox_core.h"
#include "radix_select.h"
#include "rpn_common.h"

namespace rpn {

constexpr int kTgtThreads = 1024;

struct TargetArgs {
    const float *anchors;     // (A,4)
    const float *gt;          // (B,G,4)
    const int *labels;        // (B,G), -1 = padding
    const int *rand_pos;      // (B,A) >= 1
    const int *rand_neg;      // (B,A) >= 1
    int B, A, G;
    int total_pos, total_neg;
    float var[4];
    float pos_thr, neg_thr;   // 0.7, 0.3 (train_utils.py:114,128)
    float *out_deltas;        // (B,A,4)
    float *out_labels;        // (B,A)
    // workspace
    float *merged;            // (B,A) max IoU over gt
    int *argrow;              // (B,A) first argmax over gt
    unsigned long long *colkey;   // (B,G) max over anchors of (orderable(iou) << 32 | ~anchor)
    unsigned char *best;      // (B,A) 1 = best anchor of some valid gt
};

__global__ void __launch_bounds__(256)
iou_argmax_kernel(TargetArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float gts[];      // [G][4] + area[G]
    const int b = blockIdx.y;
    const int G = p.G;
    float *garea = gts + 4 * G;
    for (int i = threadIdx.x; i < G; i += 256) {
        const Box g = load_box(p.gt + 4 * ((size_t)b * G + i));
        store_box(gts + 4 * i, g);
        garea[i] = box_area_plain(g);                                // :138
    }
    __syncthreads();
    const int a = blockIdx.x * 256 + threadIdx.x;
    const bool live = a < p.A;
    Box bb{0.f, 0.f, 0.f, 0.f};
    float barea = 0.f;
    if (live) {
        bb = load_box(p.anchors + 4 * (size_t)a);
        barea = box_area_plain(bb);                                  // :139
    }
    float best = 0.f;
    int arg = 0;
    for (int g = 0; g < G; ++g) {
        const Box gg = load_box(gts + 4 * g);                        // LDS broadcast
        const float iou = iou_map_pair(bb, barea, gg, garea[g]);     // :141-150
        if (g == 0 || iou > best) {                                  // argmax: first maximum (:108)
            best = iou;
            arg = g;
        }
        // per-gt argmax over anchors (:110): wave max of the key, one atomic per wave and gt
        unsigned long long key = live ? (((unsigned long long)orderable(iou) << 32) |
                                         (unsigned long long)(0xFFFFFFFFu - (unsigned)a)) : 0ull;
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_down(key, off, 64);
            key = o > key ? o : key;
        }
        if ((threadIdx.x & 63) == 0 && key) atomicMax(&p.colkey[(size_t)b * G + g], key);
    }
    if (live) {
        p.merged[(size_t)b * p.A + a] = best;                        // :112
        p.argrow[(size_t)b * p.A + a] = arg;
    }
}

__global__ void __launch_bounds__(kTgtThreads)
target_select_kernel(TargetArgs p)
{
    __shared__ unsigned hist[kRsHistWords];
    __shared__ int ctl[8];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int A = p.A, G = p.G;
    const float *merged = p.merged + (size_t)b * A;
    unsigned char *best = p.best + (size_t)b * A;
    const int *rpos = p.rand_pos + (size_t)b * A, *rneg = p.rand_neg + (size_t)b * A;

    // scatter: the best anchor of every valid gt box is positive (:116-122); best[] was zeroed by the launcher
    for (int g = tid; g < G; g += kTgtThreads)
        if (p.labels[(size_t)b * G + g] != -1) {
            const unsigned a = 0xFFFFFFFFu - (unsigned)(p.colkey[(size_t)b * G + g] & 0xFFFFFFFFull);
            if (a < (unsigned)A) best[a] = 1;
        }
    __syncthreads();

    auto key_pos = [&](int i) -> unsigned long long {
        const bool m = merged[i] > p.pos_thr || best[i] != 0;                               // :114, :122
        return m ? (((unsigned long long)(unsigned)rpos[i] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i)) : 0ull;
    };
    int pos_count = 0;
    unsigned long long thr_pos = 0ull;
    if (p.total_pos > 0)
        thr_pos = radix_select<kTgtThreads>(key_pos, A, ~0ull, p.total_pos, p.total_pos, hist, ctl, &pos_count);   // :123
    if (thr_pos == 0ull) pos_count = 0;
    auto is_pos = [&](int i) -> bool {
        const unsigned long long k = key_pos(i);
        return thr_pos != 0ull && k != 0ull && k >= thr_pos;
    };
    const int neg_want = p.total_pos + p.total_neg - pos_count;                              // :126
    auto key_neg = [&](int i) -> unsigned long long {
        const bool m = merged[i] < p.neg_thr && !is_pos(i);                                  // :128
        return m ? (((unsigned long long)(unsigned)rneg[i] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i)) : 0ull;
    };
    int neg_count = 0;
    unsigned long long thr_neg = 0ull;
    if (neg_want > 0) thr_neg = radix_select<kTgtThreads>(key_neg, A, ~0ull, neg_want, neg_want, hist, ctl, &neg_count);   // :129

    for (int i = tid; i < A; i += kTgtThreads) {
        const bool pos = is_pos(i);
        const unsigned long long kn = key_neg(i);
        const bool neg = thr_neg != 0ull && kn != 0ull && kn >= thr_neg;
        p.out_labels[(size_t)b * A + i] = (pos ? 1.0f : -1.0f) + (neg ? 1.0f : 0.0f);       // :131-133
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pos) {                                                                           // :135-139
            const Box gtb = load_box(p.gt + 4 * ((size_t)b * G + p.argrow[(size_t)b * A + i]));
            d = encode_box(load_box(p.anchors + 4 * (size_t)i), gtb);
        }
        // non-positive anchors get an all-zero "gt" box: zero width/height -> zero deltas (:117-122), 0 / v = 0
        d.x = d.x / p.var[0];
        d.y = d.y / p.var[1];
        d.z = d.z / p.var[2];
        d.w = d.w / p.var[3];
        *reinterpret_cast<float4 *>(p.out_deltas + 4 * ((size_t)b * A + i)) = d;
    }
}

static size_t a256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace rpn

using namespace rpn;

extern "C" size_t rpn_targets_workspace_bytes(int B, int A, int G)
{
    if (B <= 0 || A <= 0 || G <= 0) return 0;
    return a256((size_t)B * A * 4) + a256((size_t)B * A * 4) + a256((size_t)B * G * 8) + a256((size_t)B * A);
}

extern "C" int rpn_rpn_targets(const float *d_anchors, const float *d_gt_boxes, const int32_t *d_gt_labels, int B, int A,
                               int G, int total_pos, int total_neg, const float *variances, const int32_t *d_random_pos,
                               const int32_t *d_random_neg, float *d_bbox_deltas, float *d_bbox_labels, void *d_workspace,
                               size_t workspace_bytes, void *stream)
{
    RPN_REQUIRE(B >= 0 && A >= 1 && G >= 1 && total_pos >= 0 && total_neg >= 0, "rpn_rpn_targets: bad sizes");
    if (B == 0) return RPN_OK;
    RPN_REQUIRE(d_anchors && d_gt_boxes && d_gt_labels && variances && d_random_pos && d_random_neg && d_bbox_deltas &&
                    d_bbox_labels,
                "rpn_rpn_targets: null pointer");
    RPN_REQUIRE(B <= 65535 && G <= 2048, "rpn_rpn_targets: B > 65535 or G > 2048");
    const size_t need = rpn_targets_workspace_bytes(B, A, G);
    if (!d_workspace || workspace_bytes < need)
        return fail(RPN_ERR_WORKSPACE, "rpn_rpn_targets: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    RPN_REQUIRE_DEVICE();
    hipStream_t s = as_stream(stream);
    TargetArgs p{};
    p.anchors = d_anchors; p.gt = d_gt_boxes; p.labels = d_gt_labels;
    p.rand_pos = d_random_pos; p.rand_neg = d_random_neg;
    p.B = B; p.A = A; p.G = G; p.total_pos = total_pos; p.total_neg = total_neg;
    for (int i = 0; i < 4; ++i) p.var[i] = variances[i];
    p.pos_thr = 0.7f; p.neg_thr = 0.3f;
    p.out_deltas = d_bbox_deltas; p.out_labels = d_bbox_labels;
    unsigned char *ws = reinterpret_cast<unsigned char *>(d_workspace);
    p.merged = reinterpret_cast<float *>(ws);            ws += a256((size_t)B * A * 4);
    p.argrow = reinterpret_cast<int *>(ws);              ws += a256((size_t)B * A * 4);
    p.colkey = reinterpret_cast<unsigned long long *>(ws);
    const size_t zero_bytes = a256((size_t)B * G * 8) + a256((size_t)B * A);      // colkey + best, contiguous
    ws += a256((size_t)B * G * 8);
    p.best = ws;
    RPN_HIP_CHECK(hipMemsetAsync(p.colkey, 0, zero_bytes, s));
    hipLaunchKernelGGL(iou_argmax_kernel, dim3((A + 255) / 256, B), dim3(256), (size_t)G * 20, s, p);
    RPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(target_select_kernel, dim3(B), dim3(kTgtThreads), 0, s, p);
    RPN_CHECK_LAUNCH();
    return RPN_OK;
}
