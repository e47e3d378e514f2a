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

__global__ void __launch_bounds__(kThreads)
decode_kernel(const float *__restrict__ anchors, int anchors_batched, const float *__restrict__ deltas,
              Variances var, int A, long long total, float *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total;
         i += (long long)gridDim.x * kThreads) {
        const long long ai = anchors_batched ? i : (i % A);
        const Box an = load_box(anchors + 4 * ai);
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
              long long total, float *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total;
         i += (long long)gridDim.x * kThreads) {
        const long long bi = bboxes_batched ? i : (i % A);
        const float4 d = encode_box(load_box(bboxes + 4 * bi), load_box(gt + 4 * i));
        *reinterpret_cast<float4 *>(out + 4 * i) = d;
    }
}

// IoU map: each lane produces 4 consecutive floats of the flat (B,A,G) output and stores
// them with one 16-byte store (when the total is a multiple of 4; otherwise the tail lane
// falls back to scalar stores).  The (anchor, gt) operands are a few hundred KB and are
// served by L1/L2; the kernel is bound by the 4*B*A*G output bytes.
__global__ void __launch_bounds__(kThreads)
iou_map_kernel(const float *__restrict__ bboxes, int bboxes_batched, int A, const float *__restrict__ gt,
               int G, long long total, float *__restrict__ out)
{
    const long long nvec = (total + 3) / 4;
    for (long long v = (long long)blockIdx.x * kThreads + threadIdx.x; v < nvec;
         v += (long long)gridDim.x * kThreads) {
        const long long e0 = v * 4;
        long long row = e0 / G;                 // row = b*A + a
        int g = (int)(e0 - row * G);
        float r[4];
        long long cur_row = -1;
        Box bb{};
        float barea = 0.0f;
        long long b = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (e0 + j < total) {
                if (row != cur_row) {
                    b = row / A;
                    const long long a = row - b * A;
                    bb = load_box(bboxes + 4 * (bboxes_batched ? row : a));
                    barea = box_area_plain(bb);             // :139
                    cur_row = row;
                }
                const Box gg = load_box(gt + 4 * (b * G + g));
                r[j] = iou_map_pair(bb, barea, gg, box_area_plain(gg));
                if (++g == G) {
                    g = 0;
                    ++row;
                }
            } else {
                r[j] = 0.0f;
            }
        }
        if (e0 + 3 < total) {
            *reinterpret_cast<float4 *>(out + e0) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
            for (int j = 0; j < 4 && e0 + j < total; ++j) out[e0 + j] = r[j];
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
    const long long total = (long long)B * A;
    hipLaunchKernelGGL(decode_kernel, dim3(grid_for(total)), dim3(kThreads), 0, as_stream(stream), d_anchors,
                       anchors_batched, d_deltas, var, A, total, d_boxes);
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
    const long long total = (long long)B * A;
    hipLaunchKernelGGL(encode_kernel, dim3(grid_for(total)), dim3(kThreads), 0, as_stream(stream), d_bboxes,
                       bboxes_batched, d_gt, A, total, d_deltas);
    RPN_CHECK_LAUNCH();
    return RPN_OK;
}

extern "C" int rpn_iou_map(const float *d_bboxes, int bboxes_batched, int A, const float *d_gt, int B, int G,
                           float *d_iou, void *stream)
{
    RPN_REQUIRE(B >= 0 && A >= 0 && G >= 0, "rpn_iou_map: negative size");
    const long long total = (long long)B * A * G;
    if (total == 0) return RPN_OK;
    RPN_REQUIRE(d_bboxes && d_gt && d_iou, "rpn_iou_map: null pointer");
    RPN_REQUIRE_DEVICE();
    hipLaunchKernelGGL(iou_map_kernel, dim3(grid_for((total + 3) / 4)), dim3(kThreads), 0, as_stream(stream),
                       d_bboxes, bboxes_batched, A, d_gt, G, total, d_iou);
    RPN_CHECK_LAUNCH();
    return RPN_OK;
}
