// bbox_core.h -- per-box arithmetic of the proposal path, written once for device code.
//
// Every function keeps the reference's operation order exactly (one float32 rounding
// per multiply / add; this translation unit is compiled with -ffp-contract=off so no
// FMA is formed) because NMS keep-masks and anchor values are required to be bit-exact.
// Box order: [y1, x1, y2, x2].  Citations are /root/reference paths.
#pragma once
#include <hip/hip_runtime.h>

namespace rpn {

struct Box {
    float y1, x1, y2, x2;
};

__device__ __forceinline__ Box load_box(const float *p)
{
    const float4 v = *reinterpret_cast<const float4 *>(p);
    return Box{v.x, v.y, v.z, v.w};
}

__device__ __forceinline__ void store_box(float *p, const Box &b)
{
    *reinterpret_cast<float4 *>(p) = make_float4(b.y1, b.x1, b.y2, b.x2);
}

__device__ __forceinline__ float clip01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

// get_bboxes_from_deltas -- utils/bbox_utils.py:81-94; d = [dy, dx, dh, dw] (already x variances)
__device__ __forceinline__ Box decode_box(const Box &an, float dy, float dx, float dh, float dw)
{
    const float w = an.x2 - an.x1;              // :81
    const float h = an.y2 - an.y1;              // :82
    const float cx = an.x1 + 0.5f * w;          // :83
    const float cy = an.y1 + 0.5f * h;          // :84
    const float bw = expf(dw) * w;              // :86
    const float bh = expf(dh) * h;              // :87
    const float bcx = (dx * w) + cx;            // :88
    const float bcy = (dy * h) + cy;            // :89
    Box o;
    o.y1 = bcy - (0.5f * bh);                   // :91
    o.x1 = bcx - (0.5f * bw);                   // :92
    o.y2 = bh + o.y1;                           // :93
    o.x2 = bw + o.x1;                           // :94
    return o;
}

// get_deltas_from_bboxes -- utils/bbox_utils.py:107-124; returns [dy, dx, dh, dw]
__device__ __forceinline__ float4 encode_box(const Box &bb, const Box &gt)
{
    float bw = bb.x2 - bb.x1, bh = bb.y2 - bb.y1;
    const float bcx = bb.x1 + 0.5f * bw, bcy = bb.y1 + 0.5f * bh;
    const float gw = gt.x2 - gt.x1, gh = gt.y2 - gt.y1;
    const float gcx = gt.x1 + 0.5f * gw, gcy = gt.y1 + 0.5f * gh;
    if (bw == 0.0f) bw = 1e-3f;                                  // :117
    if (bh == 0.0f) bh = 1e-3f;                                  // :118
    float4 d;
    d.y = (gw == 0.0f) ? 0.0f : (gcx - bcx) / bw;                // :119 dx
    d.x = (gh == 0.0f) ? 0.0f : (gcy - bcy) / bh;                // :120 dy
    d.w = (gw == 0.0f) ? 0.0f : logf(gw / bw);                   // :121 dw
    d.z = (gh == 0.0f) ? 0.0f : logf(gh / bh);                   // :122 dh
    return d;
}

// generate_iou_map -- utils/bbox_utils.py:138-150 for one (bbox, gt) pair
__device__ __forceinline__ float iou_map_pair(const Box &b, float b_area, const Box &g, float g_area)
{
    // fmaxf / fminf = one v_max_f32 / v_min_f32 each (a compare + select pair otherwise).  They differ from tf.maximum only
    // in which operand survives a NaN, and a NaN coordinate makes its box's area -- hence the union and the result -- NaN
    // under either rule.
    const float xt = fmaxf(b.x1, g.x1);                          // :141
    const float yt = fmaxf(b.y1, g.y1);                          // :142
    const float xb = fminf(b.x2, g.x2);                          // :143
    const float yb = fminf(b.y2, g.y2);                          // :144
    const float dx = xb - xt, dy = yb - yt;
    const float inter = fmaxf(dx, 0.0f) * fmaxf(dy, 0.0f);       // :146
    const float uni = b_area + g_area - inter;                   // :148
    return inter / uni;                                          // :150 (no epsilon)
}

__device__ __forceinline__ float box_area_plain(const Box &b) { return (b.y2 - b.y1) * (b.x2 - b.x1); }

// ---- TF CombinedNonMaxSuppression internals (utils/bbox_utils.py:66) ----
__device__ __forceinline__ float tf_min(float a, float b) { return b < a ? b : a; }   // std::min
__device__ __forceinline__ float tf_max(float a, float b) { return a < b ? b : a; }   // std::max

// canonical corners + area of one box, computed once per candidate
struct CBox {
    float ymin, xmin, ymax, xmax, area;
};

__device__ __forceinline__ CBox canonical(const Box &b)
{
    CBox c;
    c.ymin = tf_min(b.y1, b.y2);
    c.xmin = tf_min(b.x1, b.x2);
    c.ymax = tf_max(b.y1, b.y2);
    c.xmax = tf_max(b.x1, b.x2);
    c.area = (c.ymax - c.ymin) * (c.xmax - c.xmin);
    return c;
}

// IoU(candidate i, selected j) exactly as TF's NMS kernel evaluates it
__device__ __forceinline__ float nms_iou(const CBox &i, const CBox &j)
{
    if (i.area <= 0.0f || j.area <= 0.0f) return 0.0f;
    const float iymin = tf_max(i.ymin, j.ymin), ixmin = tf_max(i.xmin, j.xmin);
    const float iymax = tf_min(i.ymax, j.ymax), ixmax = tf_min(i.xmax, j.xmax);
    const float inter = tf_max(iymax - iymin, 0.0f) * tf_max(ixmax - ixmin, 0.0f);
    return inter / (i.area + j.area - inter);
}

// nms_iou(i, j) > thr, bit for bit, without the IEEE divide except within 2^-18 (relative) of the threshold:
// inter > thr_hi * uni  =>  fl(inter / uni) > thr;   inter < thr_lo * uni  =>  fl(inter / uni) < thr
// (thr_hi / thr_lo = thr * (1 +- 2^-18); one rounding of the product and one of the quotient are 2^-24 each).  The host
// passes thr_hi = +inf, thr_lo = -inf when thr is not a positive finite number: both shortcuts are then never taken.
__device__ __forceinline__ bool nms_suppresses(const CBox &i, const CBox &j, float thr, float thr_lo, float thr_hi)
{
    if (i.area <= 0.0f || j.area <= 0.0f) return 0.0f > thr;
    // v_max_f32 / v_min_f32 instead of TF's std::max / std::min (compare + select): identical on non-NaN operands; a NaN
    // corner makes that box's area NaN (canonical() keeps TF's operand order), so `uni` is NaN, both shortcuts below are
    // skipped and the quotient is NaN: "not suppressed" under either rule
    const float iymin = fmaxf(i.ymin, j.ymin), ixmin = fmaxf(i.xmin, j.xmin);
    const float iymax = fminf(i.ymax, j.ymax), ixmax = fminf(i.xmax, j.xmax);
    const float inter = fmaxf(iymax - iymin, 0.0f) * fmaxf(ixmax - ixmin, 0.0f);
    const float uni = i.area + j.area - inter;
    if (uni >= 1e-30f) {                       // products below stay normal
        if (inter > thr_hi * uni) return true;
        if (inter < thr_lo * uni) return false;
    }
    return inter / uni > thr;
}

// v_max_f32 / v_min_f32 as written: fmaxf / fminf on values that come from memory make the compiler put a canonicalising
// v_max_f32 v, v, v in front of every operand (a third of the VALU work of the test loops below).  The instruction itself
// already returns the other operand for a NaN one.
__device__ __forceinline__ float vmax_vv(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmin_vv(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// The same decision, branch-free, for the inner loops: `decided` is cleared when the quotient would be needed (the ratio
// within 2^-18 of the threshold, a tiny or NaN union, a threshold for which the host switched the shortcuts off) -- the
// caller then repeats its tests with nms_suppresses.  Boxes of area <= 0 must carry area = +inf here (nms_area_key): the
// union is then +inf and every path gives TF's answer for them, `0 > thr`, without a test of its own.
__device__ __forceinline__ bool nms_suppresses_fast(const CBox &i, const CBox &j, float thr_lo, float thr_hi, bool &decided)
{
    const float iymin = vmax_vv(i.ymin, j.ymin), ixmin = vmax_vv(i.xmin, j.xmin);
    const float iymax = vmin_vv(i.ymax, j.ymax), ixmax = vmin_vv(i.xmax, j.xmax);
    const float inter = fmaxf(iymax - iymin, 0.0f) * fmaxf(ixmax - ixmin, 0.0f);
    const float uni = i.area + j.area - inter;
    const bool hi = inter > thr_hi * uni, lo = inter < thr_lo * uni;
    decided = decided && (uni >= 1e-30f) && (hi || lo);
    return hi;
}
// The loop form of the same decision (what the NMS kernel's inner loops use).  Returns r = inter - thr * uni from ONE fused
// multiply-add: one rounding, so r has the sign of the exact difference and `r > 0` is `inter / uni > thr` in exact
// arithmetic.  `m` collects (v_min) the smallest |r| - eps * uni over the tests of a loop, eps = thr * 2^-18: if it ends
// above 1e-30, every exact ratio was further than 2^-18 (relative) from the threshold -- 2^5 times the rounding of the IEEE
// quotient TF compares -- and every `r > 0` is TF's answer; otherwise the caller repeats its tests with nms_suppresses.
// A NaN or -inf r (a box of area key +inf, NaN or infinite coordinates) is "not suppressed", which is TF's answer for
// those (0 > thr, or a NaN quotient); its margin is NaN, which v_min passes over.  Only for 1e-6 <= thr <= 1e6 (the host
// starts `m` at 0 otherwise).
__device__ __forceinline__ float nms_excess(const CBox &i, const CBox &j, float thr, float eps, float &m)
{
    const float iymin = vmax_vv(i.ymin, j.ymin), ixmin = vmax_vv(i.xmin, j.xmin);
    const float iymax = vmin_vv(i.ymax, j.ymax), ixmax = vmin_vv(i.xmax, j.xmax);
    const float inter = fmaxf(iymax - iymin, 0.0f) * fmaxf(ixmax - ixmin, 0.0f);
    const float uni = i.area + j.area - inter;
    const float r = __builtin_fmaf(-thr, uni, inter);
    m = vmin_vv(m, __builtin_fmaf(-eps, uni, __builtin_fabsf(r)));
    return r;
}
__device__ __forceinline__ float nms_area_key(float area) { return area <= 0.0f ? INFINITY : area; }

// monotone map float -> uint32 (descending float order == descending uint order);
// -0.0 is folded onto +0.0 so that equal scores tie exactly as a float compare would.
__device__ __forceinline__ unsigned orderable(float s)
{
    if (s == 0.0f) s = 0.0f;
    const unsigned u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

}  // namespace rpn
