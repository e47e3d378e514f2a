/*
 * rpn_oracle.c -- plain-C restatement of the RPN proposal path's box math.
 *
 * TEST INFRASTRUCTURE ONLY.  Used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker / reported CPU baseline; never
 * linked, loaded or called by the product (tf_rpn_amd/).
 *
 * PARITY UNPINNED: the reference has no tests or golden vectors and TensorFlow
 * cannot be run in the build environment (see oracle/__init__.py).
 *
 * Each function cites the reference lines (under /root/reference) it follows.
 * Build:  gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC   (oracle/Makefile)
 * -ffp-contract=off matters: the reference's eager ops round after every
 * multiply and add, so no FMA may be formed.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- anchors: utils/bbox_utils.py:3-46 ---------------------------------- */
/* ratios/scales are the python floats of hyper_params (doubles). */
void orc_generate_base_anchors(double img_size, const double *ratios, int nr,
                               const double *scales, int ns, float *base /* (ns*nr,4) */)
{
    int k = 0;
    for (int si = 0; si < ns; ++si) {
        double scale = scales[si] / img_size;                    /* :16 python double */
        for (int ri = 0; ri < nr; ++ri, ++k) {
            float w = sqrtf((float)(scale * scale / ratios[ri]));   /* :18 double -> f32 -> f32 sqrt */
            float h = w * (float)ratios[ri];                     /* :19 */
            base[4 * k + 0] = -h / 2.0f;                         /* :20 */
            base[4 * k + 1] = -w / 2.0f;
            base[4 * k + 2] = h / 2.0f;
            base[4 * k + 3] = w / 2.0f;
        }
    }
}

static float clip01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

void orc_generate_anchors(double img_size, int fm, const double *ratios, int nr,
                          const double *scales, int ns, float *anchors /* (fm*fm*ns*nr,4) */)
{
    int K = nr * ns;
    float *base = (float *)malloc(sizeof(float) * 4 * (size_t)K);
    float *grid = (float *)malloc(sizeof(float) * (size_t)fm);
    orc_generate_base_anchors(img_size, ratios, nr, scales, ns, base);
    double stride = 1.0 / (double)fm;                            /* :35 */
    for (int i = 0; i < fm; ++i)
        grid[i] = (float)((double)i / (double)fm + stride / 2.0);   /* :36 f64 then cast */
    for (int y = 0; y < fm; ++y)
        for (int x = 0; x < fm; ++x)                             /* :38-40 row-major, [y,x,y,x] */
            for (int k = 0; k < K; ++k) {
                float *a = anchors + 4 * (((size_t)y * fm + x) * K + k);
                a[0] = clip01(base[4 * k + 0] + grid[y]);        /* :44-46 */
                a[1] = clip01(base[4 * k + 1] + grid[x]);
                a[2] = clip01(base[4 * k + 2] + grid[y]);
                a[3] = clip01(base[4 * k + 3] + grid[x]);
            }
    free(base);
    free(grid);
}

/* ---- decode: predictor.py:55 + utils/bbox_utils.py:72-96 ----------------- */
/* variances may be NULL (no scaling).  anchors_batched: anchors are (B,A,4). */
void orc_decode(const float *anchors, int anchors_batched, const float *deltas,
                const float *variances, int B, int A, float *out)
{
    for (int b = 0; b < B; ++b)
        for (int a = 0; a < A; ++a) {
            const float *an = anchors + 4 * ((size_t)(anchors_batched ? b : 0) * A + a);
            const float *d = deltas + 4 * ((size_t)b * A + a);
            float *o = out + 4 * ((size_t)b * A + a);
            float dy = d[0], dx = d[1], dh = d[2], dw = d[3];
            if (variances) {                                     /* predictor.py:55 */
                dy = dy * variances[0]; dx = dx * variances[1];
                dh = dh * variances[2]; dw = dw * variances[3];
            }
            float w = an[3] - an[1];                             /* :81 */
            float h = an[2] - an[0];                             /* :82 */
            float cx = an[1] + 0.5f * w;                         /* :83 */
            float cy = an[0] + 0.5f * h;                         /* :84 */
            float bw = expf(dw) * w;                             /* :86 */
            float bh = expf(dh) * h;                             /* :87 */
            float bcx = (dx * w) + cx;                           /* :88 */
            float bcy = (dy * h) + cy;                           /* :89 */
            float y1 = bcy - (0.5f * bh);                        /* :91 */
            float x1 = bcx - (0.5f * bw);                        /* :92 */
            o[0] = y1; o[1] = x1;
            o[2] = bh + y1;                                      /* :93 */
            o[3] = bw + x1;                                      /* :94 */
        }
}

/* ---- encode: utils/bbox_utils.py:98-124 ---------------------------------- */
void orc_encode(const float *bboxes, int bboxes_batched, const float *gt, int B, int A, float *out)
{
    for (int b = 0; b < B; ++b)
        for (int a = 0; a < A; ++a) {
            const float *bb = bboxes + 4 * ((size_t)(bboxes_batched ? b : 0) * A + a);
            const float *g = gt + 4 * ((size_t)b * A + a);
            float *o = out + 4 * ((size_t)b * A + a);
            float bw = bb[3] - bb[1], bh = bb[2] - bb[0];
            float bcx = bb[1] + 0.5f * bw, bcy = bb[0] + 0.5f * bh;
            float gw = g[3] - g[1], gh = g[2] - g[0];
            float gcx = g[1] + 0.5f * gw, gcy = g[0] + 0.5f * gh;
            if (bw == 0.0f) bw = 1e-3f;                          /* :117 */
            if (bh == 0.0f) bh = 1e-3f;                          /* :118 */
            o[1] = (gw == 0.0f) ? 0.0f : (gcx - bcx) / bw;       /* :119 */
            o[0] = (gh == 0.0f) ? 0.0f : (gcy - bcy) / bh;       /* :120 */
            o[3] = (gw == 0.0f) ? 0.0f : logf(gw / bw);          /* :121 */
            o[2] = (gh == 0.0f) ? 0.0f : logf(gh / bh);          /* :122 */
        }
}

/* ---- normalize / denormalize: utils/bbox_utils.py:152-182 (tf.round = half to even = rintf) ---- */
void orc_scale_boxes(const float *in, long long n, float height, float width, int denorm, float *out)
{
    for (long long i = 0; i < n; ++i) {
        const float *b = in + 4 * i;
        float *o = out + 4 * i;
        if (denorm) {
            o[0] = rintf(b[0] * height); o[1] = rintf(b[1] * width);
            o[2] = rintf(b[2] * height); o[3] = rintf(b[3] * width);
        } else {
            o[0] = b[0] / height; o[1] = b[1] / width; o[2] = b[2] / height; o[3] = b[3] / width;
        }
    }
}

/* ---- IoU map: utils/bbox_utils.py:126-150 -------------------------------- */
static float fmaxx(float a, float b) { return a > b ? a : b; }   /* tf.maximum on finite data */
static float fminn(float a, float b) { return a < b ? a : b; }

void orc_iou_map(const float *bboxes, int bboxes_batched, int A, const float *gt, int B, int G,
                 float *out /* (B,A,G) */)
{
    for (int b = 0; b < B; ++b)
        for (int a = 0; a < A; ++a) {
            const float *bb = bboxes + 4 * ((size_t)(bboxes_batched ? b : 0) * A + a);
            float barea = (bb[2] - bb[0]) * (bb[3] - bb[1]);     /* :139 */
            for (int g = 0; g < G; ++g) {
                const float *gg = gt + 4 * ((size_t)b * G + g);
                float garea = (gg[2] - gg[0]) * (gg[3] - gg[1]); /* :138 */
                float xt = fmaxx(bb[1], gg[1]);                  /* :141 */
                float yt = fmaxx(bb[0], gg[0]);                  /* :142 */
                float xb = fminn(bb[3], gg[3]);                  /* :143 */
                float yb = fminn(bb[2], gg[2]);                  /* :144 */
                float inter = fmaxx(xb - xt, 0.0f) * fmaxx(yb - yt, 0.0f);   /* :146 */
                float uni = barea + garea - inter;               /* :148 */
                out[((size_t)b * A + a) * G + g] = inter / uni;  /* :150 */
            }
        }
}

/* ---- NMS: utils/bbox_utils.py:48-70 -> tf.image.combined_non_max_suppression
 *      (TF 2.0.0 kernel, un-vendored; semantics restated) -- */
static float mn(float a, float b) { return b < a ? b : a; }      /* std::min */
static float mx(float a, float b) { return a < b ? b : a; }      /* std::max */

float orc_nms_iou(const float *bi, const float *bj)
{
    float ymin_i = mn(bi[0], bi[2]), xmin_i = mn(bi[1], bi[3]);
    float ymax_i = mx(bi[0], bi[2]), xmax_i = mx(bi[1], bi[3]);
    float ymin_j = mn(bj[0], bj[2]), xmin_j = mn(bj[1], bj[3]);
    float ymax_j = mx(bj[0], bj[2]), xmax_j = mx(bj[1], bj[3]);
    float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.0f || area_j <= 0.0f) return 0.0f;
    float iymin = mx(ymin_i, ymin_j), ixmin = mx(xmin_i, xmin_j);
    float iymax = mn(ymax_i, ymax_j), ixmax = mn(xmax_i, xmax_j);
    float inter = mx(iymax - iymin, 0.0f) * mx(ixmax - ixmin, 0.0f);
    return inter / (area_i + area_j - inter);
}

typedef struct { float score; int idx; } cand_t;

static int cand_cmp(const void *pa, const void *pb)
{
    const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
    if (a->score > b->score) return -1;      /* descending score */
    if (a->score < b->score) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);   /* ties: lower index first */
}

typedef struct { float score; int cls; int rank; int idx; } ent_t;

static int ent_cmp(const void *pa, const void *pb)
{
    const ent_t *a = (const ent_t *)pa, *b = (const ent_t *)pb;
    if (a->score > b->score) return -1;
    if (a->score < b->score) return 1;
    if (a->cls != b->cls) return (a->cls > b->cls) - (a->cls < b->cls);
    return (a->rank > b->rank) - (a->rank < b->rank);
}

/* boxes (B,N,q,4), scores (B,N,C).  Outputs sized M = max_total (caller applies
 * the pad_per_class rule to max_total).  out_idx may be NULL. */
void orc_combined_nms(const float *boxes, const float *scores, int B, int N, int q, int C,
                      int max_per_class, int max_total, float iou_thr, float score_thr,
                      int clip_boxes, float *out_boxes, float *out_scores, float *out_classes,
                      int32_t *out_idx, int32_t *out_valid)
{
    cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * (size_t)(N > 0 ? N : 1));
    int *sel = (int *)malloc(sizeof(int) * (size_t)(max_per_class > 0 ? max_per_class : 1));
    ent_t *ents = (ent_t *)malloc(sizeof(ent_t) * (size_t)((size_t)C * (max_per_class > 0 ? max_per_class : 1)));
    for (int b = 0; b < B; ++b) {
        int ne = 0;
        for (int c = 0; c < C; ++c) {
            int qc = (q == 1) ? 0 : c;
            int nc = 0;
            for (int i = 0; i < N; ++i) {
                float s = scores[((size_t)b * N + i) * C + c];
                if (s > score_thr) { cand[nc].score = s; cand[nc].idx = i; ++nc; }
            }
            qsort(cand, (size_t)nc, sizeof(cand_t), cand_cmp);
            int ns = 0;
            for (int t = 0; t < nc && ns < max_per_class; ++t) {
                const float *bi = boxes + 4 * (((size_t)b * N + cand[t].idx) * q + qc);
                int keep = 1;
                for (int j = ns - 1; j >= 0; --j) {
                    const float *bj = boxes + 4 * (((size_t)b * N + sel[j]) * q + qc);
                    if (orc_nms_iou(bi, bj) > iou_thr) { keep = 0; break; }
                }
                if (keep) {
                    sel[ns] = cand[t].idx;
                    ents[ne].score = cand[t].score; ents[ne].cls = c; ents[ne].rank = ns; ents[ne].idx = cand[t].idx;
                    ++ns; ++ne;
                }
            }
        }
        qsort(ents, (size_t)ne, sizeof(ent_t), ent_cmp);
        if (ne > max_total) ne = max_total;
        out_valid[b] = ne;
        for (int r = 0; r < max_total; ++r) {
            float *ob = out_boxes + 4 * ((size_t)b * max_total + r);
            if (r < ne) {
                int qc = (q == 1) ? 0 : ents[r].cls;
                const float *bx = boxes + 4 * (((size_t)b * N + ents[r].idx) * q + qc);
                for (int k = 0; k < 4; ++k) ob[k] = clip_boxes ? clip01(bx[k]) : bx[k];
                out_scores[(size_t)b * max_total + r] = ents[r].score;
                out_classes[(size_t)b * max_total + r] = (float)ents[r].cls;
                if (out_idx) out_idx[(size_t)b * max_total + r] = ents[r].idx;
            } else {
                ob[0] = ob[1] = ob[2] = ob[3] = 0.0f;
                out_scores[(size_t)b * max_total + r] = 0.0f;
                out_classes[(size_t)b * max_total + r] = 0.0f;
                if (out_idx) out_idx[(size_t)b * max_total + r] = -1;
            }
        }
    }
    free(cand); free(sel); free(ents);
}

/* ---- conv stack pieces (Keras Conv2D / MaxPooling2D semantics; models/rpn_vgg16.py:16-20) ----
 * Direct NHWC x HWIO cross-correlation, double accumulation, for SMALL cross-checks only.
 * pad_t/pad_l are explicit zero pads (Keras 'same' 3x3 s1 -> 1; ZeroPadding2D + 'valid' for MNv2 s2).
 * act: 0 linear, 1 relu, 2 sigmoid, 3 relu6.  groups==Cin && Cout==Cin -> depthwise (w is (R,S,Cin,1)). */
void orc_conv2d(const float *in, int B, int H, int W, int Cin, const float *w, const float *bias,
                int R, int S, int Cout, int stride, int pad_t, int pad_l, int OH, int OW,
                int depthwise, int act, float *out)
{
    for (int b = 0; b < B; ++b)
        for (int oy = 0; oy < OH; ++oy)
            for (int ox = 0; ox < OW; ++ox)
                for (int n = 0; n < Cout; ++n) {
                    double acc = bias ? (double)bias[n] : 0.0;
                    for (int r = 0; r < R; ++r) {
                        int iy = oy * stride + r - pad_t;
                        if (iy < 0 || iy >= H) continue;
                        for (int s = 0; s < S; ++s) {
                            int ix = ox * stride + s - pad_l;
                            if (ix < 0 || ix >= W) continue;
                            const float *ip = in + (((size_t)b * H + iy) * W + ix) * Cin;
                            if (depthwise) {
                                acc += (double)ip[n] * (double)w[((size_t)r * S + s) * Cin + n];
                            } else {
                                const float *wp = w + (((size_t)r * S + s) * Cin) * Cout + n;
                                for (int c = 0; c < Cin; ++c) acc += (double)ip[c] * (double)wp[(size_t)c * Cout];
                            }
                        }
                    }
                    float v = (float)acc;
                    if (act == 1) v = v > 0.0f ? v : 0.0f;
                    else if (act == 2) v = 1.0f / (1.0f + expf(-v));
                    else if (act == 3) v = v < 0.0f ? 0.0f : (v > 6.0f ? 6.0f : v);
                    out[(((size_t)b * OH + oy) * OW + ox) * Cout + n] = v;
                }
}

void orc_maxpool2x2(const float *in, int B, int H, int W, int C, float *out /* (B,H/2,W/2,C) */)
{
    int OH = H / 2, OW = W / 2;                                  /* 'valid' floors */
    for (int b = 0; b < B; ++b)
        for (int oy = 0; oy < OH; ++oy)
            for (int ox = 0; ox < OW; ++ox)
                for (int c = 0; c < C; ++c) {
                    const float *p = in + (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + c;
                    float m = p[0];
                    m = fmaxx(m, p[C]);
                    m = fmaxx(m, p[(size_t)W * C]);
                    m = fmaxx(m, p[(size_t)W * C + C]);
                    out[(((size_t)b * OH + oy) * OW + ox) * C + c] = m;
                }
}
