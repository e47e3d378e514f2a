// nms_kernels.hip -- per-image greedy NMS (tf.image.combined_non_max_suppression semantics,
// utils/bbox_utils.py:48-70) as wavefront-level HIP for gfx950.  No MFMA: the work is
// compares, one IEEE divide per pair, LDS traffic and cross-lane ballots.
//
// One 1024-thread workgroup per (image, class):
//   1. keys   : 64-bit (orderable(score) << 32 | ~index) for every candidate with
//               score > score_threshold, 0 otherwise, staged in LDS;
//   2. sort   : bitonic sort of the LDS keys, descending  => score desc, lower index first;
//   3. greedy : the sorted candidates are consumed in chunks of 256.  For a chunk,
//               (A) all 16 waves test chunk candidates against the already selected boxes
//                   (selected boxes are broadcast LDS reads),
//               (B) all waves build the 256x256 "j is suppressed by i" bit matrix of the chunk,
//               (C) wave 0 walks the chunk in order with wave-uniform 64-bit live masks
//                   (__ballot / readfirstlane / ffs), appending to the selected list.
//               The loop stops as soon as max_output_size_per_class boxes are selected, so a
//               typical image touches 2-4 chunks, not all A candidates.
//   4. output : selected boxes gathered (and clipped) straight into the padded outputs.
// Per image the HBM traffic is 4*A (scores) + 16 B per visited candidate + the outputs; the
// kernel is latency-bound, the serial part is step (C): ~1 LDS round trip
// per selected box.
//
// DECODE variant: candidate boxes are decoded on the fly from (anchor, delta*variance)
// (utils/bbox_utils.py:72-96, predictor.py:55), so decoded boxes never exist in HBM.
//
// C > 1: each (image, class) block writes its selection to a staging area; a second
// kernel merges the classes (score desc, ties class asc then selection order).
// Compiled with -ffp-contract=off.
#include "bbox_core.h"
#include "rpn_common.h"

namespace rpn {

constexpr int kNmsThreads = 1024;
constexpr int kNmsMaxSort = 16384;     // candidates per (image, class) sortable in LDS
constexpr int kChunk = 256;
constexpr int kChunkWords = kChunk / 64;
constexpr int kKeysPerThread = kNmsMaxSort / kNmsThreads;
constexpr size_t kLdsLimit = 160 * 1024;

struct NmsArgs {
    const float *boxes;      // (B,N,q,4)            [plain]   | deltas (B,N,4) [decode]
    const float *scores;     // (B,N,C)
    const float *anchors;    // (N,4)                [decode only]
    float var[4];
    int var_enabled;
    int B, N, q, C;
    int n_sort;              // power of two >= N
    int max_sel;             // min(max_per_class, [C==1: max_total])
    int max_total;
    float iou_thr, score_thr;
    int clip;
    // final outputs (written directly when C == 1)
    float *out_boxes, *out_scores, *out_classes;
    int *out_idx, *out_valid;
    // per-class staging (C > 1): sel index (B,C,max_sel) and count (B,C)
    int *stage_idx;
    int *stage_cnt;
};

// Descending bitonic sort of n (power of two) 64-bit keys in LDS by the whole workgroup.
__device__ void bitonic_sort_desc(unsigned long long *keys, int n)
{
    const int tid = threadIdx.x;
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int p = tid; p < (n >> 1); p += kNmsThreads) {
                const int lo = 2 * p - (p & (j - 1));      // index with bit j clear
                const int hi = lo + j;
                const unsigned long long a = keys[lo], b = keys[hi];
                const bool desc = (lo & k) == 0;
                if ((a < b) == desc) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    }
}

template <bool DECODE>
__device__ __forceinline__ Box fetch_box(const NmsArgs &p, int b, int idx, int qc)
{
    if constexpr (DECODE) {
        const Box an = load_box(p.anchors + 4 * (size_t)idx);
        const float4 d = *reinterpret_cast<const float4 *>(p.boxes + 4 * ((size_t)b * p.N + idx));
        float dy = d.x, dx = d.y, dh = d.z, dw = d.w;
        if (p.var_enabled) {
            dy = dy * p.var[0];
            dx = dx * p.var[1];
            dh = dh * p.var[2];
            dw = dw * p.var[3];
        }
        return decode_box(an, dy, dx, dh, dw);
    } else {
        return load_box(p.boxes + 4 * (((size_t)b * p.N + idx) * p.q + qc));
    }
}

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// LDS carve-up (dynamic, 16-byte aligned):
//   [0, 8*n_sort)                 keys; after the sort the first 4*n_sort bytes hold `order`
//   sel_c   : max_sel * 5 floats  canonical selected boxes (SoA)
//   sel_idx : max_sel ints
//   cbox    : kChunk * 5 floats   canonical chunk boxes (SoA)
//   mask    : kChunk * kChunkWords u64
//   dead    : kChunk ints
//   ctl     : 4 ints  {ncand, nsel, pos, _}
__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

struct LdsLayout {
    size_t keys, sel_c, sel_idx, cbox, mask, dead, ctl, total;
};

__host__ __device__ inline LdsLayout lds_layout(int n_sort, int max_sel)
{
    LdsLayout l;
    size_t o = 0;
    l.keys = o;    o = align16(o + (size_t)8 * n_sort);
    l.sel_c = o;   o = align16(o + (size_t)20 * max_sel);
    l.sel_idx = o; o = align16(o + (size_t)4 * max_sel);
    l.cbox = o;    o = align16(o + (size_t)20 * kChunk);
    l.mask = o;    o = align16(o + (size_t)8 * kChunk * kChunkWords);
    l.dead = o;    o = align16(o + (size_t)4 * kChunk);
    l.ctl = o;     o = align16(o + 16);
    l.total = o;
    return l;
}

template <bool DECODE>
__global__ void __launch_bounds__(kNmsThreads)
nms_kernel(NmsArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LdsLayout L = lds_layout(p.n_sort, p.max_sel);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem + L.keys);
    unsigned *order = reinterpret_cast<unsigned *>(smem + L.keys);
    float *sel_c = reinterpret_cast<float *>(smem + L.sel_c);          // [5][max_sel]
    int *sel_idx = reinterpret_cast<int *>(smem + L.sel_idx);
    float *cbox = reinterpret_cast<float *>(smem + L.cbox);            // [5][kChunk]
    unsigned long long *mask = reinterpret_cast<unsigned long long *>(smem + L.mask);   // [kChunk][4]
    int *dead = reinterpret_cast<int *>(smem + L.dead);
    int *ctl = reinterpret_cast<int *>(smem + L.ctl);

    const int tid = threadIdx.x;
    const int b = blockIdx.x / p.C;
    const int c = blockIdx.x - b * p.C;
    const int qc = (p.q == 1) ? 0 : c;
    const int N = p.N, n_sort = p.n_sort, max_sel = p.max_sel;

    if (tid < 4) ctl[tid] = 0;
    __syncthreads();

    // ---- 1. keys -------------------------------------------------------------------
    {
        int my_cand = 0;
        for (int i = tid; i < n_sort; i += kNmsThreads) {
            unsigned long long key = 0ull;
            if (i < N) {
                const float s = p.scores[((size_t)b * N + i) * p.C + c];
                if (s > p.score_thr) {          // strict; NaN never qualifies
                    key = ((unsigned long long)orderable(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
                    ++my_cand;
                }
            }
            keys[i] = key;
        }
        // wave-level reduction, one LDS atomic per wave
        for (int off = 32; off > 0; off >>= 1) my_cand += __shfl_down(my_cand, off, 64);
        if ((tid & 63) == 0 && my_cand) atomicAdd(&ctl[0], my_cand);
    }
    __syncthreads();
    const int ncand = ctl[0];

    // ---- 2. sort, then compact the keys to 32-bit indices in place --------------------
    if (ncand > 0) bitonic_sort_desc(keys, n_sort);
    {
        unsigned idx_reg[kKeysPerThread];
#pragma unroll
        for (int j = 0; j < kKeysPerThread; ++j) {
            const int i = tid + j * kNmsThreads;
            idx_reg[j] = (i < n_sort) ? (0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull)) : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kKeysPerThread; ++j) {
            const int i = tid + j * kNmsThreads;
            if (i < n_sort) order[i] = idx_reg[j];
        }
    }
    __syncthreads();

    // ---- 3. greedy selection over chunks of the sorted candidates -----------------------
    while (true) {
        const int nsel = ctl[1];
        const int pos = ctl[2];
        if (nsel >= max_sel || pos >= ncand) break;
        const int T = min(kChunk, ncand - pos);

        // chunk boxes -> canonical form in LDS
        if (tid < kChunk) {
            dead[tid] = (tid < T) ? 0 : 1;
            if (tid < T) {
                const CBox cb = canonical(fetch_box<DECODE>(p, b, (int)order[pos + tid], qc));
                cbox[0 * kChunk + tid] = cb.ymin;
                cbox[1 * kChunk + tid] = cb.xmin;
                cbox[2 * kChunk + tid] = cb.ymax;
                cbox[3 * kChunk + tid] = cb.xmax;
                cbox[4 * kChunk + tid] = cb.area;
            }
        }
        __syncthreads();

        // (A) chunk candidate t vs selected boxes j = part, part+4, ...  (4 waves per part;
        //     every lane of a wave reads the same selected box: LDS broadcast)
        {
            const int t = tid & (kChunk - 1);
            const int part = tid >> 8;
            if (t < T && nsel > 0) {
                const CBox ci{cbox[0 * kChunk + t], cbox[1 * kChunk + t], cbox[2 * kChunk + t],
                              cbox[3 * kChunk + t], cbox[4 * kChunk + t]};
                bool hit = false;
                for (int j = part; j < nsel; j += 4) {
                    const CBox sj{sel_c[0 * max_sel + j], sel_c[1 * max_sel + j], sel_c[2 * max_sel + j],
                                  sel_c[3 * max_sel + j], sel_c[4 * max_sel + j]};
                    hit |= nms_iou(ci, sj) > p.iou_thr;
                }
                if (hit) dead[t] = 1;
            }
        }
        __syncthreads();

        // (B) intra-chunk suppression bits: mask[i][w] bit jj set iff candidate j = 64w+jj (j > i)
        //     would be suppressed by candidate i.  Rows of dead candidates are never read.
        {
            const int i = tid >> 2;
            const int w = tid & 3;
            unsigned long long bits = 0ull;
            if (i < T && !dead[i] && (w * 64 + 63) > i) {
                const CBox si{cbox[0 * kChunk + i], cbox[1 * kChunk + i], cbox[2 * kChunk + i],
                              cbox[3 * kChunk + i], cbox[4 * kChunk + i]};
                const int j0 = w * 64;
                for (int jj = 0; jj < 64; ++jj) {
                    const int j = j0 + jj;
                    if (j > i && j < T) {
                        const CBox cj{cbox[0 * kChunk + j], cbox[1 * kChunk + j], cbox[2 * kChunk + j],
                                      cbox[3 * kChunk + j], cbox[4 * kChunk + j]};
                        if (nms_iou(cj, si) > p.iou_thr) bits |= (1ull << jj);
                    }
                }
            }
            if (i < kChunk) mask[i * kChunkWords + w] = bits;
        }
        __syncthreads();

        // (C) serial walk by wave 0 with wave-uniform live masks
        if (tid < 64) {
            unsigned long long rem[kChunkWords];
#pragma unroll
            for (int w = 0; w < kChunkWords; ++w) rem[w] = __ballot(dead[w * 64 + tid] != 0);
            int cur = nsel;
#pragma unroll
            for (int w = 0; w < kChunkWords; ++w) {
                while (cur < max_sel) {
                    const unsigned long long avail = ~rem[w];
                    if (avail == 0ull) break;
                    const int bit = __ffsll((long long)avail) - 1;
                    const int i = w * 64 + bit;
                    if (tid < 5) sel_c[tid * max_sel + cur] = cbox[tid * kChunk + i];
                    if (tid == 5) sel_idx[cur] = (int)order[pos + i];
                    ++cur;
                    rem[w] |= (1ull << bit);
#pragma unroll
                    for (int w2 = 0; w2 < kChunkWords; ++w2)
                        if (w2 >= w) rem[w2] |= uniform64(mask[i * kChunkWords + w2]);
                }
            }
            if (tid == 0) {
                ctl[1] = cur;
                ctl[2] = pos + T;
            }
        }
        __syncthreads();
    }

    // ---- 4. outputs ------------------------------------------------------------------
    const int nsel = ctl[1];
    if (p.C == 1) {
        const int M = p.max_total;
        const int nvalid = min(nsel, M);
        for (int r = tid; r < M; r += kNmsThreads) {
            const size_t o = (size_t)b * M + r;
            if (r < nvalid) {
                const int idx = sel_idx[r];
                Box bx = fetch_box<DECODE>(p, b, idx, 0);
                if (p.clip) {
                    bx.y1 = clip01(bx.y1);
                    bx.x1 = clip01(bx.x1);
                    bx.y2 = clip01(bx.y2);
                    bx.x2 = clip01(bx.x2);
                }
                store_box(p.out_boxes + 4 * o, bx);
                p.out_scores[o] = p.scores[(size_t)b * N + idx];
                if (p.out_classes) p.out_classes[o] = 0.0f;
                if (p.out_idx) p.out_idx[o] = idx;
            } else {
                store_box(p.out_boxes + 4 * o, Box{0.0f, 0.0f, 0.0f, 0.0f});
                p.out_scores[o] = 0.0f;
                if (p.out_classes) p.out_classes[o] = 0.0f;
                if (p.out_idx) p.out_idx[o] = -1;
            }
        }
        if (tid == 0) p.out_valid[b] = nvalid;
    } else {
        int *si = p.stage_idx + ((size_t)b * p.C + c) * max_sel;
        for (int r = tid; r < nsel; r += kNmsThreads) si[r] = sel_idx[r];
        if (tid == 0) p.stage_cnt[(size_t)b * p.C + c] = nsel;
    }
}

// Merge of the per-class selections of one image (C > 1): entries sorted by score desc, ties
// class asc then selection order; top max_total written, rest zero-padded.
__global__ void __launch_bounds__(kNmsThreads)
nms_merge_kernel(NmsArgs p, int n_sort_merge)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int C = p.C, max_sel = p.max_sel, N = p.N, M = p.max_total;
    int *total_s = reinterpret_cast<int *>(smem + (size_t)8 * n_sort_merge);   // all LDS in the dynamic region
    if (tid == 0) *total_s = 0;
    __syncthreads();
    int mine = 0;
    for (int e = tid; e < n_sort_merge; e += kNmsThreads) {
        unsigned long long key = 0ull;
        if (e < C * max_sel) {
            const int c = e / max_sel, r = e - c * max_sel;
            if (r < p.stage_cnt[(size_t)b * C + c]) {
                const int idx = p.stage_idx[((size_t)b * C + c) * max_sel + r];
                const float s = p.scores[((size_t)b * N + idx) * C + c];
                key = ((unsigned long long)orderable(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)e);
                ++mine;
            }
        }
        keys[e] = key;
    }
    if (mine) atomicAdd(total_s, mine);
    __syncthreads();
    const int total = *total_s;
    if (total > 0) bitonic_sort_desc(keys, n_sort_merge);
    __syncthreads();
    const int nvalid = min(total, M);
    for (int r = tid; r < M; r += kNmsThreads) {
        const size_t o = (size_t)b * M + r;
        if (r < nvalid) {
            const int e = (int)(0xFFFFFFFFu - (unsigned)(keys[r] & 0xFFFFFFFFull));
            const int c = e / max_sel, rk = e - c * max_sel;
            const int idx = p.stage_idx[((size_t)b * C + c) * max_sel + rk];
            Box bx = load_box(p.boxes + 4 * (((size_t)b * N + idx) * p.q + (p.q == 1 ? 0 : c)));
            if (p.clip) {
                bx.y1 = clip01(bx.y1);
                bx.x1 = clip01(bx.x1);
                bx.y2 = clip01(bx.y2);
                bx.x2 = clip01(bx.x2);
            }
            store_box(p.out_boxes + 4 * o, bx);
            p.out_scores[o] = p.scores[((size_t)b * N + idx) * C + c];
            if (p.out_classes) p.out_classes[o] = (float)c;
            if (p.out_idx) p.out_idx[o] = idx;
        } else {
            store_box(p.out_boxes + 4 * o, Box{0.0f, 0.0f, 0.0f, 0.0f});
            p.out_scores[o] = 0.0f;
            if (p.out_classes) p.out_classes[o] = 0.0f;
            if (p.out_idx) p.out_idx[o] = -1;
        }
    }
    if (tid == 0) p.out_valid[b] = nvalid;
}

static int next_pow2(int v)
{
    int n = 2;
    while (n < v) n <<= 1;
    return n;
}

static size_t stage_bytes(int B, int C, int max_sel)
{
    if (C <= 1) return 0;
    return align16((size_t)B * C * max_sel * sizeof(int)) + align16((size_t)B * C * sizeof(int));
}

template <bool DECODE>
static int launch_nms(NmsArgs &p, void *d_workspace, size_t workspace_bytes, hipStream_t stream)
{
    const LdsLayout L = lds_layout(p.n_sort, p.max_sel);
    if (L.total > kLdsLimit)
        return fail(RPN_ERR_UNSUPPORTED, "nms: %zu bytes of LDS needed (N=%d, max per class=%d) > %zu",
                    L.total, p.N, p.max_sel, kLdsLimit);
    if (p.C > 1) {
        const size_t need = stage_bytes(p.B, p.C, p.max_sel);
        if (!d_workspace || workspace_bytes < need)
            return fail(RPN_ERR_WORKSPACE, "nms: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
        p.stage_idx = reinterpret_cast<int *>(d_workspace);
        p.stage_cnt = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(d_workspace) +
                                              align16((size_t)p.B * p.C * p.max_sel * sizeof(int)));
    }
    auto kern = nms_kernel<DECODE>;
    RPN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total));
    hipLaunchKernelGGL(kern, dim3(p.B * p.C), dim3(kNmsThreads), L.total, stream, p);
    RPN_CHECK_LAUNCH();
    if (p.C > 1) {
        const int n_merge = next_pow2(p.C * p.max_sel);
        const size_t lds = (size_t)8 * n_merge + 16;
        if (lds > kLdsLimit)
            return fail(RPN_ERR_UNSUPPORTED, "nms merge: C*max_per_class = %d too large", p.C * p.max_sel);
        RPN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(nms_merge_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(nms_merge_kernel, dim3(p.B), dim3(kNmsThreads), lds, stream, p, n_merge);
        RPN_CHECK_LAUNCH();
    }
    return RPN_OK;
}

}  // namespace rpn

using namespace rpn;

extern "C" size_t rpn_nms_workspace_bytes(int B, int N, int C, int max_per_class, int max_total)
{
    (void)N;
    (void)max_total;
    if (B <= 0 || C <= 1 || max_per_class <= 0) return 0;
    return stage_bytes(B, C, max_per_class);
}

extern "C" int rpn_combined_nms(const float *d_boxes, const float *d_scores, int B, int N, int q, int C,
                                int max_per_class, int max_total, float iou_threshold, float score_threshold,
                                int clip_boxes, float *d_out_boxes, float *d_out_scores, float *d_out_classes,
                                int32_t *d_out_idx, int32_t *d_out_valid, void *d_workspace,
                                size_t workspace_bytes, void *stream)
{
    RPN_REQUIRE(B >= 0 && N >= 0 && C >= 1, "rpn_combined_nms: bad sizes B=%d N=%d C=%d", B, N, C);
    RPN_REQUIRE(q == 1 || q == C, "rpn_combined_nms: q must be 1 or C (q=%d, C=%d)", q, C);
    RPN_REQUIRE(max_per_class >= 0 && max_total >= 0, "rpn_combined_nms: negative output size");
    if (B == 0 || max_total == 0) return RPN_OK;
    RPN_REQUIRE(d_out_boxes && d_out_scores && d_out_valid, "rpn_combined_nms: null output pointer");
    RPN_REQUIRE(N == 0 || (d_boxes && d_scores), "rpn_combined_nms: null input pointer");
    if (N > kNmsMaxSort)
        return fail(RPN_ERR_UNSUPPORTED, "rpn_combined_nms: N=%d > %d candidates per image not supported yet", N,
                    kNmsMaxSort);
    RPN_REQUIRE_DEVICE();
    NmsArgs p{};
    p.boxes = d_boxes;
    p.scores = d_scores;
    p.B = B; p.N = N; p.q = q; p.C = C;
    p.n_sort = next_pow2(N);
    p.max_sel = (C == 1) ? (max_per_class < max_total ? max_per_class : max_total) : max_per_class;
    if (p.max_sel < 1) p.max_sel = 1;
    if (max_per_class == 0) p.max_sel = 1;
    p.max_total = max_total;
    p.iou_thr = iou_threshold;
    p.score_thr = (max_per_class == 0) ? INFINITY : score_threshold;   // nothing can be selected
    p.clip = clip_boxes;
    p.out_boxes = d_out_boxes; p.out_scores = d_out_scores; p.out_classes = d_out_classes;
    p.out_idx = d_out_idx; p.out_valid = d_out_valid;
    return launch_nms<false>(p, d_workspace, workspace_bytes, as_stream(stream));
}

extern "C" int rpn_decode_nms(const float *d_anchors, const float *d_deltas, const float *variances,
                              const float *d_scores, int B, int A, int max_total, float iou_threshold,
                              float score_threshold, int clip_boxes, float *d_out_boxes, float *d_out_scores,
                              int32_t *d_out_idx, int32_t *d_out_valid, void *d_workspace,
                              size_t workspace_bytes, void *stream)
{
    RPN_REQUIRE(B >= 0 && A >= 0 && max_total >= 0, "rpn_decode_nms: negative size");
    if (B == 0 || max_total == 0) return RPN_OK;
    RPN_REQUIRE(d_out_boxes && d_out_scores && d_out_valid, "rpn_decode_nms: null output pointer");
    RPN_REQUIRE(A == 0 || (d_anchors && d_deltas && d_scores), "rpn_decode_nms: null input pointer");
    if (A > kNmsMaxSort)
        return fail(RPN_ERR_UNSUPPORTED, "rpn_decode_nms: A=%d > %d candidates per image not supported yet", A,
                    kNmsMaxSort);
    RPN_REQUIRE_DEVICE();
    NmsArgs p{};
    p.boxes = d_deltas;
    p.scores = d_scores;
    p.anchors = d_anchors;
    if (variances) {
        for (int i = 0; i < 4; ++i) p.var[i] = variances[i];
        p.var_enabled = 1;
    }
    p.B = B; p.N = A; p.q = 1; p.C = 1;
    p.n_sort = next_pow2(A);
    p.max_sel = max_total;
    p.max_total = max_total;
    p.iou_thr = iou_threshold;
    p.score_thr = score_threshold;
    p.clip = clip_boxes;
    p.out_boxes = d_out_boxes; p.out_scores = d_out_scores; p.out_classes = nullptr;
    p.out_idx = d_out_idx; p.out_valid = d_out_valid;
    return launch_nms<true>(p, d_workspace, workspace_bytes, as_stream(stream));
}
