// nms_kernels.hip -- per-image greedy NMS (tf.image.combined_non_max_suppression semantics,
// utils/bbox_utils.py:48-70) as wavefront-level HIP for gfx950.  No MFMA: the work is compares, one IEEE
// divide per pair, LDS traffic and cross-lane ballots / readlanes.
//
// One 1024-thread workgroup per (image, class).  Candidates are ordered by the 64-bit key
// (orderable(score) << 32 | ~index): descending key == descending score, ties lower index first.
//   1. band   : instead of sorting all N keys, a radix select (11-bit digits, LDS histogram, keys recomputed
//               from the L2-resident scores on every pass) finds the threshold that isolates the next
//               <= 4096 best candidates; only that band is compacted into LDS and bitonic-sorted.  If the
//               band is exhausted before max_output_size boxes are selected (rare), the next band is taken.
//               Any N works (no LDS-capacity limit on the number of candidates).
//   2. greedy : the sorted band is consumed in chunks of 256.  For a chunk,
//               (A) all 16 waves test chunk candidates against the already selected boxes (broadcast reads),
//               (B) all waves build the 256x256 "j is suppressed by i" bit matrix for live (i, j) only,
//               (C) wave 0 walks the chunk in order: lanes 0-3 hold the four 64-bit live masks, one LDS round
//                   trip (the selected row of the bit matrix) per selected box; the selected boxes are copied
//                   to the selected list afterwards, in parallel.
//               The loop stops as soon as max_output_size_per_class boxes are selected.
//   3. output : selected boxes gathered (and clipped) straight into the padded outputs.
// Per image the HBM traffic is 4*A (scores) + 16 B per visited candidate + the outputs; the kernel is
// latency-bound: the serial part is step (C).
//
// DECODE variant: candidate boxes are decoded on the fly from (anchor, delta*variance)
// (utils/bbox_utils.py:72-96, predictor.py:55), so decoded boxes never exist in HBM.
//
// C > 1: each (image, class) block writes its selection to a staging area; a second kernel merges the
// classes (score desc, ties class asc then selection order).
// Compiled with -ffp-contract=off.
#include "bbox_core.h"
#include "radix_select.h"
#include "rpn_common.h"

#include <cstdlib>
#include <mutex>

namespace rpn {

// The thread index as a value the optimiser cannot see through.  Every phase of the NMS kernel maps threads to work in its
// own way (tid / 16, cslot(tid), 2047 - 2 tid, ...); with the plain threadIdx.x all of those are loop-invariant, get
// hoisted to the top of the kernel and stay live across the band loop: 46 spilled VGPRs = 180 B of scratch per thread,
// 11.8 MB written per launch at 64 images (round-2 profile).  Taken fresh at the head of a phase, the derived values
// die with the phase: no scratch at the same 128-VGPR budget.
__device__ __forceinline__ int fresh_tid()
{
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// A workgroup-uniform 64-bit value that was read from LDS, moved to scalar registers (two v_readfirstlane): it then costs
// no vector registers across the band loop.
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

#ifdef RPN_NMS_STAMP
// Debug build only (-DRPN_NMS_STAMP, scripts/nms_stamp_probe.py): cycle stamps of thread 0 at the phase boundaries of the first 64
// workgroups, in program order.  The stamp index lives in (static) LDS so that the band functions can stamp too.
__device__ unsigned long long g_nms_stamps[64 * 512];
__device__ __forceinline__ int &nms_stamp_idx() { __shared__ int idx; return idx; }
#define NMS_STAMP(code)                                                                                       \
    do {                                                                                                      \
        if (threadIdx.x == 0 && blockIdx.x < 64 && nms_stamp_idx() < 512)                                     \
            g_nms_stamps[blockIdx.x * 512 + nms_stamp_idx()++] = ((unsigned long long)(code) << 56) | (__builtin_readcyclecounter() & 0x00FFFFFFFFFFFFFFull); \
    } while (0)
extern "C" int rpn_debug_read_nms_stamps(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nms_stamps), (size_t)n * 8);
}
#else
#define NMS_STAMP(code) ((void)0)
#endif

constexpr int kNmsThreads = 1024;
constexpr int kBandCap = 4096;         // candidates sorted per band (LDS: 32 KB of keys)
constexpr int kBandTarget = 3072;      // the radix select aims at this many, accepts up to kBandCap
constexpr int kBins = kRsHistWords;    // radix-select histogram words (radix_select.h)
#ifndef RPN_NMS_CHUNK
#define RPN_NMS_CHUNK 256
#endif
constexpr int kChunk = RPN_NMS_CHUNK;
constexpr int kChunkWords = kChunk / 64;
// Chunk boxes are stored at slot(j) = (j % 16) * 16 + j / 16: the threads that build one suppression row read 16
// candidates 16 apart (j = 16 * piece + jj, piece = lane % 16) -- consecutive slots, conflict-free; in natural order
// those reads are 8-way bank conflicts (measured: 8 us per 64-row group).
// LDS slot of chunk candidate j = 16 * piece + jj: row jj of a 16 x 16 transpose, the piece XOR-swizzled by the row's upper
// two bits.  Threads that differ in the piece (step B2) and threads that cover a group's 4 pieces x 4 rows (step B1) both read
// 16 different 16-byte columns.
constexpr int kPieceBits = kChunk == 512 ? 5 : 4;                    // log2 of the chunk's 16-candidate pieces
static_assert((16 << kPieceBits) == kChunk, "chunk size");
__device__ __forceinline__ int cslot2(int jj, int piece) { return (jj << kPieceBits) | (piece ^ (jj & 12)); }
__device__ __forceinline__ int cslot(int j) { return cslot2(j & 15, j >> 4); }
constexpr size_t kLdsLimit = 160 * 1024;
constexpr int kLoadBatch = 9;          // score loads in flight per thread in the passes over all N scores (9 x 1024 threads cover the 8649 / 9216 anchors of one 500 x 500 image in one round)
constexpr int kScoreCacheMax = 16384;  // scores per (image, class) pair that the LDS cache may hold          // score loads in flight per thread in the passes over all N scores

struct NmsArgs {
    const float *boxes;      // (B,N,q,4)            [plain]   | deltas (B,N,4) [decode]
    const float *scores;     // (B,N,C)
    const float *anchors;    // (N,4)                [decode only]
    float var[4];
    int var_enabled;
    int B, N, q, C;
    int max_sel;             // min(max_per_class, [C==1: max_total])
    int max_total;
    float iou_thr, score_thr;
    float iou_lo, iou_hi;    // iou_thr * (1 -+ 2^-18) (nms_suppresses); -inf / +inf when iou_thr is not positive and finite
    float iou_eps, m0;       // nms_excess: iou_thr * 2^-18 and the start of its margin (+inf; 0 = always take the exact tests)
    int clip;
    // final outputs (written directly when C == 1)
    float *out_boxes, *out_scores, *out_classes;
    int *out_idx, *out_valid;
    // per-class staging (C > 1): sel index (B,C,max_sel) and count (B,C)
    int *stage_idx;
    int *stage_cnt;
#ifdef RPN_LAB
    int stop_after;          // timing experiments only (RPN_NMS_STOP): 2 = stop after the first band sort; 0 = run all
#endif
    int linear_select;       // band threshold by the one-pass linear histogram first (RPN_NMS_LINEAR=0: radix select only)
    // cluster mode (few images, many candidates): `cluster` workgroups per (image, class) share the passes over the scores
    // of the FIRST band; cl_ctl / cl_band: the cluster workspace (cluster_layout), zeroed control part
    int cache_n;             // scores of a pair cached in LDS after the first pass (0: none; >= N when granted)
    int cluster, cl_region;
    unsigned *cl_ctl;
    unsigned long long *cl_band;
    // area pruning of the tests against the boxes selected in earlier chunks (prune_bin): 0 = off
    int band1_mult;          // first band = band1_mult x the boxes wanted: 5 (round 3: 3).  On a conv head's outputs -- smooth score
                             // fields, heavily overlapping boxes: ~2 900 candidates visited for 300 boxes on the bench model's --
                             // the walk then finishes inside the first band instead of paying a second select + order pass (83 ->
                             // 73 us at batch 8, IoU 0.7); on configs[2]'s spread-out permutation scores the larger band costs 1 %
    int band1_sat;           // ... but band1_sat x (3) on a SATURATED head (band_select_linear: >= 128 scores in the top bin --
                             // MobileNetV2's random-init heads: 330 scores >= 0.9995 of 9 216): ordering a large band of equal scores
                             // costs more than a second band (one 500 x 500 image 61.9 -> 64.3 us, configs[4] in cluster mode 56.8 ->
                             // 97.6 us with 5 x); the cluster path always takes 3 x
    int prune;
    float prune_lo, prune_hi;   // a pair can only pass the IoU test if area_a * prune_lo <= area_b <= area_a * prune_hi
    // chunk compaction (round 5): step (A) for the WHOLE chunk first, the survivors moved to the front of the chunk, the group
    // loop (B1 / walk / B2) over the survivors only.  With a low threshold nine candidates in ten die in (A): one group per chunk
    // instead of four (two barrier intervals each).  configs[2], IoU 0.5: 126 -> 102 us, 0.7: 36.0 -> 34.5; on the bench model's
    // head outputs 0.5: 393 -> 300 us, 0.7: 76 -> 66 (smooth-score "model-like" inputs at 0.7: 40.1 -> 42.0, the one loss).
    // 512-candidate chunks with it: slower (0.7: 41 us, 0.5: 112 us).
    int compact;             // 0: off; n: for chunks that start with at least n boxes selected
};

// ---- area pruning ------------------------------------------------------------------------------------------------
// IoU(a, b) <= min(area) / max(area), in float32 arithmetic as well: the intersection's height (width) is the rounded
// difference of a min and a max of the corners, so it is at most either box's own rounded height (width) -- rounding is
// monotone -- and inter = fl(ih * iw) <= min(area_a, area_b); uni = fl(fl(area_a + area_b) - inter) >= max(area) * (1 - 3 *
// 2^-24); so TF's quotient fl(inter / uni) <= min / max * (1 + 2^-21).  A pair with min(area) < thr * (1 - 2^-19) * max(area)
// is therefore "not suppressed" without being tested.  The selected boxes are kept a second time ORDERED BY AREA BIN
// (quarter octaves: the float's exponent and two mantissa bits, 64 bins from 2^-14 up; everything smaller shares bin 0,
// everything larger -- and the +inf key of boxes without area, and NaN -- bin 63), with the bins' start offsets; a candidate
// of area a tests the contiguous run of bins [bin(a * prune_lo), bin(a * prune_hi)] only.  bin() is monotone and the two
// products are rounded by 2^-24 against 2^-19 of slack in the factors, so every selected box within the ratio is in the run.
// RPN proposals come from anchors of several scales: at IoU 0.7 a candidate meets about a third of the selected list.
constexpr int kPruneBins = 64;
constexpr int kPruneBase = (127 - 14) * 4;
__device__ __forceinline__ int prune_bin(float area)
{
    const int k = (int)(__float_as_uint(area) >> 21) - kPruneBase;     // (a NaN with the sign bit set: > 63 as well)
    return min(max(k, 0), kPruneBins - 1);
}

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

// Descending bitonic sort of 1024 64-bit keys, one per thread (pad with 0): compare-exchange distances below 64 stay
// inside a wave (two 32-bit shuffles, no barrier), only the 10 steps with distance >= 64 go through LDS.  The generic
// network above spends 0.4 us per step on its workgroup barrier (55 steps for 1024 keys).  Returns the key that ends up
// at position tid; `keys` (1024 entries of LDS) is scratch.
__device__ __forceinline__ unsigned long long bitonic1024_desc(unsigned long long mine, unsigned long long *keys)
{
    const int tid = threadIdx.x;
    for (int k = 2; k <= 1024; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            unsigned long long other;
            if (j >= 64) {
                keys[tid] = mine;
                __syncthreads();
                other = keys[tid ^ j];
                __syncthreads();
            } else {
                const unsigned lo = __shfl_xor((unsigned)mine, j, 64);
                const unsigned hi = __shfl_xor((unsigned)(mine >> 32), j, 64);
                other = ((unsigned long long)hi << 32) | lo;
            }
            const bool lower = (tid & j) == 0;                 // this thread holds the lower index of the pair
            const bool desc = (tid & k) == 0;                  // direction of this pair's block (k = 1024: always descending)
            const bool keep_max = lower == desc;
            mine = keep_max ? (mine > other ? mine : other) : (mine < other ? mine : other);
        }
    }
    return mine;
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

// 64-bit candidate key of box i (0 = not a candidate: score <= threshold or NaN)
__device__ __forceinline__ unsigned long long make_key(float s, float thr, int i)
{
    if (!(s > thr)) return 0ull;
    return ((unsigned long long)orderable(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
}

// Band threshold in ONE pass for scores that live in [0, 1) (objectness after a sigmoid): a LINEAR histogram of
// floor(score * 2048).  The generic radix select takes its 11-bit digits from the top of the float's bit pattern --
// sign, exponent, two mantissa bits -- so scores in (0, 1) fall into a handful of bins and it needs three passes over
// the keys (38 k of the kernel's 165 k cycles at 8649 anchors; each pass with its LDS atomics serialised on those
// few bins).  Linear bins spread such scores evenly: the bin where the count from the top reaches `want` gives the
// threshold score d / 2048 (exact in float; score >= d / 2048 <=> floor(score * 2048) >= d), i.e. the key threshold
// orderable(d / 2048) << 32.  Returns false -- caller falls back to the radix select -- when that bin holds more than
// the band may take, or is bin 0 (which also collects the scores below 0).  hist: 2048 words; ctl: 4 ints.
// All threads: exclusive suffix sums of the 2048 counts, top bin first (sfx[b] = keys in the bins above b; thread t owns bins
// 2047 - 2t and 2046 - 2t: a wave scan + 16 wave totals), and the bin where the count from the top reaches `want` ->
// ctl[0] (-1: no keys at all, -2: fewer than `want` in all), ctl[1] = keys in the bins >= it (or the total).  `scan`: 17
// words of LDS.  The caller puts a barrier behind it.  band_order_linear takes its segment offsets from the same `sfx`.
// (First version: wave 0 alone, lane l summing bins 2047 - 32 l .. serially -- 64 lanes on one LDS bank -- ~4 k cycles.)
__device__ __forceinline__ void hist_suffix_find(const unsigned *h, unsigned *sfx, int want, int *ctl, unsigned *scan)
{
    const int tid = fresh_tid(), lane = tid & 63, wave = tid >> 6;
    const int b0 = 2047 - 2 * tid, b1 = 2046 - 2 * tid;
    const unsigned c0 = h[b0], c1 = h[b1];
    unsigned incl = c0 + c1;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) scan[wave] = incl;
    __syncthreads();
    unsigned wpre = 0u, total = 0u;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const unsigned v = scan[w];
        total += v;
        if (w < wave) wpre += v;
    }
    const unsigned excl = wpre + incl - (c0 + c1);
    sfx[b0] = excl;
    sfx[b1] = excl + c0;
    if (total == 0u) {
        if (tid == 0) { ctl[0] = -1; ctl[1] = 0; }
    } else if ((int)total < want) {         // fewer than `want` keys in all: take everything
        if (tid == 0) { ctl[0] = -2; ctl[1] = (int)total; }
    } else if ((int)excl < want && (int)(excl + c0) >= want) {
        ctl[0] = b0;
        ctl[1] = (int)(excl + c0);          // keys in bins >= b0
    } else if ((int)(excl + c0) < want && (int)(excl + c0 + c1) >= want) {
        ctl[0] = b1;
        ctl[1] = (int)(excl + c0 + c1);
    }
}

// The scores of one (image, class) pair as the passes read them: from global memory (stride C), or -- once the first
// pass over all of them has copied them there -- from an LDS cache (p.cache_n scores; the host grants it when they fit:
// N <= 16 384 beside the kernel's other LDS).  A pass by one workgroup runs at one compute unit's memory-level parallelism
// (~7 B / clk: 5 k cycles for 8649 scores, NOTES.md); every pass after the first then costs LDS latency instead.
// Global reads go through a raw buffer descriptor (4 scalar registers) with a 32-bit byte offset: a plain pointer costs
// each read loop a loop-invariant 64-bit address pair per thread, which the register allocator parks in scratch.
struct ScoreSrc {
    __amdgpu_buffer_rsrc_t rs;   // the pair's scores: N floats at a stride of C
    int stride_bytes;            // 4 * C
    float *cache;                // null: no cache
    bool cached;                 // the cache has been filled
    __device__ __forceinline__ float operator()(int i) const
    {
        return cached ? cache[i] : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, i * stride_bytes, 0, 0));
    }
};
__device__ __forceinline__ ScoreSrc make_score_src(const float *g, int N, int C, float *cache)
{
    const long long bytes = (long long)N * C * 4;
    return ScoreSrc{__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g), (short)0, (int)(bytes > 0x7fffffffll ? 0x7fffffff : bytes), 0x00020000),
                    4 * C, cache, false};
}

__device__ __forceinline__ int linear_bin(float s) { return (int)fminf(fmaxf(s * 2048.0f, 0.0f), 2047.0f); }
// sub-bin of a score inside bin d: floor((score * 2048 - d) * 2048), exact (the difference of two floats in [d, d + 1) and a
// power-of-two scale); the threshold score of sub-bin k is (2048 d + k) / 2^22
__device__ __forceinline__ int linear_sub_bin(float s, int d) { return (int)fminf(fmaxf((s * 2048.0f - (float)d) * 2048.0f, 0.0f), 2047.0f); }

__device__ inline bool band_select_linear(ScoreSrc &sc, int N, float score_thr,
                                          unsigned long long hi_bound, int want, int cap, unsigned *hist, int *ctl,
                                          unsigned long long *thr_out, int *count, int *bin_out, int want_sat = 0, int cap_sat = 0)
{
    constexpr int NB = 2048;
    const int tid = fresh_tid();
    NMS_STAMP(33);
    // hist[0, 2048): the bins; hist[2048, 4096): the 2048 sub-bins of the TOP bin, filled in the same pass -- a saturated
    // sigmoid puts most of a head's 61 440 scores there, and the refinement below then needs no pass of its own
    for (int i = tid; i < 2 * NB; i += kNmsThreads) hist[i] = 0u;
    const bool fill = sc.cache != nullptr && !sc.cached;          // this pass also copies the scores into the LDS cache
    NMS_STAMP(34);
    __syncthreads();
    NMS_STAMP(30);
    for (int base = tid; base < N; base += kLoadBatch * kNmsThreads) {
        float sb[kLoadBatch];
#pragma unroll
        for (int u = 0; u < kLoadBatch; ++u) sb[u] = (base + u * kNmsThreads < N) ? sc(base + u * kNmsThreads) : NAN;
        if (fill) {
#pragma unroll
            for (int u = 0; u < kLoadBatch; ++u)
                if (base + u * kNmsThreads < N) sc.cache[base + u * kNmsThreads] = sb[u];
        }
#pragma unroll
        for (int u = 0; u < kLoadBatch; ++u) {
            const unsigned long long key = make_key(sb[u], score_thr, base + u * kNmsThreads);     // 0 for NaN / below threshold
            if (key != 0ull && key < hi_bound) {
                const int bin = linear_bin(sb[u]);
                atomicAdd(&hist[bin], 1u);
                if (bin == NB - 1) atomicAdd(&hist[NB + linear_sub_bin(sb[u], NB - 1)], 1u);
            }
        }
    }
    NMS_STAMP(35);
    __syncthreads();
    NMS_STAMP(31);
    if (fill) sc.cached = true;
    // The alternative band size for a SATURATED head -- the top bin (scores >= 2047 / 2048) alone holds 128 candidates or more:
    // hundreds of equal or nearly equal scores make a large band expensive to order (ranked by counting inside a sub-bin), so such
    // heads keep the small first band.  (workgroup-uniform: every thread reads the same word)
    if (want_sat > 0 && hist[NB - 1] >= 128u) {
        want = want_sat;
        cap = cap_sat;
    }
    hist_suffix_find(hist, hist + 4096, want, ctl, reinterpret_cast<unsigned *>(ctl) + 12);
    __syncthreads();
    const int d = ctl[0], n = ctl[1];
    const int in_d = d > 0 ? (int)hist[d] : 0;
    __syncthreads();                        // (ctl is reused below and by the caller)
    NMS_STAMP(32);
    *bin_out = d;
    if (d == -1) { *thr_out = 0ull; *count = 0; return true; }
    if (d == -2) { *thr_out = 1ull; *count = n; *bin_out = -1; return true; }       // (bin 0 included: no histogram order)
    if (d == 0) return false;
    if (n <= cap) {
        *thr_out = (unsigned long long)orderable((float)d * (1.0f / 2048.0f)) << 32;
        *count = n;
        return true;
    }
    // The crossing bin holds more than the band may take (scores bunched in a narrow range -- a head's sigmoid outputs at
    // 61 440 anchors put hundreds of keys into one bin): one more pass refines INSIDE that bin with 2048 sub-bins,
    // sub = floor((score * 2048 - d) * 2048) (exact: the difference of two floats in [d, d + 1) and a power-of-two scale),
    // instead of handing the whole problem to the three-pass radix select.  Threshold score (2048 d + sub) / 2^22.
    const int above = n - in_d;             // keys in the bins above d: all in the band
    const unsigned *sub = hist + NB;        // the top bin's sub-bins are there already
    if (d != NB - 1) {
        for (int i = tid; i < NB; i += kNmsThreads) hist[i] = 0u;
        __syncthreads();
        for (int base = tid; base < N; base += kLoadBatch * kNmsThreads) {
            float sb[kLoadBatch];
#pragma unroll
            for (int u = 0; u < kLoadBatch; ++u) sb[u] = (base + u * kNmsThreads < N) ? sc(base + u * kNmsThreads) : NAN;
#pragma unroll
            for (int u = 0; u < kLoadBatch; ++u) {
                const unsigned long long key = make_key(sb[u], score_thr, base + u * kNmsThreads);
                if (key != 0ull && key < hi_bound && linear_bin(sb[u]) == d) atomicAdd(&hist[linear_sub_bin(sb[u], d)], 1u);
            }
        }
        __syncthreads();
        sub = hist;
    }
    hist_suffix_find(sub, sub == hist ? hist + 4096 : hist + 6144, want - above, ctl, reinterpret_cast<unsigned *>(ctl) + 12);
    __syncthreads();
    const int d2 = ctl[0], n2 = ctl[1];
    __syncthreads();
    // the band can be ordered from a histogram only when it is the top bin's sub-bins >= d2 (no bins above it): 4096 + d2
    *bin_out = (d == NB - 1 && d2 >= 0) ? 4096 + d2 : -1;
    // (bin 2047 and its sub-bin 2047 also collect the scores >= 1 -- a saturated sigmoid: still exactly {score >= threshold})
    if (d2 < 0 || above + n2 > cap) return false;
    *thr_out = (unsigned long long)orderable((float)(d * 2048 + d2) * (1.0f / 4194304.0f)) << 32;
    *count = above + n2;
    return true;
}

__device__ __forceinline__ float key_score(unsigned long long key)            // inverse of orderable()
{
    const unsigned o = (unsigned)(key >> 32);
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

// the band's keys by a scan of all N scores (8 loads in flight per thread)
template <class Fn>
__device__ __forceinline__ void scan_band_keys(const ScoreSrc &sc, int N, float score_thr, unsigned long long thr,
                                               unsigned long long hi_bound, Fn fn)
{
    const int tid = fresh_tid();
    for (int base = tid; base < N; base += 8 * kNmsThreads) {
        float sb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) sb[u] = (base + u * kNmsThreads < N) ? sc(base + u * kNmsThreads) : NAN;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned long long key = make_key(sb[u], score_thr, base + u * kNmsThreads);
            if (key != 0ull && key >= thr && key < hi_bound) fn(key);
        }
    }
}

// The band in descending key order WITHOUT a sorting network, from the linear histogram that chose it: bin b's keys go
// to positions [off(b), off(b) + count(b)), off(b) = keys in the bins above b (a suffix scan over the 2048 counts); inside
// a bin (4 keys on average at 8649 anchors) a key's place is the number of larger keys in its bin.  One pass over the
// scores scatters the band's keys to their bins' segments (per-bin LDS cursors), a second step ranks inside the segments.
// hist: [0,2048) counts (from band_select_linear), [4096,6144) offsets (hist_suffix_find), [6144,8192) cursors.
// Returns false (nothing written) when a bin of the band holds more than 32 keys: the caller then sorts the old way.
// `for_each_band_key(fn)`: calls fn(key) for this thread's share of the band's keys (every key exactly once over the
// workgroup): a scan of all N scores, or -- cluster mode -- the list the cluster's ranks compacted into the workspace.
// Two levels: hist[0, 2048) are the linear bins' counts, hist[2048, 4096) the counts of the TOP bin's 2048 sub-bins (filled in the same
// pass); hist[4096, 6144) / [6144, 8192) their exclusive suffix sums (hist_suffix_find).  A key's segment index is its linear bin, or --
// `use_sub`: the top bin holds more than 32 keys, a saturated head -- 2048 + its sub-bin when it lies in the top bin (whose keys come
// first).  The band = the main bins >= d_main (without the top bin when use_sub) + the sub-bins >= d_sub.  Cursors: the upper half of
// the `band` area (n <= 2048 keys use the lower half).
template <class ForEachKey>
__device__ inline bool band_order_linear(ForEachKey for_each_band_key, int d_main, int d_sub, bool use_sub, int n, unsigned *hist,
                                         unsigned *flag, unsigned long long *band, unsigned *order, unsigned long long *low_out)
{
    const int tid = fresh_tid();
    const unsigned *counts = hist, *boff = hist + 4096;
    unsigned *cur = reinterpret_cast<unsigned *>(band + 2048);
    auto seg_of = [&](unsigned long long key) {
        const float sc = key_score(key);
        const int mb = linear_bin(sc);
        return (use_sub && mb == 2047) ? 2048 + linear_sub_bin(sc, 2047) : mb;
    };
    const int b0 = 2047 - 2 * tid, b1 = 2046 - 2 * tid;
    if (tid == 0) *flag = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[tid + i * kNmsThreads] = 0u;
    __syncthreads();
    // Keys are ranked inside their segment by counting the larger ones: a loop of the segment's length per key.  Up to kSegMax
    // keys that is still cheaper than compaction + the 1024-key network (~31 k cycles) -- and it is what a saturated head needs:
    // hundreds of scores that are EXACTLY 1.0 share one sub-bin whatever the binning, and are ordered by their index alone.
    constexpr unsigned kSegMax = 768u;
    bool fat = (b0 >= d_main && !(use_sub && b0 == 2047) && counts[b0] > kSegMax) || (b1 >= d_main && counts[b1] > kSegMax);
    if (use_sub) fat = fat || (b0 >= d_sub && counts[2048 + b0] > kSegMax) || (b1 >= d_sub && counts[2048 + b1] > kSegMax);
    if (fat) atomicOr(flag, 1u);
    __syncthreads();
    if (*flag != 0u) return false;
    NMS_STAMP(40);
    // scatter: the band's keys to their segments
    for_each_band_key([&](unsigned long long key) {
        const int b = seg_of(key);
        band[boff[b] + atomicAdd(&cur[b], 1u)] = key;
    });
    NMS_STAMP(43);
    __syncthreads();
    NMS_STAMP(41);
    // rank inside the segment (keys are unique); up to two keys per thread (n <= 2048)
    unsigned long long key[2] = {0ull, 0ull};
    int pos[2] = {-1, -1};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + u * kNmsThreads;
        if (i < n) {
            key[u] = band[i];
            const int b = seg_of(key[u]);
            const int seg0 = (int)boff[b], cnt = (int)counts[b];
            // (four reads in flight per trip: a saturated head's segments are hundreds of keys long -- bins next to 1.0 -- and one
            // dependent LDS read per key made this loop 15 k of configs[0]'s 26 k ordering cycles)
            int rank = 0, q = 0;
            for (; q + 4 <= cnt; q += 4) {
                const unsigned long long k0 = band[seg0 + q], k1 = band[seg0 + q + 1], k2 = band[seg0 + q + 2], k3 = band[seg0 + q + 3];
                rank += (k0 > key[u] ? 1 : 0) + (k1 > key[u] ? 1 : 0) + (k2 > key[u] ? 1 : 0) + (k3 > key[u] ? 1 : 0);
            }
            for (; q < cnt; ++q) rank += band[seg0 + q] > key[u] ? 1 : 0;
            pos[u] = seg0 + rank;
        }
    }
    __syncthreads();                         // every key is in a register: `order` aliases `band`
    NMS_STAMP(42);
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (pos[u] >= 0) {
            order[pos[u]] = 0xFFFFFFFFu - (unsigned)(key[u] & 0xFFFFFFFFull);
            if (pos[u] == n - 1) *low_out = key[u];
        }
    __syncthreads();
    return true;
}


// Laboratory build only (-DRPN_LAB, RPN_NMS_STOP=n): leave the kernel after phase n (outputs are then not written).
#ifdef RPN_LAB
#define NMS_STOP_AT(n) do { if (p.stop_after == (n)) return; } while (0)
#else
#define NMS_STOP_AT(n) ((void)0)
#endif


// ---- cluster mode: several workgroups per (image, class) for the first band ------------------------------------------
// One workgroup walks N scores at one compute unit's memory-level parallelism (~7 B / clk: 33 k cycles per pass at 61 440
// candidates, three passes for a saturated head: histogram, refinement, compaction).  With few images the other 255 CUs
// idle meanwhile.  In cluster mode G workgroups (blockIdx = pair * G + rank, dispatched together) each take a contiguous
// slice of the scores: slice histogram in LDS -> device-scope atomic adds into the cluster's global histogram -> cluster
// barrier -> every workgroup reads the 8 KB sum back and takes the same decision the single workgroup would have taken;
// the band's keys are compacted per slice into the rank's region of the workspace, and after the last barrier the LEADER
// (rank 0) orders the band and runs the greedy selection (and any later band) alone; the other ranks have left.
// No hang by construction of the WAIT, not of the dispatch: a waiting workgroup holds its CU and nothing guarantees that the
// ranks it waits for are resident (persistent kernels of another stream can hold every other CU for their whole run), so
// every wait is bounded in wall-clock time (kClusterWaitTicks) and a rank that does not show up in time is given up -- the
// leader recomputes alone.  The host keeps pairs * G <= 128 workgroups so that, on an idle device, all of them fit at once.
constexpr int kClusterMax = 16;
constexpr int kClCtlWords = 64 + 3 * 2048;      // [0..2] barrier counters, [16 + g] keys of rank g, [64..) three histograms

struct ClusterCtx {
    int G, g;
    unsigned *ctr;               // control words of this (image, class) pair
    unsigned *hist;              // global counters: [0, 2048) bins | [2048, 4096) sub-bins of the top bin | [4096, 6144) refinement
    unsigned long long *band;    // G regions of `region` keys
    int region;
};

// Arrive at cluster barrier `k`; `wait`: and wait for all G workgroups.  Hand-off WITHOUT cache-wide fences
// (MI355X_MICROARCH.md, inter-workgroup visibility, second form): every word one workgroup hands to another is written with a
// device-scope (sc1, write-through) store or atomic and read with a device-scope load; every storing thread drains its stores
// (vmcnt(0)) in front of the workgroup barrier that precedes thread 0's arrival (a relaxed device-scope atomic); thread 0 polls
// with relaxed device-scope loads.  (First version: __threadfence() by all 1024 threads on both sides and an acquire load per
// poll: ~25 k cycles per barrier; one release / acquire fence pair by thread 0: ~4-6 k -- an L2 write-back and invalidate
// each, which also evict what the kernels of the other stream are working from.)
// The wait is BOUNDED in wall-clock time (s_memrealtime, the constant 100 MHz counter: 2 ms -- several times the longest
// conv kernel another stream could be holding the CUs with, and independent of the shader clock and of load latency):
// returns false when a rank did not arrive in time, and the caller then gives the cluster up -- the leader recomputes the
// band alone, so a late rank costs at most the budget, never a hang or a wrong result.  `flag`: one int of LDS.
constexpr unsigned long long kClusterWaitTicks = 200000ull;
__device__ __forceinline__ bool cluster_arrive(const ClusterCtx &cl, int k, bool wait, int *flag)
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cl.ctr + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        if (wait) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(cl.ctr + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)cl.G) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > kClusterWaitTicks) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        *flag = ok;
    }
    __syncthreads();
    return *flag != 0;
}

// slice histograms (LDS, NH x 2048 words) -> + global -> barrier -> the cluster's sums back into LDS `hist`.
// bin_of(score, &second): first bin (< 0: not counted) and, optionally, a second one in [2048, 4096) (< 0: none).
template <int NH, class BinFn>
__device__ __forceinline__ bool cluster_histogram(const ClusterCtx &cl, int k, unsigned *ghist, const ScoreSrc &sc,
                                                  int lo, int hi, float score_thr, unsigned *hist, int *flag, BinFn bin_of)
{
    const int tid = fresh_tid();
    for (int i = tid; i < NH * 2048; i += kNmsThreads) hist[i] = 0u;
    __syncthreads();
    for (int base = lo + tid; base < hi; base += kLoadBatch * kNmsThreads) {
        float sb[kLoadBatch];
#pragma unroll
        for (int u = 0; u < kLoadBatch; ++u) sb[u] = (base + u * kNmsThreads < hi) ? sc(base + u * kNmsThreads) : NAN;
#pragma unroll
        for (int u = 0; u < kLoadBatch; ++u) {
            if (sb[u] > score_thr) {                                         // (NaN: not a candidate)
                int second = -1;
                const int bin = bin_of(sb[u], second);
                if (bin >= 0) atomicAdd(&hist[bin], 1u);
                if (NH > 1 && second >= 0) atomicAdd(&hist[second], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < NH * 2048; i += kNmsThreads) {
        const unsigned v = hist[i];
        if (v) __hip_atomic_fetch_add(ghist + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!cluster_arrive(cl, k, true, flag)) return false;
    for (int i = tid; i < NH * 2048; i += kNmsThreads)
        hist[i] = __hip_atomic_load(ghist + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return true;
}

// First band, cluster form of band_select_linear (hi_bound = all keys).  Every rank returns the same value:
//   0: fall back (the linear histogram does not apply, or a rank never arrived at a barrier) -- the leader runs the regular
//      single-workgroup path from the start, the others leave;
//   1: *thr_out / *count / *bin_out as band_select_linear would set them, and every rank has written its slice's band keys
//      to its region (count in ctr[16 + g]); ranks other than the leader have arrived at barrier 2 and must leave.
__device__ inline int cluster_first_band(const ClusterCtx &cl, const ScoreSrc &sc, int N, float score_thr,
                                         int want, int cap, unsigned *hist, int *ctl, unsigned long long *lds_keys,
                                         unsigned long long *thr_out, int *count, int *bin_out)
{
    const int tid = fresh_tid(), lane = tid & 63;
    const int per = (N + cl.G - 1) / cl.G;
    const int lo = min(N, cl.g * per), hi = min(N, lo + per);
    if (!cluster_histogram<2>(cl, 0, cl.hist, sc, lo, hi, score_thr, hist, ctl + 3, [](float s, int &second) {
            const int bin = linear_bin(s);
            if (bin == 2047) second = 2048 + linear_sub_bin(s, 2047);
            return bin;
        }))
        return 0;
    NMS_STAMP(20);
    hist_suffix_find(hist, hist + 4096, want, ctl, reinterpret_cast<unsigned *>(ctl) + 12);
    __syncthreads();
    const int d = ctl[0], n = ctl[1];
    const int in_d = d > 0 ? (int)hist[d] : 0;
    __syncthreads();
    unsigned long long thr = 0ull;
    int total = 0, bin = d;
    if (d == -1) { thr = 0ull; total = 0; }
    else if (d == -2) { thr = 1ull; total = n; bin = -1; }
    else if (d == 0) return 0;
    else if (n <= cap) { thr = (unsigned long long)orderable((float)d * (1.0f / 2048.0f)) << 32; total = n; }
    else {
        // the crossing bin holds more than the band may take: refine inside it (band_select_linear's second step; the top
        // bin's sub-bins came with the first pass)
        const int above = n - in_d;
        const unsigned *sub = hist + 2048;
        if (d != 2047) {
            if (!cluster_histogram<1>(cl, 1, cl.hist + 4096, sc, lo, hi, score_thr, hist, ctl + 3, [=](float s, int &) {
                    return linear_bin(s) == d ? linear_sub_bin(s, d) : -1;
                }))
                return 0;
            sub = hist;
        }
        hist_suffix_find(sub, sub == hist ? hist + 4096 : hist + 6144, want - above, ctl, reinterpret_cast<unsigned *>(ctl) + 12);
        __syncthreads();
        const int d2 = ctl[0], n2 = ctl[1];
        __syncthreads();
        bin = (d == 2047 && d2 >= 0) ? 4096 + d2 : -1;        // (as band_select_linear: the top bin's sub-bins can order the band)
        if (d2 < 0 || above + n2 > cap) return 0;
        thr = (unsigned long long)orderable((float)(d * 2048 + d2) * (1.0f / 4194304.0f)) << 32;
        total = above + n2;
    }
    NMS_STAMP(21);
    *thr_out = thr;
    *count = total;
    *bin_out = bin;
    if (thr == 0ull) return 1;               // no candidate at all: nothing to compact (the leader ends at once)
    // compact this slice's band keys: LDS first (ballot ranks, one LDS atomic per wave), then the rank's region
    if (tid == 0) ctl[2] = 0;
    __syncthreads();
    for (int base = lo + tid; base < hi; base += kLoadBatch * kNmsThreads) {
        float sb[kLoadBatch];
#pragma unroll
        for (int u = 0; u < kLoadBatch; ++u) sb[u] = (base + u * kNmsThreads < hi) ? sc(base + u * kNmsThreads) : NAN;
#pragma unroll
        for (int u = 0; u < kLoadBatch; ++u) {
            const unsigned long long key = make_key(sb[u], score_thr, base + u * kNmsThreads);
            const bool in = key >= thr && key != 0ull;
            const unsigned long long bal = __ballot(in);
            if (bal) {
                int slot0 = 0;
                if (lane == __ffsll((long long)bal) - 1) slot0 = atomicAdd(&ctl[2], __popcll(bal));
                slot0 = __shfl(slot0, __ffsll((long long)bal) - 1, 64);
                const int slot = slot0 + __popcll(bal & ((1ull << lane) - 1ull));
                if (in && slot < kBandCap) lds_keys[slot] = key;
            }
        }
    }
    __syncthreads();
    const int mine = min(ctl[2], cl.region);
    unsigned long long *dst = cl.band + (size_t)cl.g * cl.region;
    for (int i = tid; i < mine; i += kNmsThreads) __hip_atomic_store(dst + i, lds_keys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) __hip_atomic_store(cl.ctr + 16 + cl.g, (unsigned)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    NMS_STAMP(22);
    const int ok = cluster_arrive(cl, 2, cl.g == 0, ctl + 3) ? 1 : 0;
    NMS_STAMP(23);
    return ok;
}

// LDS carve-up (dynamic, 16-byte aligned)
__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

struct LdsLayout {
    size_t band, hist, sel_c, sel_idx, cbox, mask, dead, picked, ctl, scache, srt, pbin, total;
};

__host__ __device__ inline LdsLayout lds_layout(int max_sel, int cache_n, int prune)
{
    LdsLayout l;
    size_t o = 0;
    l.band = o;    o = align16(o + (size_t)8 * kBandCap);               // u64 keys, later u32 order in place
    l.hist = o;    o = align16(o + (size_t)4 * kBins);
    // canonical selected boxes: float4 [max_sel + 32] + area [max_sel + 32].  The 32 more: entries past the selected
    // count hold a box of infinite area (suppresses nothing, is decided at once), so the loops over the list run to the next
    // multiple of 32 with a uniform trip count and no lane predicate
    l.sel_c = o;   o = align16(o + (size_t)16 * (max_sel + 32));
    o = align16(o + (size_t)4 * (max_sel + 32));
    l.sel_idx = o; o = align16(o + (size_t)4 * max_sel);
    l.cbox = o;    o = align16(o + (size_t)20 * kChunk);                // canonical chunk boxes: float4 [kChunk] + area [kChunk]
    l.mask = o;    o = align16(o + (size_t)8 * 64 * kChunkWords);       // [64 rows of the current group][4] u64
    l.dead = o;    o = align16(o + (size_t)8 * kChunkWords);            // 4 u64 words
    l.picked = o;  o = align16(o + (size_t)4 * kChunk);
    l.ctl = o;     o = align16(o + 256);     // 64 ints: CTL_* | [16, 33) scan scratch | [36, 53) cluster prefix
    l.scache = o;  o = align16(o + (size_t)4 * cache_n);
    // area pruning: the selected boxes a second time, ordered by area bin (same padding rule as sel_c), and per bin the
    // count [64] | start offset [65] | placement cursor [64]
    l.srt = o;     if (prune) o = align16(o + (size_t)16 * (max_sel + 32)) + align16((size_t)4 * (max_sel + 32));
    l.pbin = o;    if (prune) o = align16(o + (size_t)4 * (3 * kPruneBins + 4));
    l.total = o;
    return l;
}

// ctl words
enum { CTL_NSEL = 0, CTL_BANDN = 1, CTL_POS = 2, CTL_SEL = 4 /* 4 words used by the band selection (+ its scan scratch at
       CTL_SEL + 12 .. + 28) */, CTL_LOW = 8 /* u64 */, CTL_OFLAG = 10 /* band_order_linear's "a segment is too long" flag */, CTL_CLPRE = 36 /* 17 ints: first key of each cluster rank's region */ };

// Debug build only (-DRPN_NMS_STAMP, scripts/nms_stamp_probe.py): cycle stamps of thread 0 at the phase boundaries of the
// first 64 workgroups, in program order (slot 0 = start; the probe knows the sequence).
template <bool DECODE>
__global__ void __launch_bounds__(kNmsThreads)
nms_kernel(NmsArgs p)
{
#ifdef RPN_NMS_STAMP
    if (threadIdx.x == 0) nms_stamp_idx() = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LdsLayout L = lds_layout(p.max_sel, p.cache_n, p.prune);
    unsigned long long *band = reinterpret_cast<unsigned long long *>(smem + L.band);
    unsigned *order = reinterpret_cast<unsigned *>(smem + L.band);
    unsigned *hist = reinterpret_cast<unsigned *>(smem + L.hist);
    float4 *sel_c4 = reinterpret_cast<float4 *>(smem + L.sel_c);
    float *sel_a = reinterpret_cast<float *>(smem + L.sel_c + align16((size_t)16 * (p.max_sel + 32)));
    int *sel_idx = reinterpret_cast<int *>(smem + L.sel_idx);
    float4 *cbox4 = reinterpret_cast<float4 *>(smem + L.cbox);               // canonical chunk boxes (ymin, xmin, ymax, xmax)
    float *carea = reinterpret_cast<float *>(smem + L.cbox) + 4 * kChunk;    // ... and their areas
    unsigned long long *mask = reinterpret_cast<unsigned long long *>(smem + L.mask);
    unsigned long long *deadw = reinterpret_cast<unsigned long long *>(smem + L.dead);
    int *ctl = reinterpret_cast<int *>(smem + L.ctl);
    int *cidx = reinterpret_cast<int *>(smem + L.picked);                    // compact mode: survivor -> its place in the chunk
    float4 *srt_c4 = reinterpret_cast<float4 *>(smem + L.srt);               // (p.prune only)
    float *srt_a = reinterpret_cast<float *>(smem + L.srt + align16((size_t)16 * (p.max_sel + 32)));
    int *pcnt = reinterpret_cast<int *>(smem + L.pbin), *pstart = pcnt + kPruneBins, *pcur = pstart + kPruneBins + 1;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int pair = (int)blockIdx.x / p.cluster;      // (image, class) pair; p.cluster workgroups each (1 = the normal case)
    const int b = pair / p.C;
    const int c = pair - b * p.C;
    const int qc = (p.q == 1) ? 0 : c;
    const int N = p.N, max_sel = p.max_sel;
    const float *__restrict__ sc = p.scores + (size_t)b * N * p.C + c;
    const int C = p.C;
    ScoreSrc src = make_score_src(sc, N, C, p.cache_n >= N ? reinterpret_cast<float *>(smem + L.scache) : nullptr);

    if (tid < 64) ctl[tid] = 0;
    for (int i = tid; i < max_sel + 32; i += kNmsThreads) {
        sel_c4[i] = float4{0.f, 0.f, 0.f, 0.f};
        sel_a[i] = INFINITY;
        if (p.prune) {
            srt_c4[i] = float4{0.f, 0.f, 0.f, 0.f};
            srt_a[i] = INFINITY;
        }
    }
    if (p.prune && tid < kPruneBins) pcnt[tid] = 0;
    __syncthreads();
    NMS_STAMP(1);

    unsigned long long hi_bound = ~0ull;       // keys of the current band are < hi_bound (exclusive)
    int visited = 0, have_before = 0;          // candidates of the band before this one; boxes selected before it (uniform)

    // cluster mode: the first band's threshold and keys come from all ranks of the cluster (cluster_first_band)
    ClusterCtx cl{};
    int cl_state = 0;                          // 1: the first band is in the workspace (cl_thr / cl_count / cl_bin)
    unsigned long long cl_thr = 0ull;
    int cl_count = 0, cl_bin = -1;
    if (p.cluster > 1) {
        cl.G = p.cluster;
        cl.g = (int)blockIdx.x % p.cluster;
        cl.ctr = p.cl_ctl + (size_t)pair * kClCtlWords;
        cl.hist = cl.ctr + 64;
        cl.region = p.cl_region;
        cl.band = p.cl_band + (size_t)pair * p.cluster * p.cl_region;
        if (p.linear_select) {
            const int band_target = min(kBandTarget, max(512, 3 * max_sel));
            const int band_cap = band_target <= 960 ? 1024 : kBandCap;
            cl_state = cluster_first_band(cl, src, N, p.score_thr, band_target, band_cap, hist, ctl + CTL_SEL, band, &cl_thr,
                                          &cl_count, &cl_bin);
        }
        if (cl.g != 0) return;                 // the leader goes on alone (with the cluster's band, or from scratch)
        __syncthreads();
    }

    while (true) {
        // ================= 1. pick the next band: keys in [thr, hi_bound) ==========================
        int band_expected = 0;
        // band size: a few times the number of boxes still wanted (most images finish inside the first band), so that
        // the sort below is a 1024-key rank sort instead of a 4096-key bitonic network
        const int have = ctl[CTL_NSEL];
        const int want_left = max_sel - have;
        // first band: three times the boxes wanted (ordering 900 keys costs what ordering 600 does; the chunks are only built
        // as far as the walk gets).  Later bands: sized from the yield of the band before (boxes selected per
        // candidate visited; it falls as the selected set grows, hence the factor 2), so that a deep walk -- low IoU
        // threshold, clustered boxes -- takes 2 bands instead of 4; up to what the histogram ordering handles (2 keys
        // per thread).  A band that is too large costs little: its chunks are only built as far as the walk gets.
        const int by_yield = (int)min((long long)want_left * visited / max(have - have_before, 1) * 2, 1792ll);
        const int band_target = visited == 0 ? min(min(kBandTarget, 1792), max(512, p.band1_mult * want_left)) : max(512, by_yield);
        have_before = have;
        const int band_cap = band_target <= 960 ? 1024 : kBandCap;
        unsigned long long thr = 0ull;
        int lin_bin = -1;                    // >= 0: the band was chosen by the linear histogram (still in `hist`) at this bin
        const bool from_cluster = cl_state == 1 && visited == 0;
        if (from_cluster) {
            thr = cl_thr;
            band_expected = cl_count;
            lin_bin = cl_bin;
        } else if (!p.linear_select ||
            !band_select_linear(src, N, p.score_thr, hi_bound, band_target, band_cap, hist, ctl + CTL_SEL, &thr, &band_expected,
                                &lin_bin, visited == 0 ? max(512, p.band1_sat * want_left) : 0,
                                max(512, p.band1_sat * want_left) <= 960 ? 1024 : kBandCap)) {
            lin_bin = -1;
            thr = radix_select<kNmsThreads>([&](int i) { return make_key(src(i), p.score_thr, i); }, N, hi_bound,
                                            band_target, band_cap, hist, ctl + CTL_SEL, &band_expected);
        }
        thr = uniform_u64(thr);
        if (thr == 0ull) break;              // nothing left
        NMS_STOP_AT(1);
        NMS_STAMP(2);

        int band_n = 0;
        bool ordered = false;
        // cluster mode: key i of the band's list = key (i - first[r]) of rank r's region
        if (from_cluster) {                  // ctl[CTL_CLPRE + r] = list position of rank r's first key
            if (tid < 64) {
                const int cnt = lane < cl.G ? (int)__hip_atomic_load(cl.ctr + 16 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                int incl = cnt;
                for (int off = 1; off < kClusterMax; off <<= 1) {
                    const int v = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += v;
                }
                if (lane < kClusterMax) ctl[CTL_CLPRE + lane] = incl - cnt;
            }
            __syncthreads();
        }
        auto cluster_key = [&](int i) {
            int r = 0;
#pragma unroll
            for (int q = 1; q < kClusterMax; ++q) r += (q < cl.G && i >= ctl[CTL_CLPRE + q]) ? 1 : 0;   // regions are in list order
            return __hip_atomic_load(cl.band + (size_t)r * cl.region + (i - ctl[CTL_CLPRE + r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        if (lin_bin > 0 && band_expected <= 2 * kNmsThreads && p.linear_select != 2) {
            // the band: main bins >= lin_bin, or (lin_bin >= 4096: the selection refined inside the top bin) the top bin's
            // sub-bins >= lin_bin - 4096.  A top bin of more than 32 keys (a saturated head) is ordered by its sub-bins either way.
            const bool refined_top = lin_bin >= 4096;
            const bool use_sub = refined_top || hist[2047] > 32u;       // (a top bin of a few keys is one segment like any other)
            if (use_sub && !refined_top) {       // the sub-bins' suffix sums (the refined selection left them at hist + 6144)
                hist_suffix_find(hist + 2048, hist + 6144, 0x3fffffff, ctl + CTL_SEL, reinterpret_cast<unsigned *>(ctl + CTL_SEL) + 12);
                __syncthreads();
            }
            const int d_main = refined_top ? 2048 : lin_bin, d_sub = refined_top ? lin_bin - 4096 : 0;
            unsigned *oflag = reinterpret_cast<unsigned *>(ctl + CTL_OFLAG);
            if (from_cluster)
                ordered = band_order_linear([&](auto fn) { for (int i = fresh_tid(); i < band_expected; i += kNmsThreads) fn(cluster_key(i)); },
                                            d_main, d_sub, use_sub, band_expected, hist, oflag, band, order,
                                            reinterpret_cast<unsigned long long *>(ctl + CTL_LOW));
            else
            ordered = band_order_linear([&](auto fn) { scan_band_keys(src, N, p.score_thr, thr, hi_bound, fn); },
                                        d_main, d_sub, use_sub, band_expected, hist, oflag, band, order,
                                        reinterpret_cast<unsigned long long *>(ctl + CTL_LOW));
            if (ordered) {
                band_n = band_expected;
                hi_bound = uniform_u64(*reinterpret_cast<unsigned long long *>(ctl + CTL_LOW));   // the next band continues strictly below
            }
        }
        if (!ordered) {

        // ---- compact the band's keys into LDS, pad to a power of two, sort descending ------------------
        if (tid == 0) ctl[CTL_BANDN] = from_cluster ? min(band_expected, kBandCap) : 0;
        __syncthreads();
        if (from_cluster) {                  // the cluster's ranks compacted their slices already: copy the list
            for (int i = tid; i < min(band_expected, kBandCap); i += kNmsThreads) band[i] = cluster_key(i);
        } else
        for (int base = tid; base < N; base += kLoadBatch * kNmsThreads) {       // loads batched: kLoadBatch in flight per thread
            float sb[kLoadBatch];
#pragma unroll
            for (int u = 0; u < kLoadBatch; ++u) {
                const int i = base + u * kNmsThreads;
                sb[u] = i < N ? src(i) : NAN;
            }
#pragma unroll
            for (int u = 0; u < kLoadBatch; ++u) {
                const unsigned long long key = make_key(sb[u], p.score_thr, base + u * kNmsThreads);      // 0 for NaN
                const bool in = key >= thr && key < hi_bound && key != 0ull;
                // one LDS atomic per wave: slots handed out by ballot rank (the band is sorted afterwards, so the order
                // of the slots is irrelevant)
                const unsigned long long bal = __ballot(in);
                if (bal) {
                    int slot0 = 0;
                    if (lane == __ffsll((long long)bal) - 1) slot0 = atomicAdd(&ctl[CTL_BANDN], __popcll(bal));
                    slot0 = __shfl(slot0, __ffsll((long long)bal) - 1, 64);
                    const int slot = slot0 + __popcll(bal & ((1ull << lane) - 1ull));
                    if (in && slot < kBandCap) band[slot] = key;
                }
            }
        }
        __syncthreads();
        NMS_STAMP(3);
        NMS_STOP_AT(3);
        band_n = ctl[CTL_BANDN];
        if (band_n > kBandCap) band_n = kBandCap;    // cannot happen: the select guarantees <= kBandCap
        if (band_n == 0) break;
        if (band_n <= kNmsThreads) {
            // one key per thread; keys -> box indices in place (order aliases band: every key is in a register first)
            const unsigned long long sorted = bitonic1024_desc(tid < band_n ? band[tid] : 0ull, band);
            NMS_STOP_AT(2);
            if (tid < band_n) order[tid] = 0xFFFFFFFFu - (unsigned)(sorted & 0xFFFFFFFFull);
            if (tid == band_n - 1) *reinterpret_cast<unsigned long long *>(ctl + CTL_LOW) = sorted;
            __syncthreads();
            hi_bound = uniform_u64(*reinterpret_cast<unsigned long long *>(ctl + CTL_LOW));   // the next band continues strictly below
        } else {
            int n_sort = 2;
            while (n_sort < band_n) n_sort <<= 1;
            for (int i = band_n + tid; i < n_sort; i += kNmsThreads) band[i] = 0ull;
            __syncthreads();
            bitonic_sort_desc(band, n_sort);
            NMS_STOP_AT(2);
            // keys -> 32-bit box indices, in place (read all, barrier, write)
            unsigned idx_reg[kBandCap / kNmsThreads];
#pragma unroll
            for (int j = 0; j < kBandCap / kNmsThreads; ++j) {
                const int i = tid + j * kNmsThreads;
                idx_reg[j] = (i < band_n) ? (0xFFFFFFFFu - (unsigned)(band[i] & 0xFFFFFFFFull)) : 0u;
            }
            const unsigned long long lowest = band[band_n - 1];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kBandCap / kNmsThreads; ++j) {
                const int i = tid + j * kNmsThreads;
                if (i < band_n) order[i] = idx_reg[j];
            }
            hi_bound = uniform_u64(lowest);          // the next band continues strictly below this one
        }
        }   // !ordered
        if (tid == 0) ctl[CTL_POS] = 0;
        __syncthreads();
        NMS_STAMP(4);

        // ================= 2. greedy selection over chunks of the sorted band ===============================
        while (true) {
            const int nsel = __builtin_amdgcn_readfirstlane(ctl[CTL_NSEL]);
            const int pos = ctl[CTL_POS];
            if (nsel >= max_sel || pos >= band_n) break;
            const int T = min(kChunk, band_n - pos);

            if (tid < kChunk && tid < T) {           // chunk boxes -> canonical form in LDS
                const CBox cb = canonical(fetch_box<DECODE>(p, b, (int)order[pos + tid], qc));
                const int st = cslot(tid);
                cbox4[st] = make_float4(cb.ymin, cb.xmin, cb.ymax, cb.xmax);
                carea[st] = nms_area_key(cb.area);          // (+inf for area <= 0: see nms_suppresses_fast)
            }
            if (tid < kChunkWords) deadw[tid] = 0ull;
            if (p.prune && nsel > 0 && (tid >> 6) == 8) {      // one wave: the bins' start offsets from their counts
                const int cnt = pcnt[lane];
                int incl = cnt;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int up = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += up;
                }
                pstart[lane] = incl - cnt;
                pcur[lane] = incl - cnt;
                if (lane == 63) pstart[64] = incl;
            }
            __syncthreads();

            if (tid < kChunkWords) {                 // slots past the band end count as dead
                const int lo = tid * 64;
                if (T < lo + 64) deadw[tid] = T <= lo ? ~0ull : (~0ull << (T - lo));
            }
            if (p.prune) {                           // the selected list, re-ordered by area bin (the order inside a bin is free)
                for (int i = tid; i < nsel; i += kNmsThreads) {
                    const float ar = sel_a[i];
                    const int at = atomicAdd(&pcur[prune_bin(ar)], 1);
                    srt_c4[at] = sel_c4[i];
                    srt_a[at] = ar;
                }
            }
            __syncthreads();
            NMS_STAMP(5);
            NMS_STOP_AT(4);

            // ---- compact mode: (A) for the whole chunk, then the survivors to the front -----------------------------------
            int Tg = T;                                  // candidates the group loop walks (compact: the survivors)
            const bool compacted = p.compact > 0 && nsel >= p.compact;
            if (compacted) {
                {   // thread (candidate = tid / P, part = tid % P), P = threads per candidate (4; 2 with 512-candidate chunks), tests
                    // selected boxes part, part + P, ... (the list is padded to a multiple of 32 with boxes that suppress nothing);
                    // the tests are those of the per-group step (A) below
                    constexpr int P = kNmsThreads / kChunk, CPW = 64 / P;          // candidates per wave
                    const int tid = fresh_tid(), lane = tid & 63;
                    const int t = tid / P, part = tid % P;
                    bool hit = false;
                    if (t < T) {
                        const int st = cslot(t);
                        const float4 c4 = cbox4[st];
                        const CBox ci{c4.x, c4.y, c4.z, c4.w, carea[st]};
                        int run0 = 0, run_n = nsel;
                        if (p.prune) {
                            run0 = pstart[prune_bin(ci.area * p.prune_lo)];
                            run_n = pstart[prune_bin(ci.area * p.prune_hi) + 1] - run0;
                        }
                        const float4 *sp = (p.prune ? srt_c4 : sel_c4) + run0 + part;
                        const float *ap = (p.prune ? srt_a : sel_a) + run0 + part;
                        float ha = -INFINITY, hb = -INFINITY, ma = p.m0, mb = p.m0;
                        for (int j0 = 0; j0 < run_n; j0 += 2 * P) {
                            const float4 a4 = sp[j0], b4 = sp[j0 + P];
                            const CBox sa{a4.x, a4.y, a4.z, a4.w, ap[j0]}, sb{b4.x, b4.y, b4.z, b4.w, ap[j0 + P]};
                            ha = vmax_vv(ha, nms_excess(ci, sa, p.iou_thr, p.iou_eps, ma));
                            hb = vmax_vv(hb, nms_excess(ci, sb, p.iou_thr, p.iou_eps, mb));
                        }
                        hit = vmax_vv(ha, hb) > 0.0f;
                        if (!(vmin_vv(ma, mb) > 1e-30f)) {
                            hit = false;
                            for (int j = 0; j < ((run_n + 2 * P - 1) & ~(2 * P - 1)); j += P) {
                                const float4 a4 = sp[j];
                                const CBox sa{a4.x, a4.y, a4.z, a4.w, ap[j]};
                                hit |= nms_suppresses(ci, sa, p.iou_thr, p.iou_lo, p.iou_hi);
                            }
                        }
                    }
                    // lanes P q .. P q + P - 1 of a wave belong to candidate CPW * wave + q
                    const unsigned long long bal = __ballot(hit);
                    if (lane == 0 && bal) {
                        unsigned long long bits = 0ull;
#pragma unroll
                        for (int q = 0; q < CPW; ++q)
                            if ((bal >> (P * q)) & ((1ull << P) - 1ull)) bits |= 1ull << q;
                        const int c0 = (tid >> 6) * CPW;                           // the wave's first candidate
                        atomicOr(&deadw[c0 >> 6], bits << (c0 & 63));
                    }
                }
                __syncthreads();
                NMS_STAMP(44);
                // survivors keep their order: rank = live candidates in front
                float4 my4 = float4{0.f, 0.f, 0.f, 0.f};
                float my_a = 0.0f;
                int my_rank = -1, n_live = 0;
                {
                    const int tid = fresh_tid();
                    unsigned long long live[kChunkWords];
#pragma unroll
                    for (int w = 0; w < kChunkWords; ++w) {
                        live[w] = ~deadw[w];
                        n_live += __popcll(live[w]);
                    }
                    if (tid < kChunk) {
                        const int w = tid >> 6, bit = tid & 63;
                        if ((live[w] >> bit) & 1ull) {
                            int r = __popcll(live[w] & ((1ull << bit) - 1ull));
#pragma unroll
                            for (int v = 0; v < kChunkWords; ++v)
                                if (v < w) r += __popcll(live[v]);
                            my_rank = r;
                            my4 = cbox4[cslot(tid)];
                            my_a = carea[cslot(tid)];
                        }
                    }
                }
                __syncthreads();                         // every survivor holds its box: the slots may be rewritten
                {
                    const int tid = fresh_tid();
                    if (my_rank >= 0) {
                        cbox4[cslot(my_rank)] = my4;
                        carea[cslot(my_rank)] = my_a;
                        cidx[my_rank] = tid;
                    }
                    if (tid < kChunkWords) {             // the group loop's dead words: only the slots past the survivors
                        const int lo = tid * 64;
                        deadw[tid] = n_live <= lo ? ~0ull : (n_live >= lo + 64 ? 0ull : (~0ull << (n_live - lo)));
                    }
                }
                Tg = n_live;
                __syncthreads();
                NMS_STAMP(45);
            }

            // The chunk is walked in 4 groups of 64 candidates (one 64-bit word of the live mask).  Per group:
            // (B) all 16 waves build the suppression rows of the group's LIVE candidates against the live candidates after
            //     them: thread (i = tid / 16, piece = tid % 16) produces 16 bits of row i.  Rows of later groups are only
            //     built if the walk gets there (a chunk that fills the output early costs one group, not a 256 x 256 matrix).
            // (C) wave 0 walks the group.  Lane l holds row l's own-group word in registers, the live word lives in scalar
            //     registers: one selection = find-first-set + two v_readlane (the selected row) + and-not, no LDS round
            //     trip (the first version read the row from LDS per selection: 245 cycles each, 30 us per full chunk).
            //     Selected lanes then OR the rest of their rows into the dead words of the later groups and append
            //     themselves to the selected list at position nsel + (rank among the group's selected).
            int cur = nsel;                                          // uniform: re-read from LDS after every group
            for (int gw = 0; gw < kChunkWords && gw * 64 < Tg && cur < max_sel; ++gw) {
                // (A) the group's candidates against the boxes selected in EARLIER chunks (this chunk's selections reach
                //     later groups through the row ORs below): thread (candidate = tid / 16, part = tid % 16) tests selected
                //     boxes part, part + 16, ... (LDS broadcast within a part).  Done per group, so a chunk that completes
                //     the output after one group tests 64 x nsel pairs, not 256 x nsel.  (Round 5: batches of several groups per
                //     interval measured SLOWER -- 124 -> 132 us at configs[2], IoU 0.5 -- although thread 0's own step (A) took
                //     6.8 k cycles for four groups against 4 x 3.5 k: the interval [B2, A, B1] overlaps its pieces well as it is.)
                if (nsel > 0 && !compacted) {
                    const int tid = fresh_tid(), lane = tid & 63;
                    const int il = tid >> 4, part = tid & 15;
                    const int t = gw * 64 + il;
                    bool hit = false;
                    if (t < T) {
                        const int st = cslot(t);
                        const float4 c4 = cbox4[st];
                        const CBox ci{c4.x, c4.y, c4.z, c4.w, carea[st]};
                        // two selected boxes per trip (their LDS reads overlap), a uniform number of trips: the list is
                        // padded to a multiple of 32 with boxes that suppress nothing.  Branch-free: the largest excess
                        // and the smallest margin (nms_excess); the exact tests only if one ratio was close
                        // (area pruning: only the run of area bins whose boxes can reach the threshold with this one)
                        int run0 = 0, run_n = nsel;
                        if (p.prune) {
                            run0 = pstart[prune_bin(ci.area * p.prune_lo)];
                            run_n = pstart[prune_bin(ci.area * p.prune_hi) + 1] - run0;
                        }
                        const float4 *sp = (p.prune ? srt_c4 : sel_c4) + run0 + part;
                        const float *ap = (p.prune ? srt_a : sel_a) + run0 + part;
                        float ha = -INFINITY, hb = -INFINITY, ma = p.m0, mb = p.m0;
                        for (int j0 = 0; j0 < run_n; j0 += 32) {
                            const float4 a4 = sp[j0], b4 = sp[j0 + 16];
                            const CBox sa{a4.x, a4.y, a4.z, a4.w, ap[j0]}, sb{b4.x, b4.y, b4.z, b4.w, ap[j0 + 16]};
                            ha = vmax_vv(ha, nms_excess(ci, sa, p.iou_thr, p.iou_eps, ma));
                            hb = vmax_vv(hb, nms_excess(ci, sb, p.iou_thr, p.iou_eps, mb));
                        }
                        hit = vmax_vv(ha, hb) > 0.0f;
                        if (!(vmin_vv(ma, mb) > 1e-30f)) {
                            hit = false;                // (the same entries as this lane's loop above, padding included)
                            for (int j = 0; j < ((run_n + 31) & ~31); j += 16) {
                                const float4 a4 = sp[j];
                                const CBox sa{a4.x, a4.y, a4.z, a4.w, ap[j]};
                                hit |= nms_suppresses(ci, sa, p.iou_thr, p.iou_lo, p.iou_hi);
                            }
                        }
                    }
                    // lanes 16q .. 16q+15 of a wave belong to candidate 4 * wave + q
                    const unsigned long long bal = __ballot(hit);
                    if (lane == 0 && bal) {
                        unsigned long long bits = 0ull;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if ((bal >> (16 * q)) & 0xFFFFull) bits |= 1ull << (4 * (tid >> 6) + q);
                        atomicOr(&deadw[gw], bits);
                    }
                    NMS_STAMP(6);
                }
                // No barrier between (A), (B1) and the (B2) of the group before: all three only OR bits into dead words or
                // write this group's rows; (B1) reads the dead word merely to SKIP work (a candidate that is dead, or whose
                // death is not visible yet, may sit in a suppressor set: the walk only ever looks at live candidates' sets
                // and only at their live / selected members).  One barrier in front of the walk closes the interval, so the
                // latency-bound pieces (B1, B2) run under the issue-bound one (A): 4 barrier intervals per group -> 2.
                // (B1) the group's own 64 x 64 block only: for each live candidate i the candidates BEFORE it that suppress it
                // (its suppressor set: the walk below resolves the group from these, in parallel; the test is symmetric in
                // its two boxes).  16 threads per row, 4 candidates each.  What the group's SELECTED boxes suppress in the
                // later groups of the chunk is computed after the walk (B2) -- for the selected rows only, where building
                // full rows before the walk did it for every live candidate, selected or not.
                {
                    // 16 threads per row, 4 candidates each: piece (sub & 3) of the group, rows 4 (sub >> 2) .. + 3 of it
                    const int tid = fresh_tid();
                    const int il = tid >> 4, sub = tid & 15;
                    const int piece = 4 * gw + (sub & 3), jj0 = 4 * (sub >> 2);
                    const int i = gw * 64 + il;
                    const int j0 = piece * 16 + jj0;
                    unsigned bits = 0u;
                    const unsigned long long dw = deadw[gw];
                    if (!((dw >> il) & 1ull) && j0 < i) {
                        const int is = cslot(i);
                        const float4 s4 = cbox4[is];
                        const CBox si{s4.x, s4.y, s4.z, s4.w, carea[is]};
                        unsigned todo = (unsigned)(~dw >> (j0 & 63)) & 0xFu;
                        if (j0 + 4 > i) todo &= (1u << (i - j0)) - 1u;             // only j < i
                        float m = p.m0;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int js = cslot2(jj0 + t, piece);
                            const float4 q4 = cbox4[js];
                            const CBox cj{q4.x, q4.y, q4.z, q4.w, carea[js]};
                            bits |= (unsigned)(nms_excess(cj, si, p.iou_thr, p.iou_eps, m) > 0.0f) << t;
                        }
                        if (!(m > 1e-30f)) {                                // a ratio close to the threshold: the exact tests
                            bits = 0u;
                            for (int t = 0; t < 4; ++t) {
                                const int js = cslot2(jj0 + t, piece);
                                const float4 q4 = cbox4[js];
                                const CBox cj{q4.x, q4.y, q4.z, q4.w, carea[js]};
                                if (nms_suppresses(cj, si, p.iou_thr, p.iou_lo, p.iou_hi)) bits |= 1u << t;
                            }
                        }
                        bits &= todo;
                    }
                    // bits 16 (sub & 3) + 4 (sub >> 2) + t of row il's own word: lanes sub and sub ^ 4 share a byte
                    const unsigned other = (unsigned)__shfl_xor((int)bits, 4);
                    if (!(sub & 4))
                        reinterpret_cast<unsigned char *>(mask)[(il * kChunkWords + gw) * 8 + 2 * (sub & 3) + (sub >> 3)] =
                            (unsigned char)(bits | (other << 4));
                }
                __syncthreads();
                NMS_STAMP(7);
                NMS_STOP_AT(5);
                if (tid < 64) {
                    const unsigned long long own = mask[lane * kChunkWords + gw];
                    const unsigned own_lo = (unsigned)own, own_hi = (unsigned)(own >> 32);
                    const unsigned long long dead0 = deadw[gw];
                    // live candidates of the group, uniform (scalar registers); slots past the band end are dead already
                    // (the builtins return int: widen through unsigned, or bit 31 sign-extends over the upper word)
                    unsigned long long avail =
                        ~(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(dead0 >> 32)) << 32) |
                          (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)dead0));
                    // Greedy selection inside the group as a fixed point instead of a 64-step serial walk (one selection per
                    // ~200 cycles of scalar <-> vector round trips: 7-9k cycles per group in the stamps): candidate l is
                    // selected iff it is live and no SELECTED earlier candidate is in its suppressor set `own`.  Per round
                    // every undecided lane looks at its set: a selected member -> dead; no undecided member left -> selected
                    // (the lowest undecided lane always decides, so the loop ends; chains are a few links long in practice).
                    // The unique solution of that recursion is the sequential walk's result; the output cap keeps its
                    // first (max_sel - cur) members, exactly where the walk would have stopped.
                    (void)own_lo; (void)own_hi;
                    const unsigned long long mybit = 1ull << lane;
                    unsigned long long und = avail, selall = 0ull;
                    while (und != 0ull) {
                        const bool me = (und & mybit) != 0ull;
                        const bool hit = (own & selall) != 0ull;
                        const bool blocked = (own & und) != 0ull;
                        const unsigned long long newsel = __ballot(me && !hit && !blocked);
                        const unsigned long long newdead = __ballot(me && hit);
                        selall |= newsel;
                        und &= ~(newsel | newdead);
                    }
                    const int room = max_sel - cur;
                    const bool picked_me = ((selall >> lane) & 1ull) && __popcll(selall & (mybit - 1ull)) < room;
                    const unsigned long long selmask = __ballot(picked_me);
                    const int c2 = cur + __popcll(selmask);
                    if (picked_me) {
                        const int slot = cur + __popcll(selmask & ((1ull << lane) - 1ull));
                        const int i = gw * 64 + lane;
                        const float4 b4 = cbox4[cslot(i)];
                        sel_c4[slot] = b4;
                        sel_a[slot] = carea[cslot(i)];
                        sel_idx[slot] = (int)order[pos + (compacted ? cidx[i] : i)];
                    }
                    if (lane == 0) ctl[CTL_NSEL] = c2;
                    NMS_STAMP(8);
                }
                __syncthreads();
                NMS_STAMP(9);
                {
                    // (B2) the boxes this group selected (list positions cur .. new count) against the candidates of the
                    // chunk's later groups: item = (selected box, 16-candidate piece); hits go straight into the later
                    // groups' dead words (read by their walks behind at least one more barrier)
                    // (finer items when there are few of them: 8 or 4 candidates per thread instead of 16)
                    const int tid = fresh_tid();
                    const int c_new = ctl[CTL_NSEL];
                    if (p.prune && tid >= kNmsThreads - 64 && tid - (kNmsThreads - 64) < c_new - cur)    // the new boxes' area bins
                        atomicAdd(&pcnt[prune_bin(sel_a[cur + tid - (kNmsThreads - 64)])], 1);
                    const int npl = 4 * (kChunkWords - 1 - gw);
                    const int items16 = c_new < max_sel ? (c_new - cur) * npl : 0;    // (output complete: nothing follows)
                    const int fsh = items16 <= kNmsThreads / 4 ? 2 : (items16 <= kNmsThreads / 2 ? 1 : 0);   // log2(parts per piece)
                    const int per = 16 >> fsh;
                    for (int item = tid; item < (items16 << fsh); item += kNmsThreads) {
                        const int pi = item % npl, rest = item / npl;       // the piece varies fastest: conflict-free reads
                        const int part = rest & ((1 << fsh) - 1), sl = cur + (rest >> fsh);
                        const int piece = 4 * (gw + 1) + pi, jj0 = part * per;
                        if (piece * 16 + jj0 >= Tg) continue;               // past the band end / the survivors: dead already
                        const float4 s4 = sel_c4[sl];
                        const CBox si{s4.x, s4.y, s4.z, s4.w, sel_a[sl]};
                        unsigned bits = 0u;
                        float m = p.m0;
#pragma unroll 4
                        for (int t = 0; t < per; ++t) {
                            const int js = cslot2(jj0 + t, piece);
                            const float4 q4 = cbox4[js];
                            const CBox cj{q4.x, q4.y, q4.z, q4.w, carea[js]};
                            bits |= (unsigned)(nms_excess(cj, si, p.iou_thr, p.iou_eps, m) > 0.0f) << t;
                        }
                        if (!(m > 1e-30f)) {
                            bits = 0u;
                            for (int t = 0; t < per; ++t) {
                                const int js = cslot2(jj0 + t, piece);
                                const float4 q4 = cbox4[js];
                                const CBox cj{q4.x, q4.y, q4.z, q4.w, carea[js]};
                                if (nms_suppresses(cj, si, p.iou_thr, p.iou_lo, p.iou_hi)) bits |= 1u << t;
                            }
                        }
                        if (bits) atomicOr(&deadw[piece >> 2], (unsigned long long)bits << ((piece & 3) * 16 + jj0));
                    }
                    // (no barrier: the next group's walk reads its dead word behind the barrier that follows its (B1))
                }
                cur = ctl[CTL_NSEL];
                NMS_STOP_AT(6);
            }
            if (tid == 0) ctl[CTL_POS] = pos + T;
            __syncthreads();
            NMS_STOP_AT(7);
        }
        if (ctl[CTL_NSEL] >= max_sel) break;
        visited = band_n;
        __syncthreads();
    }
    __syncthreads();

    // ================= 3. outputs =================================================================
    NMS_STAMP(10);
    NMS_STOP_AT(8);
    const int nsel = ctl[CTL_NSEL];
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
        NMS_STAMP(11);
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

// hipFuncAttributeMaxDynamicSharedMemorySize of one kernel, raised only when a launch needs more than any launch before it
// on the same device (hipFuncSetAttribute is a driver call of a few microseconds; the NMS is launched every step).
struct DynLdsCeiling {
    std::mutex mu;
    size_t have[16] = {};
    hipError_t raise(const void *fn, size_t bytes)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lock(mu);
        if (dev >= 0 && dev < 16 && have[dev] >= bytes) return hipSuccess;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess && dev >= 0 && dev < 16) have[dev] = bytes;
        return e;
    }
};

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

// Cluster mode (cluster_first_band): workgroups per (image, class) pair.  Automatic for few pairs with many candidates
// (<= 32 pairs, >= 16 384 candidates: slices of >= 2048 scores, at most 128 workgroups in all -- half the chip, so that
// every workgroup of a cluster is resident or next in line while the others wait for it).  RPN_NMS_CLUSTER=1 turns it
// off, = 2 .. 16 forces that size wherever it fits (tests run the small problems through it that way).
static int cluster_size(int pairs, int N)
{
    static const int knob = RPN_KNOB("RPN_NMS_CLUSTER", 0);
    if (knob == 1 || pairs <= 0) return 1;
    int G = 1;
    if (knob >= 2) {
        G = knob < kClusterMax ? knob : kClusterMax;
        while (G > 1 && (pairs * G > 128 || N < 64 * G)) --G;
    } else if (pairs <= 32 && N >= 16384) {
        G = kClusterMax;
        while (G > 1 && (pairs * G > 128 || N / G < 2048)) G >>= 1;
    }
    return G;
}
static int cluster_region(int G, int N) { const int per = (N + G - 1) / G; return per < kBandCap ? per : kBandCap; }
static size_t cluster_ctl_bytes(int pairs) { return align16((size_t)pairs * kClCtlWords * sizeof(unsigned)); }
static size_t cluster_bytes(int pairs, int G, int N)
{
    if (G <= 1) return 0;
    return cluster_ctl_bytes(pairs) + align16((size_t)pairs * G * cluster_region(G, N) * sizeof(unsigned long long));
}

template <bool DECODE>
static int launch_nms(NmsArgs &p, void *d_workspace, size_t workspace_bytes, hipStream_t stream)
{
#ifdef RPN_LAB
    static const int stop = RPN_LAB_KNOB("RPN_NMS_STOP", 0);
    p.stop_after = stop;
#endif
    static const int linear = RPN_KNOB("RPN_NMS_LINEAR", 1);
    p.linear_select = linear;
    if (p.iou_thr > 0.0f && p.iou_thr < INFINITY) {
        p.iou_lo = p.iou_thr * (1.0f - 0x1p-18f);
        p.iou_hi = p.iou_thr * (1.0f + 0x1p-18f);
    } else {
        p.iou_lo = -INFINITY;
        p.iou_hi = INFINITY;
    }
    const bool fast = p.iou_thr >= 1e-6f && p.iou_thr <= 1e6f;
    p.iou_eps = fast ? p.iou_thr * 0x1p-18f : 0.0f;
    p.m0 = fast ? INFINITY : 0.0f;
    // LDS score cache: when the pair's scores fit beside the rest (and the first pass is the linear histogram's, by one
    // workgroup: not in cluster mode, whose first pass is shared)
    p.cache_n = 0;
    p.cluster = cluster_size(p.B * p.C, p.N);
    if (p.cluster > 1 && (!d_workspace || workspace_bytes < stage_bytes(p.B, p.C, p.max_sel) + cluster_bytes(p.B * p.C, p.cluster, p.N)))
        p.cluster = 1;
    // first band: 5 x the boxes wanted, 3 x on a saturated head (NmsArgs::band1_mult / band1_sat)
    {
        static const int b1 = RPN_LAB_KNOB("RPN_NMS_BAND1", 0), bs = RPN_LAB_KNOB("RPN_NMS_BAND1SAT", 3);    // (A/B timing; 0 = automatic)
        // automatic: 6 x below a threshold of 0.6 -- a low threshold visits more candidates per selected box, and with the chunk
        // compaction (cheap later chunks) a first band that ends the walk is worth its larger ordering pass: configs[2] at IoU 0.5
        // 102.5 -> 92.8 us (the band is capped at 1 792 candidates: 8 x is the same; a cap of 2 048: 89.4 but the bench model's head
        // outputs 305 -> 347), the bench model's head outputs 298.8 -> 305.7, smooth-score inputs 126.6 -> 131.1; at 0.7 no gain: left at 5 x
        p.band1_mult = b1 > 0 ? b1 : (p.iou_thr < 0.6f ? 6 : 5);
        p.band1_sat = bs;
    }
    // area pruning: a threshold in [0.3, 1) (a candidate's run of area bins is then at most 18 of the 64) on the fast test path
    static const int prune = RPN_KNOB("RPN_NMS_PRUNE", 1);
    p.prune = prune && fast && p.iou_thr >= 0.3f && p.iou_thr < 1.0f && lds_layout(p.max_sel, 0, 1).total <= kLdsLimit;
    {
        static const int compact = RPN_LAB_KNOB("RPN_NMS_COMPACT", 1);     // (A/B timing: 0 = never, n = chunks that start with >= n selected)
        p.compact = compact;
    }
    if (p.prune) {
        p.prune_lo = p.iou_thr * (1.0f - 0x1p-19f);
        p.prune_hi = 1.0f / p.prune_lo * (1.0f + 0x1p-22f);        // (the quotient's own rounding is 2^-24)
    }
    if (p.linear_select && p.cluster == 1 && p.N <= kScoreCacheMax && lds_layout(p.max_sel, p.N, p.prune).total <= kLdsLimit) p.cache_n = p.N;
    const LdsLayout L = lds_layout(p.max_sel, p.cache_n, p.prune);
    if (L.total > kLdsLimit)
        return fail(RPN_ERR_UNSUPPORTED, "nms: %zu bytes of LDS needed (max per class=%d) > %zu", L.total, p.max_sel,
                    kLdsLimit);
    const size_t stage = stage_bytes(p.B, p.C, p.max_sel);
    if (p.C > 1) {
        if (!d_workspace || workspace_bytes < stage)
            return fail(RPN_ERR_WORKSPACE, "nms: workspace of %zu bytes needed, %zu given", stage, workspace_bytes);
        p.stage_idx = reinterpret_cast<int *>(d_workspace);
        p.stage_cnt = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(d_workspace) +
                                              align16((size_t)p.B * p.C * p.max_sel * sizeof(int)));
    }
    // cluster mode needs its own scratch behind the staging area; a caller that passes less simply gets one workgroup per
    // pair (same results, the passes over the scores are not shared)
    if (p.cluster > 1) {
        unsigned char *base = reinterpret_cast<unsigned char *>(d_workspace) + stage;
        p.cl_ctl = reinterpret_cast<unsigned *>(base);
        p.cl_band = reinterpret_cast<unsigned long long *>(base + cluster_ctl_bytes(p.B * p.C));
        p.cl_region = cluster_region(p.cluster, p.N);
        RPN_HIP_CHECK(hipMemsetAsync(p.cl_ctl, 0, cluster_ctl_bytes(p.B * p.C), stream));   // counters + the two histograms
    }
    auto kern = nms_kernel<DECODE>;
    {   // the dynamic-LDS ceiling of a function only ever has to grow: set once per (device, size), not on every launch
        static DynLdsCeiling ceiling;
        RPN_HIP_CHECK(ceiling.raise(reinterpret_cast<const void *>(kern), L.total));
    }
    hipLaunchKernelGGL(kern, dim3(p.B * p.C * p.cluster), dim3(kNmsThreads), L.total, stream, p);
    RPN_CHECK_LAUNCH();
    if (p.C > 1) {
        const int n_merge = next_pow2(p.C * p.max_sel);
        const size_t lds = (size_t)8 * n_merge + 16;
        if (lds > kLdsLimit)
            return fail(RPN_ERR_UNSUPPORTED, "nms merge: C*max_per_class = %d too large", p.C * p.max_sel);
        {
            static DynLdsCeiling ceiling;
            RPN_HIP_CHECK(ceiling.raise(reinterpret_cast<const void *>(nms_merge_kernel), lds));
        }
        hipLaunchKernelGGL(nms_merge_kernel, dim3(p.B), dim3(kNmsThreads), lds, stream, p, n_merge);
        RPN_CHECK_LAUNCH();
    }
    return RPN_OK;
}

}  // namespace rpn

using namespace rpn;

extern "C" size_t rpn_nms_workspace_bytes(int B, int N, int C, int max_per_class, int max_total)
{
    (void)max_total;
    if (B <= 0 || C < 1 || max_per_class <= 0) return 0;
    return stage_bytes(B, C, max_per_class) + cluster_bytes(B * C, cluster_size(B * C, N), N);
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
    RPN_REQUIRE_DEVICE();
    NmsArgs p{};
    p.boxes = d_boxes;
    p.scores = d_scores;
    p.B = B; p.N = N; p.q = q; p.C = C;
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
    p.max_sel = max_total;
    p.max_total = max_total;
    p.iou_thr = iou_threshold;
    p.score_thr = score_threshold;
    p.clip = clip_boxes;
    p.out_boxes = d_out_boxes; p.out_scores = d_out_scores; p.out_classes = nullptr;
    p.out_idx = d_out_idx; p.out_valid = d_out_valid;
    return launch_nms<true>(p, d_workspace, workspace_bytes, as_stream(stream));
}
