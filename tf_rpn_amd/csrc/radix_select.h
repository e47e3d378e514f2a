// radix_select.h -- workgroup-wide radix select over 64-bit keys that are recomputed on every pass
// (never stored): finds a threshold `thr` so that the keys in [thr, hi_bound) are the `target` largest ones,
// or up to `cap` of them when a whole digit bin fits (saves passes).  Used by the NMS kernel (bands of the best
// <= 4096 candidates) and by the training-target kernel (exact top-K by random priority: cap == target).
//
// Keys are unique and non-zero for candidates, 0 for non-candidates.  11-bit digits, most significant first;
// one pass over the N keys (LDS histogram with atomics) plus one wave-level suffix scan per digit level.
// All threads of the workgroup must call it (it contains barriers); results are workgroup-uniform.
#pragma once
#include <hip/hip_runtime.h>

namespace rpn {

constexpr int kRsDigitBits = 11;
constexpr int kRsBins = 1 << kRsDigitBits;
// The histogram is kept in kRsCopies interleaved copies (bin d of copy c at [d * kRsCopies + c], c = lane % kRsCopies):
// float scores concentrate in a handful of top-digit bins (sign + exponent + 2 mantissa bits), and ~8600 LDS atomics
// on ~10 addresses serialise (measured 8.4 us per pass with one copy).
constexpr int kRsCopies = 4;
constexpr int kRsHistWords = kRsBins * kRsCopies;

// hist: kRsHistWords unsigned words of LDS; ctl: 4 ints of LDS.  Returns thr (0 = no key below hi_bound);
// *count = number of keys in [thr, hi_bound).
template <int NT, class KeyFn>
__device__ unsigned long long radix_select(KeyFn key_of, int N, unsigned long long hi_bound, int target, int cap,
                                           unsigned *hist, int *ctl, int *count)
{
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned long long thr = 1ull, prefix = 0ull;
    int above = 0;                          // keys strictly above the current prefix's bin range
    int shift = 64;                         // bits [shift, 64) are fixed by `prefix`
    bool narrowed = true;
    while (narrowed && shift > 0) {
        const int bits = shift >= kRsDigitBits ? kRsDigitBits : shift;
        const int dshift = shift - bits;
        for (int i = tid; i < kRsHistWords; i += NT) hist[i] = 0u;
        __syncthreads();
        // keys in batches of 8 per thread: the 8 global loads behind key_of are issued together (one at a time, each
        // iteration waited a full L2 round trip in front of its LDS atomic: ~7 us per pass at N = 8649)
        for (int base = tid; base < N; base += 8 * NT) {
            unsigned long long kb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kb[u] = (base + u * NT < N) ? key_of(base + u * NT) : 0ull;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned long long key = kb[u];
                if (key != 0ull && key < hi_bound && (shift == 64 || (key >> shift) == (prefix >> shift)))
                    atomicAdd(&hist[((unsigned)(key >> dshift) & ((1u << bits) - 1u)) * kRsCopies + (lane & (kRsCopies - 1))], 1u);
            }
        }
        __syncthreads();
        for (int i = tid; i < kRsBins; i += NT) {                 // fold the copies into copy 0
            unsigned sum = 0u;
#pragma unroll
            for (int c = 0; c < kRsCopies; ++c) sum += hist[i * kRsCopies + c];
            hist[i * kRsCopies] = sum;
        }
        __syncthreads();
        if (tid < 64) {                     // wave 0: suffix sums, top digit first; find where the count reaches `want`
            const int nb = 1 << bits;
            const int per = (nb + 63) / 64;                       // bins per lane, lane 0 = top bins
            const int top = nb - 1 - lane * per;
            unsigned mine = 0u;
            for (int k = 0; k < per; ++k)
                if (top - k >= 0) mine += hist[(top - k) * kRsCopies];
            unsigned incl = mine;
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned v = __shfl_up(incl, off, 64);
                if (lane >= off) incl += v;
            }
            const unsigned total = __shfl(incl, 63, 64);
            const int want = target - above;
            const unsigned excl = incl - mine;
            const bool cross = (int)excl < want && (int)incl >= want;
            const unsigned long long bal = __ballot(cross);
            if (total == 0u) {
                if (lane == 0) { ctl[0] = -1; ctl[1] = 0; }
            } else if (bal == 0ull) {                             // fewer than `want` keys in total: take all
                if (lane == 0) { ctl[0] = -2; ctl[1] = (int)total; }
            } else if (cross) {
                unsigned run = excl;
                int d = top;
                for (int k = 0; k < per; ++k) {
                    d = top - k;
                    run += hist[d * kRsCopies];
                    if ((int)run >= want) break;
                }
                ctl[0] = d;
                ctl[1] = (int)hist[d * kRsCopies];
                ctl[2] = above + (int)(run - hist[d * kRsCopies]);            // keys above bin d (all levels)
            }
        }
        __syncthreads();
        const int digit = ctl[0];
        const int bin_count = ctl[1];
        if (digit == -1) {                   // no key below hi_bound at all
            thr = 0ull;
            narrowed = false;
        } else if (digit == -2) {            // everything within the prefix fits the target: take all of it
            thr = (shift == 64) ? 1ull : (prefix & ~((1ull << shift) - 1ull));
            if (thr == 0ull) thr = 1ull;
            above += bin_count;
            narrowed = false;
        } else {
            above = ctl[2];
            prefix = (shift == 64 ? 0ull : (prefix & ~((1ull << shift) - 1ull))) |
                     ((unsigned long long)(unsigned)digit << dshift);
            shift = dshift;
            thr = prefix;                    // accept the whole bin `digit` ...
            if (above + bin_count <= cap || shift == 0) {
                above += bin_count;
                narrowed = false;            // ... if it still fits the cap; otherwise refine inside it
            }
        }
        __syncthreads();
    }
    if (thr == 0ull && above != 0) thr = 1ull;
    *count = above;
    return thr;
}

}  // namespace rpn
