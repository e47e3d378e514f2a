// store_hazard.hip -- is a vector write to a buffer store's DATA registers safe right behind the store?  (round 6: hipcc leaves no
// wait state behind `buffer_store_dwordx4 ..., s54 offen` -- a REGISTER scalar offset -- and the first layer's kernel stored garbage.)
// Every wave, ITERS times: set the data registers to a good value, store them, N x s_nop 0, overwrite the data registers with a poison
// value.  The chip's store path is kept busy by 16 waves per CU doing the same.  Then the buffer is searched for the poison.
//   W = 1 / 4: buffer_store_dword / dwordx4;   SOFF = 0: scalar offset constant 0, 1: in an SGPR
// Build: hipcc --offload-arch=gfx950 -O3 -o ab/store_hazard scripts/micro/store_hazard.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int W, int SOFF, int N>
__global__ void __launch_bounds__(256) k(unsigned *out, int iters)
{
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, (short)0, 0x7ffffff0, 0x00020000);
    const unsigned vo = (unsigned)(lane * 4 * W);
    for (int it = 0; it < iters; ++it) {
        const unsigned so = __builtin_amdgcn_readfirstlane((wave * (unsigned)iters + (unsigned)it) * (256u * W));
        const unsigned good = 0x3f800000u + (unsigned)it;
        if constexpr (W == 4 && SOFF == 1)
            asm volatile("v_mov_b32 v10, %2\n v_mov_b32 v11, %2\n v_mov_b32 v12, %2\n v_mov_b32 v13, %2\n s_nop 4\n"
                         "buffer_store_dwordx4 v[10:13], %0, %1, %3 offen\n"
                         ".rept %4\n s_nop 0\n .endr\n"
                         "v_mov_b32 v10, 0x7fc0dead\n v_mov_b32 v11, 0x7fc0dead\n v_mov_b32 v12, 0x7fc0dead\n v_mov_b32 v13, 0x7fc0dead\n"
                         :: "v"(vo), "s"(rs), "v"(good), "s"(so), "n"(N) : "v10", "v11", "v12", "v13", "memory");
        if constexpr (W == 4 && SOFF == 0)
            asm volatile("v_mov_b32 v10, %2\n v_mov_b32 v11, %2\n v_mov_b32 v12, %2\n v_mov_b32 v13, %2\n v_add_u32 v14, %0, %3\n s_nop 4\n"
                         "buffer_store_dwordx4 v[10:13], v14, %1, 0 offen\n"
                         ".rept %4\n s_nop 0\n .endr\n"
                         "v_mov_b32 v10, 0x7fc0dead\n v_mov_b32 v11, 0x7fc0dead\n v_mov_b32 v12, 0x7fc0dead\n v_mov_b32 v13, 0x7fc0dead\n"
                         :: "v"(vo), "s"(rs), "v"(good), "s"(so), "n"(N) : "v10", "v11", "v12", "v13", "v14", "memory");
        if constexpr (W == 1 && SOFF == 1)
            asm volatile("v_mov_b32 v10, %2\n s_nop 4\n"
                         "buffer_store_dword v10, %0, %1, %3 offen\n"
                         ".rept %4\n s_nop 0\n .endr\n"
                         "v_mov_b32 v10, 0x7fc0dead\n"
                         :: "v"(vo), "s"(rs), "v"(good), "s"(so), "n"(N) : "v10", "memory");
        if constexpr (W == 1 && SOFF == 0)
            asm volatile("v_mov_b32 v10, %2\n v_add_u32 v14, %0, %3\n s_nop 4\n"
                         "buffer_store_dword v10, v14, %1, 0 offen\n"
                         ".rept %4\n s_nop 0\n .endr\n"
                         "v_mov_b32 v10, 0x7fc0dead\n"
                         :: "v"(vo), "s"(rs), "v"(good), "s"(so), "n"(N) : "v10", "v14", "memory");
    }
}

template <int W, int SOFF, int N>
static void run(unsigned *buf)
{
    const int grid = 1024, iters = 64;
    const size_t words = (size_t)grid * 4 * iters * 64 * W;
    (void)hipMemset(buf, 0, words * 4);
    k<W, SOFF, N><<<grid, 256>>>(buf, iters);
    std::vector<unsigned> h(words);
    (void)hipMemcpy(h.data(), buf, words * 4, hipMemcpyDeviceToHost);
    size_t poison = 0, other = 0;
    unsigned lanes = 0, elems = 0;
    for (size_t i = 0; i < words; ++i) {
        if (h[i] == 0x7fc0deadu) {
            ++poison;
            lanes |= 1u << ((i / W) % 16);
            elems |= 1u << (i % W);
        } else if ((h[i] & 0xffffff00u) != 0x3f800000u)
            ++other;
    }
    printf("store x%d, scalar offset %s, %d wait state(s) before the overwrite: %8zu poisoned words of %zu (%s)  lanes&15 mask %04x element mask %x  other %zu\n", W,
           SOFF ? "in an SGPR " : "constant 0 ", N, poison, words, poison ? "HAZARD" : "ok", lanes, elems, other);
}

int main()
{
    unsigned *buf;
    if (hipMalloc(&buf, (size_t)1 << 30) != hipSuccess) return 1;
    run<4, 1, 0>(buf); run<4, 1, 1>(buf); run<4, 1, 2>(buf); run<4, 1, 3>(buf); run<4, 1, 4>(buf);
    run<4, 0, 0>(buf); run<4, 0, 1>(buf); run<4, 0, 2>(buf);
    run<1, 1, 0>(buf); run<1, 1, 1>(buf); run<1, 0, 0>(buf); run<1, 0, 1>(buf);
    return 0;
}
