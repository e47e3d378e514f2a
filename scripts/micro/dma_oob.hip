// Micro-test: does an out-of-range lane of `buffer_load_dwordx4 ... lds` write zeros into LDS (or leave it alone)?
// hipcc --offload-arch=gfx950 -O3 dma_oob.hip -o dma_oob && ./dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
__global__ void k(const uint4* __restrict__ g, uint4* out, int n)
{
    __shared__ uint4 lds[256];
    const int tid = threadIdx.x, wave = tid >> 6;
    lds[tid] = make_uint4(0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(g), (short)0, n * 16, 0x00020000);
    unsigned off = tid * 16u;
    if (tid % 3 == 1) off = 0x80000000u;        // far out of range
    if (tid % 3 == 2) off = (unsigned)(n * 16) + tid * 16u;   // just past the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(lds + wave * 64), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)");
    __syncthreads();
    out[tid] = lds[tid];
}
int main()
{
    const int n = 256;
    std::vector<uint4> h(n);
    for (int i = 0; i < n; ++i) h[i] = make_uint4(i + 1, i + 1, i + 1, i + 1);
    uint4 *g, *o;
    hipMalloc(&g, n * 16); hipMalloc(&o, n * 16);
    hipMemcpy(g, h.data(), n * 16, hipMemcpyHostToDevice);
    k<<<1, 256>>>(g, o, n);
    std::vector<uint4> r(n);
    if (hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL hip\n"); return 1; }
    int ok_in = 0, zero_far = 0, zero_near = 0, kept_far = 0, kept_near = 0;
    for (int i = 0; i < n; ++i) {
        if (i % 3 == 0) ok_in += (r[i].x == (unsigned)i + 1 && r[i].w == (unsigned)i + 1);
        if (i % 3 == 1) { zero_far += (r[i].x == 0 && r[i].w == 0); kept_far += (r[i].x == 0xdeadbeefu); }
        if (i % 3 == 2) { zero_near += (r[i].x == 0 && r[i].w == 0); kept_near += (r[i].x == 0xdeadbeefu); }
    }
    printf("in-range correct %d/86  far-OOB: zero %d kept %d /85  near-OOB: zero %d kept %d /85\n", ok_in, zero_far, kept_far, zero_near, kept_near);
    return 0;
}
