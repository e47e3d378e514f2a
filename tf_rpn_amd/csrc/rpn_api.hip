// rpn_api.hip -- library-wide pieces of the C ABI: version, error state, device probe.
#include "rpn_common.h"

namespace rpn {

char *error_buffer()
{
    static thread_local char buf[kErrorBufferLen] = {0};
    return buf;
}

bool have_device()
{
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        fail(RPN_ERR_NO_DEVICE, "no HIP device available (%s); librpn_hip has no CPU fallback",
             e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return false;
    }
    return true;
}

}  // namespace rpn

extern "C" int rpn_abi_version(void) { return RPN_ABI_VERSION; }

extern "C" const char *rpn_last_error(void) { return rpn::error_buffer(); }

extern "C" int rpn_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// One wave that sleeps until `ticks` of the 100 MHz real-time counter have passed (bounded: at most 10 ms).
namespace rpn {
__global__ void __launch_bounds__(64) stream_spin_kernel(unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace rpn

extern "C" int rpn_stream_spin(void *stream, int microseconds)
{
    if (microseconds < 0 || microseconds > 10000) return rpn::fail(RPN_ERR_INVALID, "rpn_stream_spin: 0 .. 10000 microseconds");
    if (!rpn::have_device()) return RPN_ERR_NO_DEVICE;
    hipLaunchKernelGGL(rpn::stream_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)microseconds * 100ull);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? RPN_OK : rpn::fail(RPN_ERR_NO_DEVICE, "rpn_stream_spin: %s", hipGetErrorString(e));
}
