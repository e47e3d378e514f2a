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
