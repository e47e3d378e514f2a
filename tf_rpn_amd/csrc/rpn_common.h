// rpn_common.h -- shared host-side helpers of librpn_hip.so (error state, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "../../include/rpn_hip.h"
#include "rpn_knobs.h"

namespace rpn {

// thread-local message returned by rpn_last_error()
char *error_buffer();
constexpr int kErrorBufferLen = 512;

inline int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), kErrorBufferLen, fmt, ap);
    va_end(ap);
    return code;
}

// true when a HIP device is usable; fills the error buffer otherwise (no CPU fallback exists)
bool have_device();

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }


}  // namespace rpn

#define RPN_REQUIRE(cond, ...)                                         \
    do {                                                               \
        if (!(cond)) return rpn::fail(RPN_ERR_INVALID, __VA_ARGS__);   \
    } while (0)

#define RPN_REQUIRE_DEVICE()                                           \
    do {                                                               \
        if (!rpn::have_device()) return RPN_ERR_NO_DEVICE;             \
    } while (0)

#define RPN_HIP_CHECK(expr)                                                                      \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return rpn::fail(RPN_ERR_NO_DEVICE, "%s failed: %s (%s:%d)", #expr,                  \
                             hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)

#define RPN_CHECK_LAUNCH() RPN_HIP_CHECK(hipGetLastError())
