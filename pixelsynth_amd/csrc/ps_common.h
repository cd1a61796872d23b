// Shared helpers for libpixelsynth_hip.so (error channel, launch checks).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/pixelsynth_hip.h"

namespace ps {

inline std::string &last_error_ref()
{
    static thread_local std::string e;
    return e;
}

inline int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

#define PS_HIP_CHECK(expr)                                                                     \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return ps::fail(PS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                            __FILE__, __LINE__);                                               \
    } while (0)

#define PS_LAUNCH_CHECK() PS_HIP_CHECK(hipGetLastError())

#define PS_REQUIRE(cond, ...)                                    \
    do {                                                         \
        if (!(cond)) return ps::fail(PS_ERR_ARG, __VA_ARGS__);   \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace ps
