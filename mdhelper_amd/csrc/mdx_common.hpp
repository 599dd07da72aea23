// mdx_common.hpp — shared host-side plumbing of libmdx.so (errors, device guard, timers).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mdx.h"

namespace mdx {

// thread-local message behind mdx_last_error()
char *error_buffer();
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define MDX_HIP(expr)                                                                   \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            int _c = (_e == hipErrorOutOfMemory) ? MDX_ERR_OUT_OF_MEMORY                \
                     : (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice)          \
                         ? MDX_ERR_NO_DEVICE                                            \
                         : MDX_ERR_HIP;                                                 \
            return ::mdx::fail(_c, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                               __FILE__, __LINE__);                                     \
        }                                                                               \
    } while (0)

#define MDX_TRY(expr)             \
    do {                          \
        int _rc = (expr);         \
        if (_rc != MDX_OK)        \
            return _rc;           \
    } while (0)

#define MDX_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond))                                            \
            return ::mdx::fail(MDX_ERR_INVALID_VALUE, __VA_ARGS__); \
    } while (0)

int set_device(int dev);

// grow-only device buffer owned by a handle
struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    int ensure(size_t need);   // reallocates (contents lost) when need > bytes
    void release();
    template <typename T> T *as() const { return static_cast<T *>(ptr); }
};

// HIP-event timer on one stream: accumulates the device time of bracketed regions
struct StreamTimer {
    bool enabled = false;
    hipStream_t stream = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<hipEvent_t> pool;
    double total_ms = 0.0;
    int64_t launches = 0;
    hipEvent_t begin();
    void end(hipEvent_t start);
    void collect();            // call after the stream has been synchronised
    void reset();
    void destroy();
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace mdx
