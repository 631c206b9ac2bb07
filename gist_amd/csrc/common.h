// Shared helpers for libgist_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gist_hip.h"

namespace gist {

constexpr int kWave = 64;

void set_error(const char *fmt, ...);

static inline hipStream_t as_stream(gist_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Checks the launch itself (not completion): calls stay asynchronous.
static inline int launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return GIST_ELAUNCH;
    }
    return GIST_OK;
}

#define GIST_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            gist::set_error(__VA_ARGS__);  \
            return GIST_EINVAL;            \
        }                                  \
    } while (0)

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned8(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace gist
