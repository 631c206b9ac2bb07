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

// HIP-event timer of the native step driver (step.hip).  While gist_sage_step runs with a timer
// armed, tl_timer points at it so that a kernel below an entry point (the split GEMM's main
// kernel) can be bracketed on its own: slot = timer_begin(...); launch; timer_end(slot).
extern thread_local gist_timer *tl_timer;
int64_t timer_begin(gist_timer *t, int kind, int64_t m, int64_t n, int64_t k, hipStream_t s);
void timer_end(gist_timer *t, int64_t slot, hipStream_t s);

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned8(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace gist
