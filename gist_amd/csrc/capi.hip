// Error plumbing + version for the C ABI (include/gist_hip.h).
#include <stdarg.h>

#include "common.h"

namespace gist {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
std::atomic<uint64_t> g_launches{0};
__global__ void empty_kernel() {}
static std::atomic<double> g_tune[GIST_TUNE_COUNT];
double tune(int knob) { return g_tune[knob].load(std::memory_order_relaxed); }
}  // namespace gist

extern "C" int gist_tuning_set(int knob, double value) {
    GIST_REQUIRE(knob >= 0 && knob < GIST_TUNE_COUNT, "gist_tuning_set: unknown knob %d", knob);
    GIST_REQUIRE(value >= 0.0, "gist_tuning_set: negative value");
    gist::g_tune[knob].store(value, std::memory_order_relaxed);
    return GIST_OK;
}
extern "C" double gist_tuning_get(int knob) {
    return knob >= 0 && knob < GIST_TUNE_COUNT ? gist::tune(knob) : -1.0;
}

extern "C" uint64_t gist_launch_count(void) { return gist::g_launches.load(std::memory_order_relaxed); }
extern "C" int gist_empty_launches(int32_t n, int32_t grid, int32_t block, gist_stream_t stream) {
    GIST_REQUIRE(n >= 0 && grid >= 1 && block >= 1 && block <= 1024, "gist_empty_launches: bad arguments");
    for (int32_t i = 0; i < n; ++i)
        hipLaunchKernelGGL(gist::empty_kernel, dim3((unsigned)grid), dim3((unsigned)block), 0, gist::as_stream(stream));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gist::set_error("gist_empty_launches: %s", hipGetErrorString(e));
        return GIST_ELAUNCH;
    }
    return GIST_OK;
}

extern "C" const char *gist_last_error(void) { return gist::g_err; }
extern "C" int gist_abi_version(void) { return 16; }
extern "C" int gist_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        gist::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return 0;
    }
    return n;
}
