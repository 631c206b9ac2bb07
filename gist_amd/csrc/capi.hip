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
}  // namespace gist

extern "C" const char *gist_last_error(void) { return gist::g_err; }
extern "C" int gist_abi_version(void) { return 5; }
extern "C" int gist_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        gist::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return 0;
    }
    return n;
}
