// Error plumbing + version for liba3r (see include/a3r.h).
#include "common.h"
#include <cstring>

namespace a3r {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace a3r

extern "C" const char* a3r_last_error(void) { return a3r::g_err; }
extern "C" int a3r_version(void) { return 100; }
extern "C" int a3r_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
