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

// ------------------------------------------------------------------------------------------- profiling
#include <vector>
namespace a3r {
struct ProfRec { int kernel; double work; hipEvent_t e0, e1; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static const char* g_names[PK_COUNT] = {"gemm_kernel<0> (linear)", "gemm_kernel<1> (conv3x3)", "attn_kernel", "layernorm_kernel",
                                        "elementwise (patchify/upsample/head_final/pack)", "align_main_kernel",
                                        "align_finalize/prep kernels"};
static hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
bool prof_enabled() { return g_prof_on; }
void prof_begin(int kernel, double work, hipStream_t st) {
    ProfRec r{kernel, work, get_event(), get_event()};
    (void)hipEventRecord(r.e0, st);
    g_recs.push_back(r);
}
void prof_end(hipStream_t st) { (void)hipEventRecord(g_recs.back().e1, st); }
}  // namespace a3r

extern "C" int a3r_prof_enable(int on) {
    using namespace a3r;
    if (on) {
        for (auto& r : g_recs) { g_pool.push_back(r.e0); g_pool.push_back(r.e1); }
        g_recs.clear();
    }
    g_prof_on = on != 0;
    return A3R_OK;
}
extern "C" int a3r_prof_kernel_count(void) { return a3r::PK_COUNT; }
extern "C" int a3r_prof_get(int kernel, const char** name, long* launches, double* total_ms, double* total_work) {
    using namespace a3r;
    A3R_CHECK_ARG(kernel >= 0 && kernel < PK_COUNT && name && launches && total_ms && total_work, "a3r_prof_get: bad argument");
    *name = g_names[kernel];
    long n = 0;
    double ms = 0, work = 0;
    for (auto& r : g_recs) {
        if (r.kernel != kernel) continue;
        A3R_HIP(hipEventSynchronize(r.e1));
        float t = 0;
        A3R_HIP(hipEventElapsedTime(&t, r.e0, r.e1));
        ms += t; work += r.work; n++;
    }
    *launches = n; *total_ms = ms; *total_work = work;
    return A3R_OK;
}
