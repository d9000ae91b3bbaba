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
// Two HIP events per kernel launch on the launch stream.  Completed pairs are folded into per-kernel totals as the
// run goes (the queue of pending pairs stays short: with thousands of live events the HIP runtime's own event
// bookkeeping made every later launch slower), so the totals are exact sums over all launches since enable(1).
#include <deque>
#include <vector>
namespace a3r {
struct ProfRec { int kernel; double work, bytes; hipEvent_t e0, e1; };
struct ProfTotal { long launches = 0; double ms = 0, work = 0, bytes = 0; };
static bool g_prof_on = false;
static std::deque<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static ProfTotal g_tot[PK_COUNT];
static const char* g_names[PK_COUNT] = {"gemm_kernel<0> (linear)", "gemm_kernel<1> (conv3x3)", "attn_kernel", "layernorm_kernel",
                                        "elementwise (patchify/upsample/head_final/pack)", "align_main_kernel",
                                        "align_finalize/prep kernels", "gemm_bf3_kernel (linear, split-bf16 MFMA)",
                                        "split_bf3_kernel", "gemm_bf3_kernel<1> (conv3x3, split-bf16 MFMA)",
                                        "attn_bf3_kernel", "gemm_fh2_kernel (linear, split-fp16 MFMA)", "attn_fh2_kernel",
                                        "gemm_fh2_kernel<1> (conv3x3, split-fp16 MFMA)"};
static hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
static void fold(const ProfRec& r) {
    float t = 0;
    if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) {
        g_tot[r.kernel].launches++; g_tot[r.kernel].ms += t; g_tot[r.kernel].work += r.work; g_tot[r.kernel].bytes += r.bytes;
    }
    g_pool.push_back(r.e0); g_pool.push_back(r.e1);
}
static void drain(bool wait) {
    while (!g_recs.empty()) {
        const ProfRec& r = g_recs.front();
        if (wait) (void)hipEventSynchronize(r.e1);
        else if (hipEventQuery(r.e1) != hipSuccess) break;
        fold(r);
        g_recs.pop_front();
    }
}
bool prof_enabled() { return g_prof_on; }
void prof_begin(int kernel, double work, hipStream_t st, double bytes) {
    if (g_recs.size() >= 128) drain(false);
    ProfRec r{kernel, work, bytes, get_event(), get_event()};
    (void)hipEventRecord(r.e0, st);
    g_recs.push_back(r);
}
void prof_end(hipStream_t st) { (void)hipEventRecord(g_recs.back().e1, st); }
}  // namespace a3r

extern "C" int a3r_prof_enable(int on) {
    using namespace a3r;
    if (on) {
        drain(true);
        for (auto& t : g_tot) t = ProfTotal();
    }
    g_prof_on = on != 0;
    return A3R_OK;
}
extern "C" int a3r_prof_kernel_count(void) { return a3r::PK_COUNT; }
extern "C" int a3r_prof_get(int kernel, const char** name, long* launches, double* total_ms, double* total_work) {
    using namespace a3r;
    A3R_CHECK_ARG(kernel >= 0 && kernel < PK_COUNT && name && launches && total_ms && total_work, "a3r_prof_get: bad argument");
    drain(true);
    *name = g_names[kernel];
    *launches = g_tot[kernel].launches; *total_ms = g_tot[kernel].ms; *total_work = g_tot[kernel].work;
    return A3R_OK;
}
extern "C" int a3r_prof_get_bytes(int kernel, double* total_bytes) {
    using namespace a3r;
    A3R_CHECK_ARG(kernel >= 0 && kernel < PK_COUNT && total_bytes, "a3r_prof_get_bytes: bad argument");
    drain(true);
    *total_bytes = g_tot[kernel].bytes;
    return A3R_OK;
}
