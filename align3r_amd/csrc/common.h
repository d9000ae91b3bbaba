// Shared helpers for liba3r (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <atomic>
#include "../../include/a3r.h"

namespace a3r {

void set_error(const char* fmt, ...);

#define A3R_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            a3r::set_error(__VA_ARGS__);         \
            return A3R_EINVAL;                   \
        }                                        \
    } while (0)

#define A3R_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t err__ = (call);                                                             \
        if (err__ != hipSuccess) {                                                             \
            a3r::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, __LINE__); \
            return A3R_EHIP;                                                                   \
        }                                                                                      \
    } while (0)

#define A3R_LAUNCH_CHECK()                                                                      \
    do {                                                                                        \
        hipError_t err__ = hipGetLastError();                                                   \
        if (err__ != hipSuccess) {                                                              \
            a3r::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(err__), __FILE__, __LINE__); \
            return A3R_EHIP;                                                                    \
        }                                                                                       \
    } while (0)

// ---- optional per-kernel timing with HIP events on the launch stream (a3r_prof_* in include/a3r.h)
enum ProfKernel { PK_LINEAR = 0, PK_CONV, PK_ATTENTION, PK_LAYERNORM, PK_ELEMENTWISE, PK_ALIGN_MAIN, PK_ALIGN_SMALL, PK_LINEAR_BF3, PK_SPLIT, PK_CONV_BF3, PK_ATTENTION_BF3, PK_LINEAR_FH2, PK_ATTENTION_FH2, PK_CONV_FH2, PK_COUNT };
bool prof_enabled();
void prof_begin(int kernel, double work, hipStream_t st, double bytes = 0);
void prof_end(hipStream_t st);
struct ProfScope {
    hipStream_t st; bool on;
    // work: algorithmic FLOP (MFMA kernels) or bytes (HBM-bound kernels); bytes: algorithmic operand + result bytes of an MFMA kernel
    // (each operand read once, each result written once), so that PMC traffic / algorithmic bytes can be read off the bench line
    ProfScope(int kernel, double work, hipStream_t s, double bytes = 0) : st(s), on(prof_enabled()) { if (on) prof_begin(kernel, work, st, bytes); }
    ~ProfScope() { if (on) prof_end(st); }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to the CURRENT DEVICE's function object: a process that builds engines
// on several GPUs must opt in on each.  One atomic bit per device and call site, set only AFTER the attribute call succeeded: a
// second host thread racing the first launch on a device sees the bit clear and sets the (idempotent) attribute itself instead of
// launching a kernel whose opt-in has not happened yet; a failed call is retried on the next launch.
struct PerDeviceOnce {
    std::atomic<unsigned long long> done{0};
    // runs `setup` (-> hipError_t) unless it already succeeded on the current device
    template <class F> hipError_t ensure(F&& setup) {
        int dev = 0;
        const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
        const unsigned long long bit = known ? 1ull << dev : 0;
        if (known && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
        const hipError_t e = setup();
        if (e == hipSuccess && known) done.fetch_or(bit, std::memory_order_release);
        return e;
    }
};

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// sum over each row of 16 lanes with DPP (every lane of the row ends with the row sum)
__device__ __forceinline__ float dpp_row_sum16(float v) {
#define A3R_DPP_ADD(ctrl) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
    A3R_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    A3R_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    A3R_DPP_ADD(0x141);  // row_half_mirror
    A3R_DPP_ADD(0x140);  // row_mirror
#undef A3R_DPP_ADD
    return v;
}

// value of the neighbouring lane (lane ^ 1) by DPP quad_perm [1,0,3,2]: one VALU op, no LDS crossbar (ds_bpermute)
__device__ __forceinline__ float dpp_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// full 64-lane sum (result valid in every lane)
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_row_sum16(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

}  // namespace a3r
