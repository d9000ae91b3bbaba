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

// Row sums of up to 16 per-lane values by a DPP reduce-scatter (32 VALU operations for 16 values, 29 for 13, instead of four
// butterfly steps per value): each level halves the number of values a lane carries -- the lanes with bit 3 (then bit 2) of the
// lane index clear keep the lower half of the indices, the others the upper half, and add what their partner lane (lane ^ 8, then
// lane ^ 4) holds of the half they keep.  The two halves of a level are written by two v_add_f32_dpp whose bank_mask enables
// only the lanes that keep that half (banks = lanes 4b .. 4b + 3 of a row), so no select instructions are needed; the last two
// levels (inside a quad, where bank masks cannot select) are plain butterflies on the remaining four values.
// On return every lane of quad q = (lane >> 2) & 3 of a row holds in u[0..3] the row sums of values 4 q + 0..3.
// NREAL = 13: values 13..15 do not exist; u[1..3] of quad 3 then hold copies of the sums 5..7 (callers ignore them).
// Inline assembly because the compiler does not fold a bank-masked DPP move into the addition; the s_nop satisfy the two wait
// states a DPP source needs after the VALU write of that register (the hazard recogniser does not look into the block).
template <int NREAL>
__device__ __forceinline__ void row_reduce_scatter16(const float (&v)[16], float (&u)[4]) {
    static_assert(NREAL == 13 || NREAL == 16, "13 or 16 values");
    float t0, t1, t2, t3, t4, t5, t6, t7;
#define A3R_RS_LO(T, A) "v_add_f32_dpp " T ", " A ", " A " row_ror:8 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
#define A3R_RS_HI(T, A) "v_add_f32_dpp " T ", " A ", " A " row_ror:8 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
#define A3R_RS_ALL(T, A) "v_add_f32_dpp " T ", " A ", " A " row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    if constexpr (NREAL == 16) {
        asm volatile("s_nop 1\n\t"
                     A3R_RS_LO("%0", "%8") A3R_RS_HI("%0", "%16") A3R_RS_LO("%1", "%9") A3R_RS_HI("%1", "%17")
                     A3R_RS_LO("%2", "%10") A3R_RS_HI("%2", "%18") A3R_RS_LO("%3", "%11") A3R_RS_HI("%3", "%19")
                     A3R_RS_LO("%4", "%12") A3R_RS_HI("%4", "%20") A3R_RS_LO("%5", "%13") A3R_RS_HI("%5", "%21")
                     A3R_RS_LO("%6", "%14") A3R_RS_HI("%6", "%22") A3R_RS_LO("%7", "%15") A3R_RS_HI("%7", "%23")
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                     : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]),
                       "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
    } else {
        asm volatile("s_nop 1\n\t"
                     A3R_RS_LO("%0", "%8") A3R_RS_HI("%0", "%16") A3R_RS_LO("%1", "%9") A3R_RS_HI("%1", "%17")
                     A3R_RS_LO("%2", "%10") A3R_RS_HI("%2", "%18") A3R_RS_LO("%3", "%11") A3R_RS_HI("%3", "%19")
                     A3R_RS_LO("%4", "%12") A3R_RS_HI("%4", "%20") A3R_RS_ALL("%5", "%13") A3R_RS_ALL("%6", "%14") A3R_RS_ALL("%7", "%15")
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                     : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]),
                       "v"(v[10]), "v"(v[11]), "v"(v[12]));
    }
#undef A3R_RS_LO
#undef A3R_RS_HI
#undef A3R_RS_ALL
    // bit 2: lanes with the bit clear (banks 0, 2) keep t0..t3 and read lane + 4 (row_ror:12 = rotate right by 12), the others
    // (banks 1, 3) keep t4..t7 and read lane - 4
#define A3R_RS_LO(T, A) "v_add_f32_dpp " T ", " A ", " A " row_ror:12 row_mask:0xf bank_mask:0x5 bound_ctrl:1\n\t"
#define A3R_RS_HI(T, A) "v_add_f32_dpp " T ", " A ", " A " row_ror:4 row_mask:0xf bank_mask:0xa bound_ctrl:1\n\t"
    asm volatile("s_nop 1\n\t"
                 A3R_RS_LO("%0", "%4") A3R_RS_HI("%0", "%8") A3R_RS_LO("%1", "%5") A3R_RS_HI("%1", "%9")
                 A3R_RS_LO("%2", "%6") A3R_RS_HI("%2", "%10") A3R_RS_LO("%3", "%7") A3R_RS_HI("%3", "%11")
                 "s_nop 1"
                 : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3])
                 : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4), "v"(t5), "v"(t6), "v"(t7));
#undef A3R_RS_LO
#undef A3R_RS_HI
#pragma unroll
    for (int k = 0; k < 4; k++) {
        u[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u[k]), 0xB1, 0xF, 0xF, true));   // lane ^ 1
        u[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u[k]), 0x4E, 0xF, 0xF, true));   // lane ^ 2
    }
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
