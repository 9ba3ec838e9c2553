// Shared plumbing of libnerfool_hip.so: error reporting, launch helpers, the host/device function marker.
#pragma once

#include <stdint.h>
#include <math.h>

#include <hip/hip_runtime.h>
#define NF_HD __host__ __device__ __forceinline__
#define NF_WAVE 64

#include "../../include/nerfool_hip.h"

void nf_set_error(const char* fmt, ...);

#define NF_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            nf_set_error(__VA_ARGS__);        \
            return 1;                         \
        }                                     \
    } while (0)

#define NF_LAUNCH_CHECK(what)                                              \
    do {                                                                   \
        hipError_t nf_e_ = hipGetLastError();                              \
        if (nf_e_ != hipSuccess) {                                         \
            nf_set_error("%s: %s", what, hipGetErrorString(nf_e_));        \
            return 2;                                                      \
        }                                                                  \
    } while (0)

// Per-device "already configured" flags (hipFuncSetAttribute is a per-device setting: a process that touches a second GPU
// must opt in to the large dynamic LDS there too).  The host layer makes the tensor's device current before every launch.
#define NF_MAX_DEVICES 16
static inline int nf_current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= NF_MAX_DEVICES) d = 0;
    return d;
}

// 1 / x as ONE vector instruction (v_rcp_f32, 1 ulp) where the result feeds continuous arithmetic; x / y stays an IEEE division (a
// ten-instruction sequence) wherever it decides a mask, an index or a bin
NF_HD float nf_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.f / x;
#endif
}

static inline unsigned nf_blocks(int64_t work, int per_block) { return (unsigned)((work + per_block - 1) / per_block); }

// wave64 butterflies (all 64 lanes must call)
__device__ __forceinline__ float nf_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, NF_WAVE);
    return v;
}
__device__ __forceinline__ float nf_wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, NF_WAVE));
    return v;
}

// ---- partner-lane exchange without the LDS crossbar ---------------------------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32 + s_waitcnt lgkmcnt: an LDS round trip (~100+ cycles) per butterfly step, and the wait
// also drains every A-operand read queued behind it.  Inside a row of 16 lanes the data-parallel-primitive modifiers do the same
// exchange in the adding instruction itself (v_add_f32_dpp: one VALU slot, no memory latency); the two wave halves swap through
// gfx950's v_permlane32_swap.
template <int CTRL>
__device__ __forceinline__ float nf_dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
#define NF_DPP_XOR1 0xB1           // quad_perm:[1,0,3,2]
#define NF_DPP_XOR2 0x4E           // quad_perm:[2,3,0,1]
#define NF_DPP_HALF_MIRROR 0x141   // lane i <-> 7 - i of its group of 8
#define NF_DPP_MIRROR 0x140        // lane i <-> 15 - i of its row of 16

// butterfly reduction over groups of V adjacent lanes (V a power of two <= 32); every lane gets the result.  Steps 4 and 8 use
// the mirror patterns: after the quad steps all lanes of a quad (of a group of 8) hold the same partial, so "7 - i" / "15 - i"
// delivers the same value as "i ^ 4" / "i ^ 8" -- the result is bit-identical to the xor butterfly.
struct NfAdd { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };
struct NfMin { __device__ __forceinline__ float operator()(float a, float b) const { return fminf(a, b); } };
struct NfMax { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };
template <int V, typename Op>
__device__ __forceinline__ float nf_grp_reduce(float x, Op op) {
    if (V >= 2) x = op(x, nf_dpp<NF_DPP_XOR1>(x));
    if (V >= 4) x = op(x, nf_dpp<NF_DPP_XOR2>(x));
    if (V >= 8) x = op(x, nf_dpp<NF_DPP_HALF_MIRROR>(x));
    if (V >= 16) x = op(x, nf_dpp<NF_DPP_MIRROR>(x));
    if (V >= 32) x = op(x, __shfl_xor(x, 16, NF_WAVE));
    return x;
}
// v_permlane32_swap vdst, src: lanes 32-63 of vdst swap with lanes 0-31 of src.  With both operands = x the results are
// (x of the low half in every lane, x of the high half in every lane).  (Take the results through a NAMED vector type and .x / .y:
// with `auto r` + r[1] this hipcc reads element 0 twice.)
typedef unsigned nf_u32x2 __attribute__((ext_vector_type(2)));
struct NfHalves { float lo, hi; };
__device__ __forceinline__ NfHalves nf_halves(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const nf_u32x2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return NfHalves{__builtin_bit_cast(float, (unsigned)r.x), __builtin_bit_cast(float, (unsigned)r.y)};
}
// x(lane ^ 32)
__device__ __forceinline__ float nf_half_other(float x) {
    const NfHalves v = nf_halves(x);
    return (threadIdx.x & 32) ? v.lo : v.hi;
}
// max(x(lane), x(lane ^ 32))
__device__ __forceinline__ float nf_half_max(float x) {
    const NfHalves v = nf_halves(x);
    return fmaxf(v.lo, v.hi);
}
// x(lane) + x(lane ^ 32)
__device__ __forceinline__ float nf_half_sum(float x) {
    const NfHalves v = nf_halves(x);
    return v.lo + v.hi;
}

// ---- the bf16x3 operand split (DESIGN "fp32-grade products on the bf16 matrix cores"): two fp32 values -> their bf16 roundings, packed for
//      the matrix instruction, and the exact remainders in place.  The packed result passes through an EMPTY asm statement: from the plain
//      C++ form (`u << 16` / `u & 0xffff0000` of the packed pair) the compiler derived x0's rounding a second time with its own
//      v_cvt_pk_bf16_f32 (6 vector instructions per pair and part instead of 5: 32 of a Winograd chunk's ~350).  (The conversion itself
//      must stay the compiler's instruction: written as inline asm it reads accumulator registers of a matrix instruction still in flight
//      -- the hazard recogniser does not look into asm -- and the sample-on-the-lane kernels, which split accumulators directly, were 3 %
//      off on the hardware.)
__device__ __forceinline__ unsigned nf_split_pair_bf16(float& x0, float& x1) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __bf16 nf_bf16x2 __attribute__((ext_vector_type(2)));
    const nf_bf16x2 pr = nf_bf16x2{(__bf16)x0, (__bf16)x1};          // v_cvt_pk_bf16_f32 (round to nearest even), x0 in the low half
    unsigned u = __builtin_bit_cast(unsigned, pr);
#if !defined(NF_SPLIT_CXX)
    asm("" : "+v"(u));
#endif
    x0 -= __builtin_bit_cast(float, u << 16);
    x1 -= __builtin_bit_cast(float, u & 0xffff0000u);
    return u;
#else
    const __bf16 b0 = (__bf16)x0, b1 = (__bf16)x1;
    unsigned short s0, s1;
    __builtin_memcpy(&s0, &b0, 2);
    __builtin_memcpy(&s1, &b1, 2);
    x0 -= (float)b0;
    x1 -= (float)b1;
    return (unsigned)s0 | ((unsigned)s1 << 16);
#endif
}
