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
