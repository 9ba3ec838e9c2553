// TEST INFRASTRUCTURE ONLY -- a minimal CPU stand-in for <hip/hip_runtime.h>.
//
// tests/host_harness/build.sh compiles the UNMODIFIED kernel sources of nerfool_amd/csrc/*.hip with g++ against this
// header into tests/host_harness/libnerfool_emu.so, so that the kernels' index arithmetic, wave-level scans, LDS
// staging and (emulated) MFMA fragment layouts can be parity-checked against the oracle on a machine without a GPU,
// through the same C ABI.  It is never loaded by the product (nerfool_amd/_lib.py refuses to run without the real
// gfx950 library) and implements only what those sources use.
//
// Execution model (hip_emu.cpp): the blocks of a launch are dealt to a few OS worker threads; the threads of a block are
// cooperative fibers on their worker; __syncthreads() and the wave-collective operations (__shfl*, MFMA) are barriers over the
// live fibers of the block / of the 64 consecutive threads of a wave.  Every live lane of a wave must therefore reach every
// collective (true on the GPU for the kernels in this repo as well).  __shared__ variables are per worker thread, i.e. per
// block in flight.
#pragma once

#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define HIP_DYNAMIC_SHARED(type, name) type* name = (type*)hip_emu::dynamic_smem();

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
enum { hipDeviceAttributeMultiprocessorCount = 1 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 256; return hipSuccess; }

namespace hip_emu {
extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
void* dynamic_smem();
uint64_t* wave_slots();          // the 64 exchange slots of the calling fiber's wave
void wave_barrier();
void block_barrier();
inline int lane() { return (int)(t_threadIdx.x & 63); }
inline int wave() { return (int)(t_threadIdx.x >> 6); }

// every lane publishes `v`, then reads the value of lane `src` (own value when src is out of range)
template <typename T>
inline T exchange(T v, int src) {
    static_assert(sizeof(T) <= 8, "exchange");
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    uint64_t* sl = wave_slots();
    sl[lane()] = raw;
    wave_barrier();
    uint64_t got = (src >= 0 && src < 64) ? sl[src] : raw;
    wave_barrier();
    T out;
    memcpy(&out, &got, sizeof(T));
    return out;
}
}  // namespace hip_emu

#define threadIdx hip_emu::t_threadIdx
#define blockIdx hip_emu::t_blockIdx
#define blockDim hip_emu::t_blockDim
#define gridDim hip_emu::t_gridDim

static inline void __syncthreads() { hip_emu::block_barrier(); }

template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; return hip_emu::exchange(v, hip_emu::lane() ^ mask); }
template <typename T>
static inline T __shfl_up(T v, unsigned d, int width = 64) { (void)width; int s = hip_emu::lane() - (int)d; return hip_emu::exchange(v, s < 0 ? hip_emu::lane() : s); }
template <typename T>
static inline T __shfl_down(T v, unsigned d, int width = 64) { (void)width; int s = hip_emu::lane() + (int)d; return hip_emu::exchange(v, s > 63 ? hip_emu::lane() : s); }
template <typename T>
static inline T __shfl(T v, int src, int width = 64) { (void)width; return hip_emu::exchange(v, src & 63); }

// data-parallel-primitive lane patterns (quad_perm 0x00-0xFF, row_half_mirror 0x141, row_mirror 0x140) and gfx950's
// v_permlane32_swap, as used by nf_common.h
static inline int __builtin_amdgcn_update_dpp(int, int src, int ctrl, int, int, bool) {
    const int l = hip_emu::lane();
    int from = l;
    if (ctrl <= 0xFF) from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
    else if (ctrl == 0x141) from = (l & ~7) | (7 - (l & 7));
    else if (ctrl == 0x140) from = (l & ~15) | (15 - (l & 15));
    else { fprintf(stderr, "hip_emu: dpp control 0x%x not emulated\n", ctrl); abort(); }
    return hip_emu::exchange(src, from);
}
typedef unsigned nf_emu_u32x2 __attribute__((ext_vector_type(2)));
// vdst[32:63] <-> vsrc[0:31]; returns (vdst', vsrc')
static inline nf_emu_u32x2 __builtin_amdgcn_permlane32_swap(unsigned a, unsigned b, bool, bool) {
    const int l = hip_emu::lane();
    const unsigned a_from = hip_emu::exchange(b, l - 32 < 0 ? l : l - 32);      // upper lanes of a' take b's lower half
    const unsigned b_from = hip_emu::exchange(a, l + 32 > 63 ? l : l + 32);      // lower lanes of b' take a's upper half
    nf_emu_u32x2 r;
    r.x = l >= 32 ? a_from : a;
    r.y = l < 32 ? b_from : b;
    return r;
}

static inline float atomicAdd(float* addr, float val) {
    uint32_t* p = (uint32_t*)addr;
    uint32_t old = __atomic_load_n(p, __ATOMIC_RELAXED);
    for (;;) {
        float f;
        memcpy(&f, &old, 4);
        f += val;
        uint32_t nw;
        memcpy(&nw, &f, 4);
        if (__atomic_compare_exchange_n(p, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
            memcpy(&f, &old, 4);
            return f;
        }
    }
}

static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
// device-scope fence (an acquire invalidates the non-coherent vector L1 on the device): a host fence here
static inline void __builtin_amdgcn_fence(int order, const char*) { __atomic_thread_fence(order); }

#define __expf(x) expf(x)   // v_exp_f32-based fast exp on the device
static inline double atomicAdd(double* addr, double val) {
    uint64_t* p = (uint64_t*)addr;
    uint64_t old = __atomic_load_n(p, __ATOMIC_RELAXED);
    for (;;) {
        double f;
        memcpy(&f, &old, 8);
        f += val;
        uint64_t nw;
        memcpy(&nw, &f, 8);
        if (__atomic_compare_exchange_n(p, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
            memcpy(&f, &old, 8);
            return f;
        }
    }
}

using std::max;
using std::min;

// ---- MFMA emulation (gfx950 semantics as documented in /opt/skills/guides/cdna_hip_programming.md section 3) ----
typedef float nf_emu_f32x16 __attribute__((ext_vector_type(16)));
typedef float nf_emu_f32x4 __attribute__((ext_vector_type(4)));

// D = A(32x2) * B(2x32) + C.  lane l: a = A[l&31][l>>5], b = B[l>>5][l&31];
// c[r] of lane l = C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].  k-ordered fmaf chain.
// all lanes publish (a, b) once; two wave barriers per MFMA
static inline void nf_emu_publish_ab(float a, float b, float* A, float* B) {
    uint64_t* sl = hip_emu::wave_slots();
    uint64_t raw = 0;
    float ab[2] = {a, b};
    memcpy(&raw, ab, 8);
    sl[hip_emu::lane()] = raw;
    hip_emu::wave_barrier();
    for (int l = 0; l < 64; ++l) {
        float t[2];
        memcpy(t, &sl[l], 8);
        A[l] = t[0];
        B[l] = t[1];
    }
    hip_emu::wave_barrier();
}

static inline nf_emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, nf_emu_f32x16 c, int, int, int) {
    float A[64], B[64];
    nf_emu_publish_ab(a, b, A, B);
    nf_emu_f32x16 d = c;
    int l = hip_emu::lane(), col = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        acc = fmaf(A[row], B[col], acc);             // k = 0: A[row][0] lives in lane row, B[0][col] in lane col
        acc = fmaf(A[row + 32], B[col + 32], acc);   // k = 1
        d[r] = acc;
    }
    return d;
}

// D = A(32x16) * B(16x32) + C, bf16 operands, fp32 accumulation.  lane l (r = l&31, h = l>>5) holds A[row r][k = 8h + j] and
// B[k = 8h + j][col r] in element j of its fragment; C/D as the 32x32x2 form (cdna_hip_programming.md section 3).
typedef __bf16 nf_emu_bf16x8 __attribute__((ext_vector_type(8)));
static inline nf_emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(nf_emu_bf16x8 a, nf_emu_bf16x8 b, nf_emu_f32x16 c, int, int, int) {
    float A[8][64], B[8][64];
    for (int j = 0; j < 8; ++j) nf_emu_publish_ab((float)a[j], (float)b[j], A[j], B[j]);
    nf_emu_f32x16 d = c;
    int l = hip_emu::lane(), col = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int h = 0; h < 2; ++h)
            for (int j = 0; j < 8; ++j) acc = fmaf(A[j][row + 32 * h], B[j][col + 32 * h], acc);
        d[r] = acc;
    }
    return d;
}

// D = A(16x4) * B(4x16) + C.  lane l: a = A[l&15][l>>4], b = B[l>>4][l&15]; c[r] = C[row = 4*(l>>4) + r][col = l&15]
static inline nf_emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, nf_emu_f32x4 c, int, int, int) {
    float A[64], B[64];
    nf_emu_publish_ab(a, b, A, B);
    nf_emu_f32x4 d = c;
    int l = hip_emu::lane(), col = l & 15, q = l >> 4;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * q + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(A[row + 16 * k], B[col + 16 * k], acc);
        d[r] = acc;
    }
    return d;
}

template <typename K, typename... Args>
static inline void hip_emu_launch(K kernel, dim3 grid, dim3 block, size_t smem, Args... args) {
    hip_emu::launch(grid, block, smem, [=]() { kernel(args...); });
}
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
    do { (void)(stream); hip_emu_launch(kernel, grid, block, smem, __VA_ARGS__); } while (0)
