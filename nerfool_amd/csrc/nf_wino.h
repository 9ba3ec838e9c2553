// Shared pieces of the Winograd convolution kernels (nf_wino.hip: fp32 matrix-core operands; nf_wino_bf.hip: bf16 operand splits):
// tile geometry, tensor strides, and the three device-only primitives of the per-wave LDS weight ring.
#ifndef NF_WINO_H
#define NF_WINO_H
#include "nf_common.h"

#include <type_traits>

typedef float w16 __attribute__((ext_vector_type(16)));
typedef float w2f __attribute__((ext_vector_type(2), aligned(4)));
typedef float w4f __attribute__((ext_vector_type(4)));
typedef float w2a __attribute__((ext_vector_type(2)));        // 8-byte aligned: LDS accesses (ds_read_b64 / ds_write_b64)
#define WN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__host__ __device__ constexpr int wn_nidx(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

#define WN_CC 16            // input channels per chunk
#define WN_PC 18            // patch columns used (8 tiles: 16 + 2)
#define WN_PS 24            // patch row stride in LDS (the four tile rows of a wave then hit disjoint bank groups)

struct WnTensor { int64_t ns, cs, rs; };        // element strides: image, channel, row (unit column stride)

// Ring slots per wave.  A step's records are issued slots - 1 steps before they are used, into the slot whose operands were
// read (one step ahead of their use) two steps earlier: the wait of the step in between also retires the wave's LDS reads
// (lgkmcnt(0)), so a refill can never overtake a read of the slot it overwrites.  (Refilling a slot right behind the step that
// read it corrupted a few outputs per launch on the large layers: an L2-hit DMA can land within ~250 cycles, before a read
// still queued behind the neighbour workgroup's LDS traffic has executed.)
// KB = 2 (64 output channels per workgroup): 4 slots -- the slot of a step is then a compile-time constant (4 divides the 8 steps
// of a chunk): no slot bookkeeping, immediate LDS offsets; 6 slots with run-time indices measured 2 % slower -- 62 KB of staging,
// 64 KB with the output exchange, two workgroups per CU.  KB = 1 (32 channels: half the multiplications per staged window):
// 4 slots, 46 KB, three workgroups per CU.
#ifndef WN_SLOTS2
#define WN_SLOTS2 4
#endif
__host__ __device__ constexpr int wn_slots(int kb) { return kb == 1 ? 4 : WN_SLOTS2; }
#define WN_FETCH_OPS 6      // VM instructions one window fetch issues per wave (two segments x three 8-byte loads)

// ---- the three device-only primitives of the weight ring (their host forms keep the file compilable in the host pass)
// wave-uniform value in a scalar register
__device__ __forceinline__ int wn_uniform(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
}
// LDS-DMA of one piece: lane l moves 16 bytes from gsrc + 4 l (floats) to dst + 4 l; no VGPR destination, counted on vmcnt
__device__ __forceinline__ void wn_dma16(const float* gsrc, float* dst, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned keep;
    const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane * 16), "s"(lds), "s"(gsrc)
                 : "memory");
#else
    for (int j = 0; j < 4; ++j) dst[4 * lane + j] = gsrc[4 * lane + j];
#endif
}
// the same for the PIECES (1 or 2) consecutive 1 KB pieces of a step under one M0 set-up: the instruction offset of the second
// piece advances the global and the LDS address alike
template <int PIECES>
__device__ __forceinline__ void wn_dma16xn(const float* gsrc, float* dst, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned keep;
    const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)dst;
    if (PIECES == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                     "global_load_lds_dwordx4 %1, %3 offset:1024\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(lane * 16), "s"(lds), "s"(gsrc)
                     : "memory");
    else
        wn_dma16(gsrc, dst, lane);
#else
    for (int j = 0; j < 4 * PIECES; ++j) dst[(j >> 2) * 256 + 4 * lane + (j & 3)] = gsrc[(j >> 2) * 256 + 4 * lane + (j & 3)];
#endif
}
// wait until at most N of this wave's VM operations are outstanding (they retire in issue order) and all its LDS reads returned
template <int N>
__device__ __forceinline__ void wn_wait_vm() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
#endif
}

#endif
