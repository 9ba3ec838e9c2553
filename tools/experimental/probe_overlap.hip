// EXPERIMENT (not product): do v_mfma_f32_32x32x2_f32 and fp32 VALU instructions of DIFFERENT waves on one SIMD overlap on gfx950,
// or do they share an execution resource?  Times, per wave count per SIMD: M = a chain of independent-accumulator MFMAs, V = a chain
// of v_fma_f32 / v_exp_f32, MV = both interleaved in every wave.  overlap => t(MV) ~ max(t(M), t(V)); shared => ~ t(M) + t(V).
// build: hipcc -O3 --offload-arch=gfx950 tools/experimental/probe_overlap.hip -o tools/experimental/probe_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NM, int NV, int NE, bool BF>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    f16v acc0 = {0}, acc1 = {0};
    float a = seed + threadIdx.x, b = 1.0001f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * i;
    typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
    bf8 ba, bb;
    for (int i = 0; i < 8; ++i) { ba[i] = (__bf16)(seed + i); bb[i] = (__bf16)(seed - i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (NM) {
                if (BF) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, acc0, 0, 0, 0);
                    if (NM > 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                    if (NM > 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
#pragma unroll
            for (int q = 0; q < NE; ++q) v[q & 7] = __builtin_amdgcn_exp2f(v[q & 7]);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, int NE, bool BF>
float run(float* out, int blocks_per_cu, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL((k<NM, NV, NE, BF>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV, NE, BF>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 2) {          // workgroups of 4 waves per CU = waves per SIMD
        // per loop body (x8 per iteration): 2 MFMA (fp32: 128 pipe cycles) against 24 fma (96 issue cycles) / 6 exp (96 cycles)
        float m = run<2, 0, 0, false>(out, wps, iters), v = run<0, 24, 0, false>(out, wps, iters), mv = run<2, 24, 0, false>(out, wps, iters);
        float e = run<0, 0, 6, false>(out, wps, iters), me = run<2, 0, 6, false>(out, wps, iters);
        float mb = run<2, 0, 0, true>(out, wps, iters), mbv = run<2, 24, 0, true>(out, wps, iters);
        printf("waves/SIMD %d: fp32 MFMA %.3f ms | 24 fma %.3f | both %.3f (sum %.3f, max %.3f) || 6 exp %.3f | MFMA+exp %.3f || bf16 MFMA %.3f | +24 fma %.3f\n",
               wps, m, v, mv, m + v, m > v ? m : v, e, me, mb, mbv);
    }
    return 0;
}
