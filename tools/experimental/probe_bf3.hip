// EXPERIMENT (not product): would a 3-way bf16 split of the Winograd products pay?  Per 16-channel chunk and wave the shipped
// fp32 loop issues 64 v_mfma_f32_32x32x2_f32 (+ ~160 vector instructions: window transform, address arithmetic); a split loop
// would issue 48 v_mfma_f32_32x32x16_bf16 (6 cross products x 4 nu x 2 kb) + ~340 vector instructions (the same transform + the
// hi / mid / lo split of 32 B-operand values).  Times both mixes (vector instructions as non-packed v_fma_f32) at 2 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/experimental/probe_bf3.hip -o tools/experimental/probe_bf3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int NM32, int NMBF, int NV>
__global__ void __launch_bounds__(256, 2) k(float* out, int iters, float seed) {
    f16v acc[8];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = seed + threadIdx.x, b = 1.0001f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * i;
    bf8 ba, bb;
    for (int i = 0; i < 8; ++i) { ba[i] = (__bf16)(seed + i); bb[i] = (__bf16)(seed - i); }
    const float c1 = 1.0001f, c2 = 0.5f;
    for (int it = 0; it < iters; ++it) {
        // one "chunk": the matrix instructions spread evenly between the vector instructions
        constexpr int NMT = (NM32 + NMBF) > 0 ? (NM32 + NMBF) : 1;
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
            if (NM32 + NMBF == 0) {} else if (NM32) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 7], 0, 0, 0);
            else acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, acc[m & 7], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV / NMT; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(c1), "v"(c2));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM32, int NMBF, int NV>
float run(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 2;
    hipLaunchKernelGGL((k<NM32, NMBF, NV>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM32, NMBF, NV>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
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
    printf("per chunk, 2 waves / SIMD, %d chunks:\n", iters);
    printf("  64 fp32 MFMA alone                 %.3f ms\n", run<64, 0, 0>(out, iters));
    printf("  64 fp32 MFMA + 128 v_fma           %.3f ms   (the shipped inner loop's mix)\n", run<64, 0, 128>(out, iters));
    printf("  64 fp32 MFMA + 192 v_fma           %.3f ms\n", run<64, 0, 192>(out, iters));
    printf("  48 bf16 MFMA alone                 %.3f ms\n", run<0, 48, 0>(out, iters));
    printf("  48 bf16 MFMA + 192 v_fma           %.3f ms\n", run<0, 48, 192>(out, iters));
    printf("  48 bf16 MFMA + 336 v_fma           %.3f ms   (3-way split: 6 cross products)\n", run<0, 48, 336>(out, iters));
    printf("  48 bf16 MFMA + 480 v_fma           %.3f ms\n", run<0, 48, 480>(out, iters));
    printf("  24 bf16 MFMA + 240 v_fma           %.3f ms   (2-way split: 3 cross products)\n", run<0, 24, 240>(out, iters));
    printf("   8 bf16 MFMA + 192 v_fma           %.3f ms   (plain bf16 operands)\n", run<0, 8, 192>(out, iters));
    printf("   0 MFMA + 336 v_fma                %.3f ms\n", run<0, 0, 336>(out, iters));
    return 0;
}
