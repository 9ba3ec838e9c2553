// Workspace-streaming dense helpers of the shape-generic kernels: outputs in registers (statically indexed), inputs
// streamed from a strided workspace column, weights read with wave-uniform addresses (scalar loads).
#pragma once
#include "nf_common.h"

// y[n] += sum_k Wt[k][n] * (x[k] * scale), x streamed from the workspace with element stride xstride
template <int N>
__device__ __forceinline__ void nf_dense_ws(const float* __restrict__ Wt, int K, const float* x, size_t xstride, float scale,
                                            float (&y)[N]) {
    for (int k = 0; k < K; ++k) {
        float xk = x[(size_t)k * xstride] * scale;
        const float* wr = Wt + k * N;
#pragma unroll
        for (int n = 0; n < N; ++n) y[n] = fmaf(wr[n], xk, y[n]);
    }
}

// same with an arbitrary leading dimension of Wt (a column block of a wider matrix)
template <int N>
__device__ __forceinline__ void nf_dense_ws_ld(const float* __restrict__ Wt, int ld, int K, const float* x, size_t xstride,
                                               float (&y)[N]) {
    for (int k = 0; k < K; ++k) {
        float xk = x[(size_t)k * xstride];
        const float* wr = Wt + k * ld;
#pragma unroll
        for (int n = 0; n < N; ++n) y[n] = fmaf(wr[n], xk, y[n]);
    }
}

// dx[k] += sum_n W[n][k] * dy[n], dy streamed from the workspace; W rows have leading dimension ldw
template <int K>
__device__ __forceinline__ void nf_dense_bwd_ws(const float* __restrict__ W, int N, int ldw, const float* dy, size_t dystride,
                                                float (&dx)[K]) {
    for (int n = 0; n < N; ++n) {
        float g = dy[(size_t)n * dystride];
        const float* wr = W + n * ldw;
#pragma unroll
        for (int k = 0; k < K; ++k) dx[k] = fmaf(wr[k], g, dx[k]);
    }
}

template <int N>
__device__ __forceinline__ void nf_load_bias(const float* __restrict__ b, float (&y)[N]) {
#pragma unroll
    for (int n = 0; n < N; ++n) y[n] = b[n];
}
