// Fused perturbation update of the PGD loop (a11/a13): one streaming pass over delta instead of ~12 ATen launches.
// ref: eval/ibrnet/eval_adv.py:28-29 (clamp), :248-254 (init), :805-819 (Adam ascent), :822-828 (sign-PGD),
//      :838-839 (eps-ball then [0,1]-box projection).  32 B/element (read delta, g, m, v, src; write delta, m, v).
#include "nf_common.h"

__device__ __forceinline__ float nf_project_delta(float d, float src, float eps, float lo, float hi) {
    if (eps >= 0.f) d = fmaxf(fminf(d, eps), -eps);          // clamp(delta, -eps, eps)
    return fmaxf(fminf(d, hi - src), lo - src);              // clamp(delta, lo - src, hi - src)
}

__global__ void __launch_bounds__(256) k_project_perturb(float* __restrict__ delta, const float* __restrict__ src, int64_t n,
                                                         float eps, float lo, float hi) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        delta[i] = nf_project_delta(delta[i], src[i], eps, lo, hi);
}

// torch.optim.Adam single-tensor step on g = -grad (ascent through a minimiser, eval_adv.py:812):
//   m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2); denom = sqrt(v)/bc2_sqrt + eps; p.addcdiv_(m, denom, -step)
__global__ void __launch_bounds__(256) k_pgd_adam_step(float* __restrict__ delta, const float* __restrict__ grad,
                                                       float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                       const float* __restrict__ src, int64_t n, float neg_step_size,
                                                       float w1, float beta2, float w2, float bc2_sqrt, float adam_eps,
                                                       float eps, float lo, float hi) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float g = -grad[i];
        float m = exp_avg[i];
        m = m + w1 * (g - m);
        float v = exp_avg_sq[i] * beta2 + w2 * (g * g);
        float denom = sqrtf(v) / bc2_sqrt + adam_eps;
        float d = delta[i] + neg_step_size * (m / denom);
        exp_avg[i] = m;
        exp_avg_sq[i] = v;
        delta[i] = nf_project_delta(d, src[i], eps, lo, hi);
    }
}

// The same update with the two per-iteration scalars read from DEVICE memory: hyper = {neg_step_size, bc2_sqrt} (computed on the
// host in double and rounded to float exactly as for nf_pgd_adam_step, then copied stream-ordered).  A PGD step captured into a
// hipGraph replays the identical launch every iteration; only these two numbers (Adam's bias corrections and the StepLR rate)
// change with the iteration count.
__global__ void __launch_bounds__(256) k_pgd_adam_step_dev(float* __restrict__ delta, const float* __restrict__ grad,
                                                           float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                           const float* __restrict__ src, int64_t n, const float* __restrict__ hyper,
                                                           float w1, float beta2, float w2, float adam_eps, float eps, float lo, float hi) {
    const float neg_step_size = hyper[0], bc2_sqrt = hyper[1];
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float g = -grad[i];
        float m = exp_avg[i];
        m = m + w1 * (g - m);
        float v = exp_avg_sq[i] * beta2 + w2 * (g * g);
        float denom = sqrtf(v) / bc2_sqrt + adam_eps;
        float d = delta[i] + neg_step_size * (m / denom);
        exp_avg[i] = m;
        exp_avg_sq[i] = v;
        delta[i] = nf_project_delta(d, src[i], eps, lo, hi);
    }
}

__global__ void __launch_bounds__(256) k_pgd_sign_step(float* __restrict__ delta, const float* __restrict__ grad,
                                                       const float* __restrict__ src, int64_t n, float alpha, float eps,
                                                       float lo, float hi) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float g = grad[i];
        float sgn = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
        delta[i] = nf_project_delta(delta[i] + alpha * sgn, src[i], eps, lo, hi);
    }
}

static unsigned nf_stream_grid(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

extern "C" int nf_project_perturb(float* delta, const float* src, int64_t n, float epsilon, float lower, float upper,
                                  nf_stream_t stream) {
    NF_REQUIRE(n >= 0, "nf_project_perturb: bad size");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_project_perturb, dim3(nf_stream_grid(n)), dim3(256), 0, (hipStream_t)stream, delta, src, n, epsilon,
                       lower, upper);
    NF_LAUNCH_CHECK("nf_project_perturb");
    return 0;
}

extern "C" int nf_pgd_adam_step(float* delta, const float* grad, float* exp_avg, float* exp_avg_sq, const float* src,
                                int64_t n, float neg_step_size, float one_minus_beta1, float beta2, float one_minus_beta2,
                                float bc2_sqrt, float adam_eps, float epsilon, float lower, float upper,
                                nf_stream_t stream) {
    NF_REQUIRE(n >= 0 && bc2_sqrt > 0.f, "nf_pgd_adam_step: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_pgd_adam_step, dim3(nf_stream_grid(n)), dim3(256), 0, (hipStream_t)stream, delta, grad, exp_avg,
                       exp_avg_sq, src, n, neg_step_size, one_minus_beta1, beta2, one_minus_beta2, bc2_sqrt, adam_eps, epsilon, lower,
                       upper);
    NF_LAUNCH_CHECK("nf_pgd_adam_step");
    return 0;
}

extern "C" int nf_pgd_adam_step_dev(float* delta, const float* grad, float* exp_avg, float* exp_avg_sq, const float* src,
                                    int64_t n, const float* hyper_dev, float one_minus_beta1, float beta2, float one_minus_beta2,
                                    float adam_eps, float epsilon, float lower, float upper, nf_stream_t stream) {
    NF_REQUIRE(n >= 0 && hyper_dev != nullptr, "nf_pgd_adam_step_dev: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_pgd_adam_step_dev, dim3(nf_stream_grid(n)), dim3(256), 0, (hipStream_t)stream, delta, grad, exp_avg,
                       exp_avg_sq, src, n, hyper_dev, one_minus_beta1, beta2, one_minus_beta2, adam_eps, epsilon, lower, upper);
    NF_LAUNCH_CHECK("nf_pgd_adam_step_dev");
    return 0;
}

extern "C" int nf_pgd_sign_step(float* delta, const float* grad, const float* src, int64_t n, float alpha, float epsilon,
                                float lower, float upper, nf_stream_t stream) {
    NF_REQUIRE(n >= 0, "nf_pgd_sign_step: bad size");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_pgd_sign_step, dim3(nf_stream_grid(n)), dim3(256), 0, (hipStream_t)stream, delta, grad, src, n,
                       alpha, epsilon, lower, upper);
    NF_LAUNCH_CHECK("nf_pgd_sign_step");
    return 0;
}
