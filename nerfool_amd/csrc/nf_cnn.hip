// Fused "glue" of the ResUNet feature extractor (a14): InstanceNorm + affine + residual + ReLU/ELU + reflect padding in
// ONE pass over a convolution output, writing the tensor the NEXT convolution consumes (pre-padded, so the vendor
// convolution runs with padding 0); and the matching backward (fold of the padded gradient, activation derivative,
// InstanceNorm backward).  The convolutions themselves stay on MIOpen in this round.
//
// ref: ibrnet/feature_network.py:38-78 (BasicBlock: conv3x3 reflect -> IN -> ReLU -> conv3x3 reflect -> IN -> +id -> ReLU),
//      :127-140 (conv + IN + ELU), :28-36 (padding_mode='reflect').  PyTorch runs this as 4-5 memory-bound kernels per
//      convolution (reflection_pad2d, batch_norm x2 passes, add, relu) in each direction.
//
// One workgroup (1024 threads) per (image, channel) plane: plane statistics are block reductions, the data is re-read
// from L2 / Infinity Cache for the later passes.
#include "nf_common.h"

#ifndef NF_CNN_THREADS
#define NF_CNN_THREADS 1024   // the CPU stand-in build of the tests overrides this with 64 (one OS thread per GPU thread)
#endif

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = nf_wave_sum(v);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                 // protect `red` from the previous use
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) t += red[w];
    return t;
}

__device__ __forceinline__ int reflect_src(int i, int n) {      // index into [0,n) of position i of the reflect-padded axis
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : __expf(v) - 1.f;
    return v;
}

__device__ __forceinline__ float act_grad_from_out(float y, int act) {
    if (act == 1) return y > 0.f ? 1.f : 0.f;
    if (act == 2) return y > 0.f ? 1.f : y + 1.f;
    return 1.f;
}

// y_p[p, i, j] = act( gamma * (x[p, si, sj] - mean_p) * rstd_p + beta + res[si, sj] ),  (si, sj) = reflect(i - pad, j - pad)
// gamma == nullptr: no normalisation (pure padding / activation).
__global__ void __launch_bounds__(NF_CNN_THREADS) k_in_act_pad_fwd(const float* __restrict__ x, int C, int H, int W,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float eps,
                                                                   const float* __restrict__ res, int64_t rs_n, int64_t rs_c,
                                                                   int64_t rs_h, int64_t rs_w, int act, int pad,
                                                                   float* __restrict__ yp, float* __restrict__ mean_out,
                                                                   float* __restrict__ rstd_out) {
    __shared__ float red[NF_CNN_THREADS / 64];
    const int64_t p = blockIdx.x;
    const int n = (int)(p / C), c = (int)(p - (int64_t)n * C);
    const int HW = H * W;
    const float* xp = x + p * HW;
    float mean = 0.f, rstd = 1.f, g = 1.f, b = 0.f;
    if (gamma) {
        float s = 0.f;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) s += xp[i];
        mean = block_sum(s, red) / (float)HW;
        float q = 0.f;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            float d = xp[i] - mean;
            q += d * d;
        }
        float var = block_sum(q, red) / (float)HW;
        rstd = 1.f / sqrtf(var + eps);
        g = gamma[c];
        b = beta[c];
        if (threadIdx.x == 0) {
            mean_out[p] = mean;
            rstd_out[p] = rstd;
        }
    }
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    float* out = yp + p * (int64_t)Hp * Wp;
    const float* rp = res ? res + n * rs_n + c * rs_c : nullptr;
    for (int i = threadIdx.x; i < Hp * Wp; i += blockDim.x) {
        int ph = i / Wp, pw = i - ph * Wp;
        int sh = reflect_src(ph - pad, H), sw = reflect_src(pw - pad, W);
        float v = (xp[sh * W + sw] - mean) * rstd * g + b;
        if (rp) v += rp[sh * rs_h + sw * rs_w];
        out[i] = act_fwd(v, act);
    }
}

// Backward of the above for one plane.
//   d[h,w]   = sum of dyp over the padded positions that mirror onto (h,w)   (+ d_extra[h,w])
//   d_pre    = d * act'(y)                                  -> d_res (gradient of the residual input), if requested
//   dx       = gamma rstd (d_pre - mean(d_pre) - xhat mean(d_pre xhat))       (InstanceNorm backward; = d_pre without norm)
__global__ void __launch_bounds__(NF_CNN_THREADS) k_in_act_pad_bwd(const float* __restrict__ dyp, const float* __restrict__ d_extra,
                                                                   const float* __restrict__ yp, const float* __restrict__ x,
                                                                   int C, int H, int W, const float* __restrict__ gamma,
                                                                   const float* __restrict__ mean_in,
                                                                   const float* __restrict__ rstd_in, int act, int pad,
                                                                   float* __restrict__ d_res, float* __restrict__ dx) {
    __shared__ float red[NF_CNN_THREADS / 64];
    const int64_t p = blockIdx.x;
    const int c = (int)(p % C);
    const int HW = H * W, Hp = H + 2 * pad, Wp = W + 2 * pad;
    const float* gp = dyp ? dyp + p * (int64_t)Hp * Wp : nullptr;
    const float* ep = d_extra ? d_extra + p * HW : nullptr;
    const float* yq = yp + p * (int64_t)Hp * Wp;
    const float* xq = x ? x + p * HW : nullptr;
    float* dxp = dx + p * HW;
    float* drp = d_res ? d_res + p * HW : nullptr;
    const float mean = gamma ? mean_in[p] : 0.f, rstd = gamma ? rstd_in[p] : 1.f;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        int h = i / W, w = i - h * W;
        float d = ep ? ep[i] : 0.f;
        if (gp) {
            // rows / columns of the padded gradient that mirror onto (h, w)
            int rows[3], cols[3], nr = 0, nc = 0;
            rows[nr++] = h + pad;
            if (h >= 1 && h <= pad) rows[nr++] = pad - h;
            if (h <= H - 2 && h >= H - 1 - pad) rows[nr++] = 2 * (H - 1) + pad - h;
            cols[nc++] = w + pad;
            if (w >= 1 && w <= pad) cols[nc++] = pad - w;
            if (w <= W - 2 && w >= W - 1 - pad) cols[nc++] = 2 * (W - 1) + pad - w;
            for (int a = 0; a < nr; ++a)
                for (int b = 0; b < nc; ++b) d += gp[rows[a] * Wp + cols[b]];
        }
        float y = yq[(h + pad) * Wp + (w + pad)];
        float dpre = d * act_grad_from_out(y, act);
        if (drp) drp[i] = dpre;
        dxp[i] = dpre;
        if (gamma) {
            float xh = (xq[i] - mean) * rstd;
            s1 += dpre;
            s2 += dpre * xh;
        }
    }
    if (!gamma) return;
    float m1 = block_sum(s1, red) / (float)HW;
    float m2 = block_sum(s2, red) / (float)HW;
    const float gr = gamma[c] * rstd;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {        // each thread re-reads the elements it wrote
        float xh = (xq[i] - mean) * rstd;
        dxp[i] = gr * (dxp[i] - m1 - xh * m2);
    }
}

extern "C" int nf_in_act_pad_fwd(const float* x, int n_img, int C, int H, int W, const float* gamma, const float* beta, float eps,
                                 const float* res, int64_t rs_n, int64_t rs_c, int64_t rs_h, int64_t rs_w, int act, int pad,
                                 float* y_padded, float* mean, float* rstd, nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && C >= 1 && H >= 1 && W >= 1 && pad >= 0 && pad < H && pad < W && act >= 0 && act <= 2,
               "nf_in_act_pad_fwd: bad arguments (N %d C %d H %d W %d pad %d act %d)", n_img, C, H, W, pad, act);
    hipLaunchKernelGGL(k_in_act_pad_fwd, dim3((unsigned)(n_img * C)), dim3(NF_CNN_THREADS), 0, (hipStream_t)stream, x, C, H, W,
                       gamma, beta, eps, res, rs_n, rs_c, rs_h, rs_w, act, pad, y_padded, mean, rstd);
    NF_LAUNCH_CHECK("nf_in_act_pad_fwd");
    return 0;
}

extern "C" int nf_in_act_pad_bwd(const float* dy_padded, const float* d_extra, const float* y_padded, const float* x, int n_img,
                                 int C, int H, int W, const float* gamma, const float* mean, const float* rstd, int act, int pad,
                                 float* d_res, float* dx, nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && C >= 1 && H >= 1 && W >= 1 && pad >= 0 && pad < H && pad < W && act >= 0 && act <= 2 &&
                   (dy_padded || d_extra),
               "nf_in_act_pad_bwd: bad arguments");
    hipLaunchKernelGGL(k_in_act_pad_bwd, dim3((unsigned)(n_img * C)), dim3(NF_CNN_THREADS), 0, (hipStream_t)stream, dy_padded,
                       d_extra, y_padded, x, C, H, W, gamma, mean, rstd, act, pad, d_res, dx);
    NF_LAUNCH_CHECK("nf_in_act_pad_bwd");
    return 0;
}
