// Fused "glue" of the ResUNet feature extractor (a14): InstanceNorm + affine + residual + ReLU/ELU + reflect padding in
// ONE pass over a convolution output, writing the tensor the NEXT convolution consumes (pre-padded, so the vendor
// convolution runs with padding 0); and the matching backward (fold of the padded gradient, activation derivative,
// InstanceNorm backward).  The convolutions themselves stay on MIOpen in this round.
//
// ref: ibrnet/feature_network.py:38-78 (BasicBlock: conv3x3 reflect -> IN -> ReLU -> conv3x3 reflect -> IN -> +id -> ReLU),
//      :127-140 (conv + IN + ELU), :28-36 (padding_mode='reflect').  PyTorch runs this as 4-5 memory-bound kernels per
//      convolution (reflection_pad2d, batch_norm x2 passes, add, relu) in each direction.
//
// Two fully parallel launches per direction: plane statistics (block partial sums in double, one fp64 atomic pair per
// block) and an element-wise apply; the second read of the convolution output comes from L2 / Infinity Cache.
#include "nf_common.h"

#define NF_CNN_BLOCK 256

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, NF_WAVE);
    return v;
}

#define NF_MAX_SPLITS 32   // partial sums per plane: scratch holds [planes][NF_MAX_SPLITS][2] doubles, no atomics, no memset

// block partial sums (double) -> this block's slot dst[0..1]
__device__ __forceinline__ void block_store_sums(double a, double b, double* dst) {
    __shared__ double red[2][NF_CNN_BLOCK / 64];
    a = wave_sum_f64(a);
    b = wave_sum_f64(b);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { ta += red[0][w]; tb += red[1][w]; }
        dst[0] = ta;
        dst[1] = tb;
    }
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

// ---- forward, pass 1: per-plane sum and sum of squares (double accumulation).  grid (splits, planes)
__global__ void __launch_bounds__(NF_CNN_BLOCK) k_in_stats(const float* __restrict__ x, int HW, double* __restrict__ sums) {
    const int64_t p = blockIdx.y;
    const float* xp = x + p * HW;
    int seg = (HW + gridDim.x - 1) / gridDim.x;
    seg = (seg + 3) & ~3;
    int lo = blockIdx.x * seg, hi = min(lo + seg, HW);
    double s = 0.0, q = 0.0;
    if ((HW & 3) == 0) {       // planes are 16-byte aligned when HW % 4 == 0: float4 stream
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        for (int i = lo / 4 + threadIdx.x; i < hi / 4; i += blockDim.x) {
            float4 v = x4[i];
            s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            float v = xp[i];
            s += (double)v;
            q += (double)v * v;
        }
    }
    block_store_sums(s, q, sums + (p * NF_MAX_SPLITS + blockIdx.x) * 2);
}

__device__ __forceinline__ void plane_sums(const double* __restrict__ sums, int64_t p, int n_splits, double& a, double& b) {
    a = 0.0;
    b = 0.0;
    for (int i = 0; i < n_splits; ++i) {
        a += sums[(p * NF_MAX_SPLITS + i) * 2];
        b += sums[(p * NF_MAX_SPLITS + i) * 2 + 1];
    }
}

// ---- forward, pass 2: y_p[p, i, j] = act( gamma * (x[p, si, sj] - mean_p) * rstd_p + beta + res[si, sj] ),
//      (si, sj) = reflect(i - pad, j - pad).  gamma == nullptr: no normalisation.  grid (chunks of the padded plane, planes)
__global__ void __launch_bounds__(NF_CNN_BLOCK) k_in_act_pad_fwd(const float* __restrict__ x, int C, int H, int W,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float eps, const double* __restrict__ sums, int n_splits,
                                                                 const float* __restrict__ res, int64_t rs_n, int64_t rs_c,
                                                                 int64_t rs_h, int64_t rs_w, int act, int pad,
                                                                 float* __restrict__ yp, float* __restrict__ mean_out,
                                                                 float* __restrict__ rstd_out) {
    const int64_t p = blockIdx.y;
    const int n = (int)(p / C), c = (int)(p - (int64_t)n * C);
    const int HW = H * W;
    float mean = 0.f, rstd = 1.f, g = 1.f, b = 0.f;
    if (gamma) {
        double sa, sb;
        plane_sums(sums, p, n_splits, sa, sb);
        double m = sa / (double)HW;
        double var = sb / (double)HW - m * m;
        mean = (float)m;
        rstd = 1.f / sqrtf(fmaxf((float)var, 0.f) + eps);
        g = gamma[c];
        b = beta[c];
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            mean_out[p] = mean;
            rstd_out[p] = rstd;
        }
    }
    const int Hp = H + 2 * pad, Wp = W + 2 * pad, HWp = Hp * Wp;
    const float* xp = x + p * HW;
    const float* rp = res ? res + n * rs_n + c * rs_c : nullptr;
    float* out = yp + p * (int64_t)HWp;
    int seg = (HWp + gridDim.x - 1) / gridDim.x;
    int lo = blockIdx.x * seg, hi = min(lo + seg, HWp);
#pragma unroll 4
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        int ph = i / Wp, pw = i - ph * Wp;
        int sh = reflect_src(ph - pad, H), sw = reflect_src(pw - pad, W);
        float v = (xp[sh * W + sw] - mean) * rstd * g + b;
        if (rp) v += rp[sh * rs_h + sw * rs_w];
        out[i] = act_fwd(v, act);
    }
}

// ---- backward, pass 1: fold the padded gradient, activation derivative -> d_pre (written to dx and d_res), and the two
//      plane sums of the InstanceNorm backward.  grid (chunks of the plane, planes)
__global__ void __launch_bounds__(NF_CNN_BLOCK) k_in_act_pad_bwd1(const float* __restrict__ dyp, const float* __restrict__ d_extra,
                                                                  const float* __restrict__ yp, const float* __restrict__ x,
                                                                  int H, int W, const float* __restrict__ mean_in,
                                                                  const float* __restrict__ rstd_in, int has_norm, int act, int pad,
                                                                  float* __restrict__ d_res, float* __restrict__ dx,
                                                                  double* __restrict__ sums) {
    const int64_t p = blockIdx.y;
    const int HW = H * W, Hp = H + 2 * pad, Wp = W + 2 * pad;
    const float* gp = dyp ? dyp + p * (int64_t)Hp * Wp : nullptr;
    const float* ep = d_extra ? d_extra + p * HW : nullptr;
    const float* yq = yp + p * (int64_t)Hp * Wp;
    const float* xq = has_norm ? x + p * HW : nullptr;
    float* dxp = dx + p * HW;
    float* drp = d_res ? d_res + p * HW : nullptr;
    const float mean = has_norm ? mean_in[p] : 0.f, rstd = has_norm ? rstd_in[p] : 1.f;
    int seg = (HW + gridDim.x - 1) / gridDim.x;
    int lo = blockIdx.x * seg, hi = min(lo + seg, HW);
    double s1 = 0.0, s2 = 0.0;
#pragma unroll 2
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        int h = i / W, w = i - h * W;
        float d = ep ? ep[i] : 0.f;
        if (gp) {
            d += gp[(h + pad) * Wp + (w + pad)];
            // border pixels also collect the padded positions that mirror onto them
            bool hb = (h >= 1 && h <= pad) || (h <= H - 2 && h >= H - 1 - pad);
            bool wb = (w >= 1 && w <= pad) || (w <= W - 2 && w >= W - 1 - pad);
            if (hb || wb) {
                int rows[3], cols[3], nr = 0, nc = 0;
                rows[nr++] = h + pad;
                if (h >= 1 && h <= pad) rows[nr++] = pad - h;
                if (h <= H - 2 && h >= H - 1 - pad) rows[nr++] = 2 * (H - 1) + pad - h;
                cols[nc++] = w + pad;
                if (w >= 1 && w <= pad) cols[nc++] = pad - w;
                if (w <= W - 2 && w >= W - 1 - pad) cols[nc++] = 2 * (W - 1) + pad - w;
                for (int a = 0; a < nr; ++a)
                    for (int b = 0; b < nc; ++b)
                        if (a + b > 0) d += gp[rows[a] * Wp + cols[b]];
            }
        }
        float y = yq[(h + pad) * Wp + (w + pad)];
        float dpre = d * act_grad_from_out(y, act);
        // with a residual output AND a normalisation pass behind, d_pre is stored once (in d_res) and pass 2 reads it there
        if (drp) drp[i] = dpre;
        if (!drp || !has_norm) dxp[i] = dpre;
        if (has_norm) {
            float xh = (xq[i] - mean) * rstd;
            s1 += (double)dpre;
            s2 += (double)dpre * xh;
        }
    }
    if (has_norm) block_store_sums(s1, s2, sums + (p * NF_MAX_SPLITS + blockIdx.x) * 2);
}

// ---- backward, pass 2: dx = gamma rstd (d_pre - mean(d_pre) - xhat mean(d_pre xhat)), in place on dx
__global__ void __launch_bounds__(NF_CNN_BLOCK) k_in_act_pad_bwd2(const float* __restrict__ x, int C, int HW,
                                                                  const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                                  const float* __restrict__ rstd_in, const double* __restrict__ sums,
                                                                  int n_splits, const float* d_pre, float* dx) {
    const int64_t p = blockIdx.y;
    const int c = (int)(p % C);
    const float mean = mean_in[p], rstd = rstd_in[p];
    double sa, sb;
    plane_sums(sums, p, n_splits, sa, sb);
    const float m1 = (float)(sa / (double)HW), m2 = (float)(sb / (double)HW);
    const float gr = gamma[c] * rstd;
    const float* xq = x + p * HW;
    const float* dpp = d_pre + p * HW;       // == dx (in place) unless pass 1 left d_pre in the residual gradient
    float* dxp = dx + p * HW;
    int seg = (HW + gridDim.x - 1) / gridDim.x;
    int lo = blockIdx.x * seg, hi = min(lo + seg, HW);
#pragma unroll 4
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        float xh = (xq[i] - mean) * rstd;
        dxp[i] = gr * (dpp[i] - m1 - xh * m2);
    }
}

// element-wise passes: ~8192 workgroups of 256 threads over the chip, each streaming >= 1024 elements of one plane
static unsigned nf_apply_splits(int planes, int n) {
    int splits = 8192 / planes;
    int cap = n / 1024;
    if (splits > cap) splits = cap;
    if (splits > NF_MAX_SPLITS) splits = NF_MAX_SPLITS;
    return (unsigned)(splits < 1 ? 1 : splits);
}

static unsigned nf_stat_splits(int planes, int HW) {
    int splits = 2048 / planes;
    int cap = HW / 4096;
    if (splits > cap) splits = cap;
    if (splits > NF_MAX_SPLITS) splits = NF_MAX_SPLITS;
    return (unsigned)(splits < 1 ? 1 : splits);
}

// ---- decoder: bilinear x2 upsampling (align_corners = True) written directly as the reflect-padded input of the 3x3
//      convolution that follows (feature_network.py:143-151: F.interpolate -> conv with padding_mode='reflect').
//      Arithmetic in the order of ATen's upsample_bilinear2d so that the values equal F.interpolate's.
__global__ void __launch_bounds__(NF_CNN_BLOCK) k_upsample2x_pad(const float* __restrict__ x, int64_t xs_plane, int64_t xs_row,
                                                                 int h, int w, int pad, float* __restrict__ yp) {
    const int64_t p = blockIdx.y;
    const int H = 2 * h, W = 2 * w, Hp = H + 2 * pad, Wp = W + 2 * pad, HWp = Hp * Wp;
    const float rh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float rw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float* xp = x + p * xs_plane;
    float* out = yp + p * (int64_t)HWp;
    int seg = (HWp + gridDim.x - 1) / gridDim.x;
    int lo = blockIdx.x * seg, hi = min(lo + seg, HWp);
#pragma unroll 2
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        int ph = i / Wp, pw = i - ph * Wp;
        int oh = reflect_src(ph - pad, H), ow = reflect_src(pw - pad, W);
        float h1r = rh * (float)oh, w1r = rw * (float)ow;
        int h1 = (int)h1r, w1 = (int)w1r;
        int h1p = h1 < h - 1 ? 1 : 0, w1p = w1 < w - 1 ? 1 : 0;
        float h1l = h1r - (float)h1, h0l = 1.f - h1l, w1l = w1r - (float)w1, w0l = 1.f - w1l;
        const float* r0 = xp + (int64_t)h1 * xs_row + w1;
        const float* r1 = r0 + (int64_t)h1p * xs_row;
        out[i] = h0l * (w0l * r0[0] + w1l * r0[w1p]) + h1l * (w0l * r1[0] + w1l * r1[w1p]);
    }
}

extern "C" int nf_in_act_pad_fwd(const float* x, int n_img, int C, int H, int W, const float* gamma, const float* beta, float eps,
                                 const float* res, int64_t rs_n, int64_t rs_c, int64_t rs_h, int64_t rs_w, int act, int pad,
                                 float* y_padded, float* mean, float* rstd, void* scratch, nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && C >= 1 && H >= 1 && W >= 1 && pad >= 0 && pad < H && pad < W && act >= 0 && act <= 2,
               "nf_in_act_pad_fwd: bad arguments (N %d C %d H %d W %d pad %d act %d)", n_img, C, H, W, pad, act);
    hipStream_t st = (hipStream_t)stream;
    const int planes = n_img * C, HW = H * W, HWp = (H + 2 * pad) * (W + 2 * pad);
    unsigned stat_splits = 1;
    if (gamma) {
        NF_REQUIRE(scratch != nullptr, "nf_in_act_pad_fwd: scratch (512 bytes per plane) required with normalisation");
        stat_splits = nf_stat_splits(planes, HW);
        hipLaunchKernelGGL(k_in_stats, dim3(stat_splits, (unsigned)planes), dim3(NF_CNN_BLOCK), 0, st, x, HW, (double*)scratch);
        NF_LAUNCH_CHECK("nf_in_act_pad_fwd (stats)");
    }
    hipLaunchKernelGGL(k_in_act_pad_fwd, dim3(nf_apply_splits(planes, HWp), (unsigned)planes), dim3(NF_CNN_BLOCK), 0, st, x, C, H, W,
                       gamma, beta, eps, (const double*)scratch, (int)stat_splits, res, rs_n, rs_c, rs_h, rs_w, act, pad, y_padded, mean,
                       rstd);
    NF_LAUNCH_CHECK("nf_in_act_pad_fwd");
    return 0;
}

extern "C" int nf_in_act_pad_bwd(const float* dy_padded, const float* d_extra, const float* y_padded, const float* x, int n_img,
                                 int C, int H, int W, const float* gamma, const float* mean, const float* rstd, int act, int pad,
                                 float* d_res, float* dx, void* scratch, nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && C >= 1 && H >= 1 && W >= 1 && pad >= 0 && pad < H && pad < W && act >= 0 && act <= 2 &&
                   (dy_padded || d_extra),
               "nf_in_act_pad_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int planes = n_img * C, HW = H * W;
    if (gamma) {
        NF_REQUIRE(scratch != nullptr, "nf_in_act_pad_bwd: scratch (512 bytes per plane) required with normalisation");
    }
    dim3 grid(nf_apply_splits(planes, HW), (unsigned)planes);
    hipLaunchKernelGGL(k_in_act_pad_bwd1, grid, dim3(NF_CNN_BLOCK), 0, st, dy_padded, d_extra, y_padded, x, H, W, mean, rstd,
                       gamma ? 1 : 0, act, pad, d_res, dx, (double*)scratch);
    NF_LAUNCH_CHECK("nf_in_act_pad_bwd (fold)");
    if (gamma) {
        hipLaunchKernelGGL(k_in_act_pad_bwd2, grid, dim3(NF_CNN_BLOCK), 0, st, x, C, HW, gamma, mean, rstd, (const double*)scratch,
                           (int)grid.x, d_res ? d_res : dx, dx);
        NF_LAUNCH_CHECK("nf_in_act_pad_bwd (norm)");
    }
    return 0;
}

extern "C" int nf_upsample2x_pad_fwd(const float* x, int64_t planes, int64_t xs_plane, int64_t xs_row, int h, int w, int pad,
                                     float* y_padded, nf_stream_t stream) {
    NF_REQUIRE(planes >= 1 && planes <= 0x7fffffff && h >= 1 && w >= 1 && pad >= 0 && pad < 2 * h && pad < 2 * w && xs_row >= w,
               "nf_upsample2x_pad_fwd: bad arguments (planes %lld h %d w %d pad %d)", (long long)planes, h, w, pad);
    const int HWp = (2 * h + 2 * pad) * (2 * w + 2 * pad);
    hipLaunchKernelGGL(k_upsample2x_pad, dim3(nf_apply_splits((int)planes, HWp), (unsigned)planes), dim3(NF_CNN_BLOCK), 0,
                       (hipStream_t)stream, x, xs_plane, xs_row, h, w, pad, y_padded);
    NF_LAUNCH_CHECK("nf_upsample2x_pad_fwd");
    return 0;
}
