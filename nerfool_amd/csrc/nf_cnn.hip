// Fused "glue" of the ResUNet feature extractor (a14): InstanceNorm + affine + residual + ReLU/ELU + reflect padding in
// ONE pass over a convolution output, writing the tensor the NEXT convolution consumes (pre-padded, so the hand-written
// convolutions -- nf_wino_bf.hip, nf_conv_s2.hip, nf_conv1x1.hip -- run with padding 0); and the matching backward (fold of
// the padded gradient, activation derivative, InstanceNorm backward).
//
// ref: ibrnet/feature_network.py:38-78 (BasicBlock: conv3x3 reflect -> IN -> ReLU -> conv3x3 reflect -> IN -> +id -> ReLU),
//      :127-140 (conv + IN + ELU), :28-36 (padding_mode='reflect').  PyTorch runs this as 4-5 memory-bound kernels per
//      convolution (reflection_pad2d, batch_norm x2 passes, add, relu) in each direction.
//
// Two fully parallel launches per direction: plane statistics (block partial sums in double, one fp64 atomic pair per
// block) and an element-wise apply; the second read of the convolution output comes from L2 / Infinity Cache.
#include "nf_common.h"

#define NF_CNN_BLOCK 256
#ifndef NF_GLUE_BLOCKS
#define NF_GLUE_BLOCKS 8192
#endif
#ifndef NF_GLUE_MIN_ELEMS
#define NF_GLUE_MIN_ELEMS 1024
#endif
#ifndef NF_PLANE_BWD_MAX
#define NF_PLANE_BWD_MAX 3
#endif
#ifndef NF_BWD1_UNROLL
#define NF_BWD1_UNROLL 2
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define NF_OPAQUE_INT(x) asm volatile("" : "+v"(x))      // the optimiser forgets what it knows about x (no instruction)
#else
#define NF_OPAQUE_INT(x) asm volatile("" : "+r"(x))
#endif
#define NF_PRAGMA(x) _Pragma(#x)
#define NF_UNROLL(n) NF_PRAGMA(unroll n)

typedef float nf_f4u __attribute__((ext_vector_type(4), aligned(4)));      // 16-byte access at 4-byte alignment

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

__device__ __forceinline__ float act_grad_from_pre(float v, int act) {      // same decisions as act_grad_from_out(act_fwd(v))
    if (act == 1) return v > 0.f ? 1.f : 0.f;
    if (act == 2) return v > 0.f ? 1.f : __expf(v);
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
                                                                 float* __restrict__ rstd_out, int64_t y_n_stride) {
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
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const float* xp = x + p * HW;
    const float* rp = res ? res + n * rs_n + c * rs_c : nullptr;
    float* out = yp + n * y_n_stride + c * (int64_t)Hp * Wp;
    // one work item = 4 consecutive interior columns of one padded row (16-byte accesses; the padded rows are only
    // 4-byte aligned, hence the unaligned vector type); the items next to the left / right border also fill the pad columns
    const int G = (W + 3) >> 2, total = Hp * G;
    int seg = (total + gridDim.x - 1) / gridDim.x;
    int lo = blockIdx.x * seg, hi = min(lo + seg, total);
#pragma unroll 4
    for (int idx = lo + threadIdx.x; idx < hi; idx += blockDim.x) {
        int ph = idx / G, gq = idx - ph * G, w0 = gq << 2;
        int sh = reflect_src(ph - pad, H);
        const float* xr = xp + sh * W;
        const float* rr = rp ? rp + sh * rs_h : nullptr;
        float* orow = out + (int64_t)ph * Wp + pad;
        if (w0 + 4 <= W) {
            nf_f4u v = *reinterpret_cast<const nf_f4u*>(xr + w0);
            nf_f4u o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[j] - mean) * rstd * g + b;
            if (rr) {
                if (rs_w == 1) {
                    nf_f4u r = *reinterpret_cast<const nf_f4u*>(rr + w0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] += r[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] += rr[(w0 + j) * rs_w];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = act_fwd(o[j], act);
            *reinterpret_cast<nf_f4u*>(orow + w0) = o;
        } else {
            for (int w = w0; w < W; ++w) {
                float v = (xr[w] - mean) * rstd * g + b;
                if (rr) v += rr[w * rs_w];
                orow[w] = act_fwd(v, act);
            }
        }
        if (gq == 0 || gq == G - 1) {          // pad columns of this row: left ones mirror columns 1..pad, right ones W-2..W-1-pad
            for (int k = 0; k < pad; ++k) {
                int sw = gq == 0 ? pad - k : W - 2 - k;
                int pw = gq == 0 ? k : pad + W + k;
                float v = (xr[sw] - mean) * rstd * g + b;
                if (rr) v += rr[sw * rs_w];
                out[(int64_t)ph * Wp + pw] = act_fwd(v, act);
            }
            if (G == 1) {                       // a single item per row owns both borders
                for (int k = 0; k < pad; ++k) {
                    int sw = W - 2 - k;
                    float v = (xr[sw] - mean) * rstd * g + b;
                    if (rr) v += rr[sw * rs_w];
                    out[(int64_t)ph * Wp + pad + W + k] = act_fwd(v, act);
                }
            }
        }
    }
}

// ---- work items: 4 consecutive columns [w0, w0+4) of one row h (the last item of a row is ragged when W % 4 != 0) --------
__device__ __forceinline__ nf_f4u load_item(const float* __restrict__ p, int nvalid) {
    if (nvalid == 4) return *reinterpret_cast<const nf_f4u*>(p);
    nf_f4u v = nf_f4u{0.f, 0.f, 0.f, 0.f};
    if (nvalid > 0) v[0] = p[0];
    if (nvalid > 1) v[1] = p[1];
    if (nvalid > 2) v[2] = p[2];
    return v;
}

__device__ __forceinline__ void store_item(float* __restrict__ p, nf_f4u v, int nvalid) {
    if (nvalid == 4) {
        *reinterpret_cast<nf_f4u*>(p) = v;
    } else {
        if (nvalid > 0) p[0] = v[0];
        if (nvalid > 1) p[1] = v[1];
        if (nvalid > 2) p[2] = v[2];
    }
}

// gradient reaching the interior pixels of an item: their own padded positions plus every padded position that mirrors
// onto them (rows pad-h / 2(H-1)+pad-h for rows next to the border; columns likewise, only in the first / last items)
__device__ __forceinline__ nf_f4u fold_item(const float* __restrict__ gp, int h, int w0, int nvalid, int H, int W, int pad, int Wp) {
    const bool top = h >= 1 && h <= pad, bot = h <= H - 2 && h >= H - 1 - pad;
    const float* r0 = gp + (h + pad) * Wp;
    const float* r1 = gp + (pad - h) * Wp;
    const float* r2 = gp + (2 * (H - 1) + pad - h) * Wp;
    nf_f4u d = load_item(r0 + pad + w0, nvalid);
    if (top) d += load_item(r1 + pad + w0, nvalid);
    if (bot) d += load_item(r2 + pad + w0, nvalid);
    if (w0 <= pad || w0 + 3 >= W - 1 - pad) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int wj = w0 + j;
            if (j < nvalid) {
                if (wj >= 1 && wj <= pad) {
                    const int cm = pad - wj;
                    d[j] += r0[cm];
                    if (top) d[j] += r1[cm];
                    if (bot) d[j] += r2[cm];
                }
                if (wj <= W - 2 && wj >= W - 1 - pad) {
                    const int cm = 2 * (W - 1) + pad - wj;
                    d[j] += r0[cm];
                    if (top) d[j] += r1[cm];
                    if (bot) d[j] += r2[cm];
                }
            }
        }
    }
    return d;
}

// d_pre of an item = (d_extra + folded padded gradient) * activation'(.), and the item of x (zeros beyond nvalid)
__device__ __forceinline__ void item_d_pre(const float* __restrict__ gp, const float* __restrict__ ep, const float* __restrict__ yq,
                                           const float* __restrict__ xq, int h, int w0, int nvalid, int H, int W, int pad, int Wp,
                                           int act, bool from_x, float mean, float rstd, float ga, float be, nf_f4u& d, nf_f4u& xv,
                                           const float* __restrict__ es) {
    const int i0 = h * W + w0;
    d = ep ? load_item(ep + i0, nvalid) : nf_f4u{0.f, 0.f, 0.f, 0.f};
    if (gp) d += fold_item(gp, h, w0, nvalid, H, W, pad, Wp);
    if (es && !(h & 1)) {       // gradient of a stride-2 consumer of the unpadded output (1x1 downsample): even rows / columns
        const float* er = es + (h >> 1) * ((W + 1) >> 1) + (w0 >> 1);      // w0 is a multiple of 4
        if (nvalid > 0) d[0] += er[0];
        if (nvalid > 2) d[2] += er[1];
    }
    xv = xq ? load_item(xq + i0, nvalid) : nf_f4u{0.f, 0.f, 0.f, 0.f};
    if (from_x) {
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] *= act_grad_from_pre((xv[j] - mean) * rstd * ga + be, act);
    } else if (act != 0) {
        nf_f4u y = load_item(yq + (h + pad) * Wp + pad + w0, nvalid);
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] *= act_grad_from_out(y[j], act);
    }
}

// ---- backward, pass 1: fold the padded gradient, activation derivative -> d_pre (written to d_res, else dx), and the two
//      plane sums of the InstanceNorm backward.  grid (chunks of the plane, planes)
__global__ void __launch_bounds__(NF_CNN_BLOCK) k_in_act_pad_bwd1(const float* __restrict__ dyp, const float* __restrict__ d_extra,
                                                                  const float* __restrict__ yp, const float* __restrict__ x,
                                                                  int H, int W, const float* __restrict__ mean_in,
                                                                  const float* __restrict__ rstd_in, int has_norm, int act, int pad,
                                                                  float* __restrict__ d_res, float* __restrict__ dx,
                                                                  double* __restrict__ sums, int from_x, int C,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  int64_t dy_n_stride, const float* __restrict__ d_extra_sub) {
    const int64_t p = blockIdx.y;
    // from_x: the activation derivative is recomputed from x (no residual went into the activation), y_padded is not read
    const float ga = from_x ? gamma[p % C] : 1.f, be = from_x ? beta[p % C] : 0.f;
    const int HW = H * W, Hp = H + 2 * pad, Wp = W + 2 * pad;
    const float* gp = dyp ? dyp + (p / C) * dy_n_stride + (p % C) * (int64_t)Hp * Wp : nullptr;
    const float* ep = d_extra ? d_extra + p * HW : nullptr;
    const float* es = d_extra_sub ? d_extra_sub + p * (int64_t)((H + 1) >> 1) * ((W + 1) >> 1) : nullptr;
    const float* yq = yp ? yp + p * (int64_t)Hp * Wp : nullptr;
    const float* xq = has_norm ? x + p * HW : nullptr;
    float* dxp = dx + p * HW;
    float* drp = d_res ? d_res + p * HW : nullptr;
    const float mean = has_norm ? mean_in[p] : 0.f, rstd = has_norm ? rstd_in[p] : 1.f;
    // with a residual output AND a normalisation pass behind, d_pre is stored once (in d_res) and pass 2 reads it there
    const bool store_dx = !drp || !has_norm;
    const int G = (W + 3) >> 2, total = H * G;
    int seg = (total + gridDim.x - 1) / gridDim.x;
    int lo = blockIdx.x * seg, hi = min(lo + seg, total);
    double s1 = 0.0, s2 = 0.0;
    NF_UNROLL(NF_BWD1_UNROLL)
    for (int idx = lo + threadIdx.x; idx < hi; idx += blockDim.x) {
        const int h = idx / G, w0 = (idx - h * G) << 2, nvalid = min(4, W - w0), i0 = h * W + w0;
        nf_f4u d, xv;
        item_d_pre(gp, ep, yq, xq, h, w0, nvalid, H, W, pad, Wp, act, from_x != 0, mean, rstd, ga, be, d, xv, es);
        if (drp) store_item(drp + i0, d, nvalid);
        if (store_dx) store_item(dxp + i0, d, nvalid);
        if (has_norm) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < nvalid) {
                    float xh = (xv[j] - mean) * rstd;
                    s1 += (double)d[j];
                    s2 += (double)d[j] * xh;
                }
            }
        }
    }
    if (has_norm) block_store_sums(s1, s2, sums + (p * NF_MAX_SPLITS + blockIdx.x) * 2);
}

// ---- backward, pass 2: dx = gamma rstd (d_pre - mean(d_pre) - xhat mean(d_pre xhat))
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
    const int G = (HW + 3) >> 2;
    int seg = (G + gridDim.x - 1) / gridDim.x;
    int lo = blockIdx.x * seg, hi = min(lo + seg, G);
#pragma unroll 4
    for (int gq = lo + threadIdx.x; gq < hi; gq += blockDim.x) {
        const int i0 = gq << 2;
        if (i0 + 4 <= HW) {
            nf_f4u xv = *reinterpret_cast<const nf_f4u*>(xq + i0);
            nf_f4u dv = *reinterpret_cast<const nf_f4u*>(dpp + i0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float xh = (xv[j] - mean) * rstd;
                dv[j] = gr * (dv[j] - m1 - xh * m2);
            }
            *reinterpret_cast<nf_f4u*>(dxp + i0) = dv;
        } else {
            for (int i = i0; i < HW; ++i) {
                float xh = (xq[i] - mean) * rstd;
                dxp[i] = gr * (dpp[i] - m1 - xh * m2);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Plane-resident variants (planes of up to NT * EPT items): ONE workgroup owns a whole (image, channel) plane and keeps it
// in registers between the statistics and the apply step -- one launch and one read of the convolution output instead of
// two (forward 2 instead of 3 tensor passes, backward 3-4 instead of 7).
// ---------------------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void block_sum2(double& a, double& b) {
    __shared__ double red[2][NT / 64];
    __shared__ double tot[2];
    a = wave_sum_f64(a);
    b = wave_sum_f64(b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int w = 0; w < NT / 64; ++w) { ta += red[0][w]; tb += red[1][w]; }
        tot[0] = ta;
        tot[1] = tb;
    }
    __syncthreads();
    a = tot[0];
    b = tot[1];
}

template <int NT, int EPT>
__global__ void __launch_bounds__(NT) k_plane_fwd(const float* __restrict__ x, int C, int H, int W, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float eps, const float* __restrict__ res,
                                                  int64_t rs_n, int64_t rs_c, int64_t rs_h, int64_t rs_w, int act, int pad,
                                                  float* __restrict__ yp, float* __restrict__ mean_out,
                                                  float* __restrict__ rstd_out, int64_t y_n_stride) {
    const int64_t p = blockIdx.x;
    const int n = (int)(p / C), c = (int)(p - (int64_t)n * C);
    const int HW = H * W, Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int G = (W + 3) >> 2, total = H * G;
    const float* xp = x + p * HW;
    nf_f4u xv[EPT];
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int idx = threadIdx.x + k * NT;
        const int h = idx / G, w0 = (idx - h * G) << 2;
        xv[k] = load_item(xp + h * W + w0, idx < total ? min(4, W - w0) : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sa += (double)xv[k][j];
            sb += (double)xv[k][j] * xv[k][j];
        }
    }
    block_sum2<NT>(sa, sb);
    const double m = sa / (double)HW;
    const double var = sb / (double)HW - m * m;
    const float mean = (float)m;
    const float rstd = 1.f / sqrtf(fmaxf((float)var, 0.f) + eps);
    const float g = gamma[c], b = beta[c];
    if (threadIdx.x == 0) {
        mean_out[p] = mean;
        rstd_out[p] = rstd;
    }
    const float* rp = res ? res + n * rs_n + c * rs_c : nullptr;
    float* out = yp + n * y_n_stride + c * (int64_t)Hp * Wp;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int idx = threadIdx.x + k * NT;
        if (idx >= total) continue;
        const int h = idx / G, w0 = (idx - h * G) << 2, nvalid = min(4, W - w0);
        nf_f4u o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (xv[k][j] - mean) * rstd * g + b;
        if (rp) {
            const float* rr = rp + h * rs_h + w0 * rs_w;
            if (rs_w == 1) {
                o += load_item(rr, nvalid);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nvalid) o[j] += rr[j * rs_w];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = act_fwd(o[j], act);
        // the item goes to its own padded row and to the padded rows mirroring it; first / last items also fill pad columns
        const bool top = h >= 1 && h <= pad, bot = h <= H - 2 && h >= H - 1 - pad;
        const bool edge = w0 <= pad || w0 + 3 >= W - 1 - pad;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if ((a == 1 && !top) || (a == 2 && !bot)) continue;
            const int row = a == 0 ? h + pad : (a == 1 ? pad - h : 2 * (H - 1) + pad - h);
            float* orow = out + (int64_t)row * Wp;
            store_item(orow + pad + w0, o, nvalid);
            if (edge) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int wj = w0 + j;
                    if (j < nvalid) {
                        if (wj >= 1 && wj <= pad) orow[pad - wj] = o[j];
                        if (wj <= W - 2 && wj >= W - 1 - pad) orow[2 * (W - 1) + pad - wj] = o[j];
                    }
                }
            }
        }
    }
}

// KEEP_XH: xhat stays in registers between the two steps; otherwise (48 floats per thread already hold d_pre) the second
// step re-reads x, which the first step just pulled through L2 / Infinity Cache
// XL > 0 (with KEEP_XH false): xhat of the first XL items per thread waits in LDS between the two steps instead of being re-read
template <int NT, int EPT, bool KEEP_XH, int XL = 0>
__global__ void __launch_bounds__(NT) k_plane_bwd(const float* __restrict__ dyp, const float* __restrict__ d_extra,
                                                  const float* __restrict__ yp, const float* __restrict__ x, int C, int H, int W,
                                                  const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                  const float* __restrict__ rstd_in, int act, int pad, float* __restrict__ d_res,
                                                  float* __restrict__ dx, const float* __restrict__ beta, int64_t dy_n_stride,
                                                  const float* __restrict__ d_extra_sub) {
    const int64_t p = blockIdx.x;
    const int c = (int)(p % C);
    const bool from_x = beta != nullptr;        // activation derivative recomputed from x, y_padded not read
    const float ga = gamma[c], be = from_x ? beta[c] : 0.f;
    const int HW = H * W, Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int G = (W + 3) >> 2, total = H * G;
    const float* gp = dyp ? dyp + (p / C) * dy_n_stride + c * (int64_t)Hp * Wp : nullptr;
    const float* ep = d_extra ? d_extra + p * HW : nullptr;
    const float* es = d_extra_sub ? d_extra_sub + p * (int64_t)((H + 1) >> 1) * ((W + 1) >> 1) : nullptr;
    const float* yq = yp ? yp + p * (int64_t)Hp * Wp : nullptr;
    const float* xq = x + p * HW;
    const float mean = mean_in[p], rstd = rstd_in[p];
    nf_f4u dv[EPT], xhs[KEEP_XH ? EPT : 1];
    HIP_DYNAMIC_SHARED(nf_f4u, xh_lds)          // [XL][NT] items
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        // compiler barrier every 2 items: bounds how many items' loads (up to 6 x 16 bytes each) are hoisted at once
        if (EPT > 4 && (k & 1) == 0 && k > 0) asm volatile("" ::: "memory");
        const int idx = threadIdx.x + k * NT;
        const int h = idx / G, w0 = (idx - h * G) << 2, nvalid = idx < total ? min(4, W - w0) : 0;
        nf_f4u xv;
        if (nvalid > 0) {
            item_d_pre(gp, ep, yq, xq, h, w0, nvalid, H, W, pad, Wp, act, from_x, mean, rstd, ga, be, dv[k], xv, es);
        } else {
            dv[k] = nf_f4u{0.f, 0.f, 0.f, 0.f};
            xv = nf_f4u{mean, mean, mean, mean};
        }
        nf_f4u xh;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xh[j] = j < nvalid ? (xv[j] - mean) * rstd : 0.f;
            if (j >= nvalid) dv[k][j] = 0.f;
            s1 += (double)dv[k][j];
            s2 += (double)dv[k][j] * xh[j];
        }
        if (KEEP_XH) xhs[k] = xh;
        else if (k < XL) xh_lds[k * NT + threadIdx.x] = xh;      // (read back by the same thread)
    }
    block_sum2<NT>(s1, s2);
    const float m1 = (float)(s1 / (double)HW), m2 = (float)(s2 / (double)HW);
    const float gr = ga * rstd;
    float* dxp = dx + p * HW;
    float* drp = d_res ? d_res + p * HW : nullptr;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        if (EPT > 4 && (k & 3) == 0 && k > 0) asm volatile("" ::: "memory");       // (the x re-reads of at most 4 items in flight)
        int idx = threadIdx.x + k * NT;
        // (the item's row / column are derived AGAIN here: remembered from the first loop they cost 3 registers per item, which at
        //  16 items is what spilled)
        if (EPT > 4) NF_OPAQUE_INT(idx);
        if (idx >= total) continue;
        const int h = idx / G, w0 = (idx - h * G) << 2, nvalid = min(4, W - w0), i0 = h * W + w0;
        nf_f4u o, xh;
        if (KEEP_XH) {
            xh = xhs[k];
        } else if (k < XL) {
            xh = xh_lds[k * NT + threadIdx.x];
        } else {
            nf_f4u xv = load_item(xq + i0, nvalid);
#pragma unroll
            for (int j = 0; j < 4; ++j) xh[j] = (xv[j] - mean) * rstd;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = gr * (dv[k][j] - m1 - xh[j] * m2);
        store_item(dxp + i0, o, nvalid);
        if (drp) store_item(drp + i0, dv[k], nvalid);
    }
}

// which plane-resident instantiation holds an H x W plane: 0 none, 1 (256 threads x 3 items), 2 (1024 x 3), 3 (768 x 16: 64 registers
// of gradient per thread at three waves per SIMD -- 1024 x 12 at four waves per SIMD spilled 46 of its 48, two extra passes of the
// tensor through scratch memory: 292 MB per launch where 195 MB are the tensors)
#ifndef NF_PLANE3_XL
#define NF_PLANE3_XL 13
#endif
static int nf_plane_variant(int H, int W) {
    const int64_t items = (int64_t)H * ((W + 3) / 4);
    if (items <= 256 * 3) return 1;
    if (items <= 1024 * 3) return 2;
    if (items <= 768 * 16) return 3;
    return 0;
}

// element-wise passes: up to NF_GLUE_BLOCKS workgroups of 256 threads over the chip, each streaming >= NF_GLUE_MIN_ELEMS
// elements of one plane in 16-byte items (tools/bench_cnn_glue.py: 2048 / 4096 / 8192 workgroups are within 3 %)
static unsigned nf_apply_splits(int planes, int n, int max_splits = NF_MAX_SPLITS) {
    int splits = (NF_GLUE_BLOCKS + planes - 1) / planes;
    int cap = n / NF_GLUE_MIN_ELEMS;
    if (splits > cap) splits = cap;
    if (splits > max_splits) splits = max_splits;
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
    // work item = 4 consecutive columns of one padded row (one 16-byte store); the taps come from the 4x smaller source
    const int G = (Wp + 3) >> 2, total = Hp * G;
    int seg = (total + gridDim.x - 1) / gridDim.x;
    int lo = blockIdx.x * seg, hi = min(lo + seg, total);
#pragma unroll 2
    for (int idx = lo + threadIdx.x; idx < hi; idx += blockDim.x) {
        const int ph = idx / G, pw0 = (idx - ph * G) << 2, nvalid = min(4, Wp - pw0);
        const int oh = reflect_src(ph - pad, H);
        const float h1r = rh * (float)oh;
        const int h1 = (int)h1r;
        const int h1p = h1 < h - 1 ? 1 : 0;
        const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
        const float* r0 = xp + (int64_t)h1 * xs_row;
        const float* r1 = r0 + (int64_t)h1p * xs_row;
        nf_f4u o = nf_f4u{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < nvalid) {
                const int ow = reflect_src(pw0 + j - pad, W);
                const float w1r = rw * (float)ow;
                const int w1 = (int)w1r;
                const int w1p = w1 < w - 1 ? 1 : 0;
                const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
                o[j] = h0l * (w0l * r0[w1] + w1l * r0[w1 + w1p]) + h1l * (w0l * r1[w1] + w1l * r1[w1 + w1p]);
            }
        }
        store_item(out + (int64_t)ph * Wp + pw0, o, nvalid);
    }
}

extern "C" int nf_in_act_pad_fwd(const float* x, int n_img, int C, int H, int W, const float* gamma, const float* beta, float eps,
                                 const float* res, int64_t rs_n, int64_t rs_c, int64_t rs_h, int64_t rs_w, int act, int pad,
                                 float* y_padded, int64_t y_n_stride, float* mean, float* rstd, void* scratch, nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && C >= 1 && H >= 1 && W >= 1 && pad >= 0 && pad < H && pad < W && act >= 0 && act <= 2,
               "nf_in_act_pad_fwd: bad arguments (N %d C %d H %d W %d pad %d act %d)", n_img, C, H, W, pad, act);
    hipStream_t st = (hipStream_t)stream;
    const int planes = n_img * C, HW = H * W, HWp = (H + 2 * pad) * (W + 2 * pad);
    if (y_n_stride == 0) y_n_stride = (int64_t)C * HWp;       // packed [N,C,Hp,Wp]
    int variant = gamma ? nf_plane_variant(H, W) : 0;
    // (variant 3 was slower than the two-pass form as 1024 threads x 12 items -- one workgroup per CU serialising load / reduce / store;
    //  as 768 x 16 it is faster: 189 x 252 planes 34.6 -> 22.2 us, one read of the convolution output instead of two)
    if (variant) {
#define NF_PLANE_FWD(NT, EPT)                                                                                                   \
    hipLaunchKernelGGL((k_plane_fwd<NT, EPT>), dim3((unsigned)planes), dim3(NT), 0, st, x, C, H, W, gamma, beta, eps, res, rs_n, \
                       rs_c, rs_h, rs_w, act, pad, y_padded, mean, rstd, y_n_stride)
        if (variant == 1) NF_PLANE_FWD(256, 3);
        else if (variant == 2) NF_PLANE_FWD(1024, 3);
        else NF_PLANE_FWD(768, 16);
#undef NF_PLANE_FWD
        NF_LAUNCH_CHECK("nf_in_act_pad_fwd (plane)");
        return 0;
    }
    unsigned stat_splits = 1;
    if (gamma) {
        NF_REQUIRE(scratch != nullptr, "nf_in_act_pad_fwd: scratch (512 bytes per plane) required with normalisation");
        stat_splits = nf_stat_splits(planes, HW);
        hipLaunchKernelGGL(k_in_stats, dim3(stat_splits, (unsigned)planes), dim3(NF_CNN_BLOCK), 0, st, x, HW, (double*)scratch);
        NF_LAUNCH_CHECK("nf_in_act_pad_fwd (stats)");
    }
    hipLaunchKernelGGL(k_in_act_pad_fwd, dim3(nf_apply_splits(planes, HWp, 1024), (unsigned)planes), dim3(NF_CNN_BLOCK), 0, st, x, C, H, W,
                       gamma, beta, eps, (const double*)scratch, (int)stat_splits, res, rs_n, rs_c, rs_h, rs_w, act, pad, y_padded, mean,
                       rstd, y_n_stride);
    NF_LAUNCH_CHECK("nf_in_act_pad_fwd");
    return 0;
}

extern "C" int nf_in_act_pad_bwd(const float* dy_padded, const float* d_extra, const float* y_padded, const float* x, int n_img,
                                 int C, int H, int W, const float* gamma, const float* beta, const float* mean, const float* rstd,
                                 int act, int pad, float* d_res, float* dx, void* scratch, int64_t dy_n_stride,
                                 const float* d_extra_sub, nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && C >= 1 && H >= 1 && W >= 1 && pad >= 0 && pad < H && pad < W && act >= 0 && act <= 2 &&
                   (dy_padded || d_extra || d_extra_sub),
               "nf_in_act_pad_bwd: bad arguments");
    if (dy_n_stride == 0) dy_n_stride = (int64_t)C * (H + 2 * pad) * (W + 2 * pad);
    // without a residual input the pre-activation is a function of x alone: its derivative is recomputed, y_padded stays unread
    const bool from_x = gamma && beta && !d_res && act != 0;
    NF_REQUIRE(from_x || y_padded || act == 0, "nf_in_act_pad_bwd: y_padded is required (residual input, or no beta given)");
    hipStream_t st = (hipStream_t)stream;
    const int planes = n_img * C, HW = H * W;
    int variant = gamma ? nf_plane_variant(H, W) : 0;
    if (variant > NF_PLANE_BWD_MAX) variant = 0;
#ifdef NF_PLANE_BWD_NO_RES3
    if (variant == 3 && d_res) variant = 0;      // (round 2, with the spilling 1024 x 12 form: the two-pass form was faster with a d_res stream)
#endif
    if (variant) {
#define NF_PLANE_BWD(NT, EPT, KEEP, XL)                                                                                                       \
    hipLaunchKernelGGL((k_plane_bwd<NT, EPT, KEEP, XL>), dim3((unsigned)planes), dim3(NT), (size_t)(XL) * NT * 16, st, dy_padded, d_extra, y_padded, \
                       x, C, H, W, gamma, mean, rstd, act, pad, d_res, dx, from_x ? beta : nullptr, dy_n_stride, d_extra_sub)
        if (variant == 1) NF_PLANE_BWD(256, 3, true, 0);
        else if (variant == 2) NF_PLANE_BWD(1024, 3, true, 0);
        else {
            constexpr int XL = NF_PLANE3_XL;        // XL x 768 x 16 bytes of LDS (one workgroup of 12 waves per CU either way)
            static bool once_on[NF_MAX_DEVICES] = {};
            bool& once = once_on[nf_current_device()];
            if (!once) {
                if (XL * 768 * 16 > 64 * 1024 &&
                    hipFuncSetAttribute((const void*)k_plane_bwd<768, 16, false, XL>, hipFuncAttributeMaxDynamicSharedMemorySize, XL * 768 * 16) !=
                        hipSuccess) {
                    nf_set_error("nf_in_act_pad_bwd: cannot reserve %d bytes of LDS", XL * 768 * 16);
                    return 1;
                }
                once = true;
            }
            NF_PLANE_BWD(768, 16, false, XL);
        }
#undef NF_PLANE_BWD
        NF_LAUNCH_CHECK("nf_in_act_pad_bwd (plane)");
        return 0;
    }
    if (gamma) {
        NF_REQUIRE(scratch != nullptr, "nf_in_act_pad_bwd: scratch (512 bytes per plane) required with normalisation");
    }
    dim3 grid(nf_apply_splits(planes, HW), (unsigned)planes);
    hipLaunchKernelGGL(k_in_act_pad_bwd1, grid, dim3(NF_CNN_BLOCK), 0, st, dy_padded, d_extra, y_padded, x, H, W, mean, rstd,
                       gamma ? 1 : 0, act, pad, d_res, dx, (double*)scratch, from_x ? 1 : 0, C, gamma, beta, dy_n_stride, d_extra_sub);
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
    hipLaunchKernelGGL(k_upsample2x_pad, dim3(nf_apply_splits((int)planes, HWp, 1024), (unsigned)planes), dim3(NF_CNN_BLOCK), 0,
                       (hipStream_t)stream, x, xs_plane, xs_row, h, w, pad, y_padded);
    NF_LAUNCH_CHECK("nf_upsample2x_pad_fwd");
    return 0;
}
