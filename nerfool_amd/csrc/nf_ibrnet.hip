// Generic IBRNet kernels (a4/a5): any number of views, any S <= 1024.  One workgroup per ray, one thread per sample;
// the thread walks the V views of its sample serially, so every cross-view reduction (pooling, blending softmax) is
// thread-local, and the only cross-thread step is the ray self-attention over the S samples (through the workspace,
// ordered by __syncthreads).  Layer activations stream through a per-ray global workspace laid out [slot][view][sample]
// (sample fastest => coalesced); weights are read with wave-uniform addresses (scalar loads).
//
// These kernels are the shape-generic path and the on-device cross-check of the MFMA tile kernels in
// nf_ibrnet_mfma.hip (V <= 32).
//
// ref: ibrnet/mlp_network.py:222-274 (IBRNet.forward), :69-119 (MultiHeadAttention), :23-43 (attention core).
// The backward follows oracle/ibrnet_manual_bwd.py step by step (d/d rgb_feat only).
#include "nf_ibrnet.h"
#include "nf_dense.h"

#include <string.h>

__device__ __forceinline__ float nf_elu(float x) { return x > 0.f ? x : expm1f(x); }
__device__ __forceinline__ float nf_elu_grad(float y) { return y > 0.f ? 1.f : y + 1.f; }   // from the ELU OUTPUT
__device__ __forceinline__ float nf_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

struct IbrCtx {
    const float* __restrict__ blob;
    const float* __restrict__ pos_enc;
    const float* __restrict__ rgb_feat;   // this ray: [S,V,35]
    const float* __restrict__ ray_diff;   // this ray: [S,V,4]
    const float* __restrict__ mask;       // this ray: [S,V]
    float* ws_row;                        // this ray: [slot][V][S]
    float* ws_smp;                        // this ray: [slot][S]
    int S, V, aa, s;
};

#define ROWP(slot) (c.ws_row + ((size_t)(slot) * c.V + v) * c.S + c.s)
#define ROWSTRIDE ((size_t)c.V * c.S)
#define ROW(slot, j) ROWP(slot)[(size_t)(j) * ROWSTRIDE]
#define SMPP(slot) (c.ws_smp + (size_t)(slot) * c.S + c.s)
#define SMP(slot, j) SMPP(slot)[(size_t)(j) * c.S]

// ---------------------------------------------------------------------------------------------------------------
// forward, per-sample part (everything except the ray attention).  Stores every activation the backward needs.
// Returns the blended colour; q/k/v of the sample are left in the workspace for the attention phase.
// ---------------------------------------------------------------------------------------------------------------
__device__ void ibr_fwd_sample(const IbrCtx& c, float (&rgb)[3]) {
    const float* B = c.blob;
    const int V = c.V;
    float emin = 3.0e38f, nval = 0.f;
    const float s_abs = fabsf(B[0]);
    // ---- direction MLP, f = rgb_feat + dir_feat, exp_dot_prod   (mlp_network.py:231-237)
    for (int v = 0; v < V; ++v) {
        const float* rd = c.ray_diff + ((size_t)c.s * V + v) * 4;
        const float* rf = c.rgb_feat + ((size_t)c.s * V + v) * 35;
        float d1[16];
        nf_load_bias(B + nf_lin_b(NF_L_DIR0), d1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float xk = rd[k];
#pragma unroll
            for (int n = 0; n < 16; ++n) d1[n] = fmaf(B[nf_lin_wt(NF_L_DIR0) + k * 16 + n], xk, d1[n]);
        }
#pragma unroll
        for (int n = 0; n < 16; ++n) d1[n] = nf_elu(d1[n]);
        float df[35];
        nf_load_bias(B + nf_lin_b(NF_L_DIR1), df);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
#pragma unroll
            for (int n = 0; n < 35; ++n) df[n] = fmaf(B[nf_lin_wt(NF_L_DIR1) + k * 35 + n], d1[k], df[n]);
        }
#pragma unroll
        for (int n = 0; n < 35; ++n) ROW(NF_R_F, n) = rf[n] + nf_elu(df[n]);
        nval += c.mask[(size_t)c.s * V + v];
        if (c.aa) {
            float e = expf(s_abs * (rd[3] - 1.f));
            ROW(NF_R_BETA, 0) = e;
            emin = fminf(emin, e);
        }
    }
    // ---- first pooling weight (constant w.r.t. the features)   (:236-241)
    float wsum = 0.f;
    for (int v = 0; v < V; ++v) {
        float mk = c.mask[(size_t)c.s * V + v];
        float w = c.aa ? (ROW(NF_R_BETA, 0) - emin) * mk : mk;
        ROW(NF_R_W, 0) = w;
        wsum += w;
    }
    wsum += 1e-8f;
    for (int v = 0; v < V; ++v) ROW(NF_R_W, 0) = ROW(NF_R_W, 0) / wsum;
    // ---- weighted mean / variance over views   (:144-149)
    {
        float mean[35];
#pragma unroll
        for (int j = 0; j < 35; ++j) mean[j] = 0.f;
        for (int v = 0; v < V; ++v) {
            float w = ROW(NF_R_W, 0);
#pragma unroll
            for (int j = 0; j < 35; ++j) mean[j] += ROW(NF_R_F, j) * w;
        }
        float var[35];
#pragma unroll
        for (int j = 0; j < 35; ++j) var[j] = 0.f;
        for (int v = 0; v < V; ++v) {
            float w = ROW(NF_R_W, 0);
#pragma unroll
            for (int j = 0; j < 35; ++j) {
                float d = ROW(NF_R_F, j) - mean[j];
                var[j] += w * (d * d);
            }
        }
#pragma unroll
        for (int j = 0; j < 35; ++j) {
            SMP(NF_S_MEAN, j) = mean[j];
            SMP(NF_S_VAR, j) = var[j];
        }
    }
    // ---- base_fc / vis_fc / vis_fc2 per view   (:246-255)
    float hs[64];   // shared (mean, var) part of base_fc.0, computed once per sample
    nf_load_bias(B + nf_lin_b(NF_L_BASE0), hs);
    nf_dense_ws<64>(B + nf_lin_wt(NF_L_BASE0), 70, SMPP(NF_S_MEAN), (size_t)c.S, 1.f, hs);
    float vsum = 0.f;
    for (int v = 0; v < V; ++v) {
        float mk = c.mask[(size_t)c.s * V + v];
        float w = ROW(NF_R_W, 0);
        {
            float h1[64];
#pragma unroll
            for (int n = 0; n < 64; ++n) h1[n] = hs[n];
            nf_dense_ws<64>(B + nf_lin_wt(NF_L_BASE0) + 70 * 64, 35, ROWP(NF_R_F), ROWSTRIDE, 1.f, h1);
#pragma unroll
            for (int n = 0; n < 64; ++n) ROW(NF_R_H1, n) = nf_elu(h1[n]);
        }
        float h[32];
        nf_load_bias(B + nf_lin_b(NF_L_BASE1), h);
        nf_dense_ws<32>(B + nf_lin_wt(NF_L_BASE1), 64, ROWP(NF_R_H1), ROWSTRIDE, 1.f, h);
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            h[n] = nf_elu(h[n]);
            ROW(NF_R_H, n) = h[n];
        }
        {
            float v1[32];
            nf_load_bias(B + nf_lin_b(NF_L_VIS0), v1);
            nf_dense_ws<32>(B + nf_lin_wt(NF_L_VIS0), 32, ROWP(NF_R_H), ROWSTRIDE, w, v1);
#pragma unroll
            for (int n = 0; n < 32; ++n) ROW(NF_R_V1, n) = nf_elu(v1[n]);
        }
        float xv[33];
        nf_load_bias(B + nf_lin_b(NF_L_VIS1), xv);
        nf_dense_ws<33>(B + nf_lin_wt(NF_L_VIS1), 32, ROWP(NF_R_V1), ROWSTRIDE, 1.f, xv);
#pragma unroll
        for (int n = 0; n < 33; ++n) {
            xv[n] = nf_elu(xv[n]);
            ROW(NF_R_XV, n) = xv[n];
        }
        float sig1 = nf_sigmoid(xv[32]);
        float vis1 = sig1 * mk;
        ROW(NF_R_SIG1, 0) = sig1;
        ROW(NF_R_VIS1, 0) = vis1;
#pragma unroll
        for (int n = 0; n < 32; ++n) ROW(NF_R_X2, n) = h[n] + xv[n];
        float u[32];
        nf_load_bias(B + nf_lin_b(NF_L_VISB0), u);
        nf_dense_ws<32>(B + nf_lin_wt(NF_L_VISB0), 32, ROWP(NF_R_X2), ROWSTRIDE, vis1, u);
        float z2 = B[nf_lin_b(NF_L_VISB1)];
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            u[n] = nf_elu(u[n]);
            ROW(NF_R_U, n) = u[n];
            z2 = fmaf(B[nf_lin_wt(NF_L_VISB1) + n], u[n], z2);
        }
        float sig2 = nf_sigmoid(z2);
        float vis2 = sig2 * mk;
        ROW(NF_R_SIG2, 0) = sig2;
        ROW(NF_R_VIS2, 0) = vis2;
        vsum += vis2;
    }
    vsum += 1e-8f;
    // ---- second pooling, geometry_fc   (:255-259)
    {
        float wmean = 0.f;
        for (int v = 0; v < V; ++v) {
            float w2 = ROW(NF_R_VIS2, 0) / vsum;
            ROW(NF_R_W2, 0) = w2;
            wmean += w2;
        }
        float mean2[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) mean2[j] = 0.f;
        for (int v = 0; v < V; ++v) {
            float w2 = ROW(NF_R_W2, 0);
#pragma unroll
            for (int j = 0; j < 32; ++j) mean2[j] += ROW(NF_R_X2, j) * w2;
        }
        float var2[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) var2[j] = 0.f;
        for (int v = 0; v < V; ++v) {
            float w2 = ROW(NF_R_W2, 0);
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                float d = ROW(NF_R_X2, j) - mean2[j];
                var2[j] += w2 * (d * d);
            }
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            SMP(NF_S_GIN, j) = mean2[j];
            SMP(NF_S_GIN, 32 + j) = var2[j];
        }
        SMP(NF_S_GIN, 64) = wmean / (float)V;
        SMP(NF_S_NVAL, 0) = nval;
        SMP(NF_S_VSUM, 0) = vsum;
    }
    {
        float g1[64];
        nf_load_bias(B + nf_lin_b(NF_L_GEO0), g1);
        nf_dense_ws<64>(B + nf_lin_wt(NF_L_GEO0), 65, SMPP(NF_S_GIN), (size_t)c.S, 1.f, g1);
#pragma unroll
        for (int n = 0; n < 64; ++n) SMP(NF_S_G1, n) = nf_elu(g1[n]);
        float g[16];
        nf_load_bias(B + nf_lin_b(NF_L_GEO1), g);
        nf_dense_ws<16>(B + nf_lin_wt(NF_L_GEO1), 64, SMPP(NF_S_G1), (size_t)c.S, 1.f, g);
        float gpe[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            g[n] = nf_elu(g[n]);
            SMP(NF_S_G, n) = g[n];
            gpe[n] = g[n] + c.pos_enc[(size_t)c.s * 16 + n];
            SMP(NF_S_GPE, n) = gpe[n];
        }
        // q, k, v projections (no bias)   (:96-98)
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            float o[16];
#pragma unroll
            for (int n = 0; n < 16; ++n) o[n] = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
#pragma unroll
                for (int n = 0; n < 16; ++n) o[n] = fmaf(B[nf_att_wt(m) + k * 16 + n], gpe[k], o[n]);
            }
            const int slot = m == 0 ? NF_S_Q : (m == 1 ? NF_S_K : NF_S_V);
#pragma unroll
            for (int n = 0; n < 16; ++n) SMP(slot, n) = o[n];
        }
    }
    // ---- colour head: rgb_fc on [x2, vis2, ray_diff], softmax over views, blend of the CLEAN colours   (:268-273)
    float ymax = -3.0e38f;
    for (int v = 0; v < V; ++v) {
        float mk = c.mask[(size_t)c.s * V + v];
        const float* rd = c.ray_diff + ((size_t)c.s * V + v) * 4;
        float r1[16];
        nf_load_bias(B + nf_lin_b(NF_L_RGB0), r1);
        nf_dense_ws<16>(B + nf_lin_wt(NF_L_RGB0), 32, ROWP(NF_R_X2), ROWSTRIDE, 1.f, r1);
        {
            float vis2 = ROW(NF_R_VIS2, 0);
#pragma unroll
            for (int n = 0; n < 16; ++n) r1[n] = fmaf(B[nf_lin_wt(NF_L_RGB0) + 32 * 16 + n], vis2, r1[n]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int n = 0; n < 16; ++n) r1[n] = fmaf(B[nf_lin_wt(NF_L_RGB0) + (33 + k) * 16 + n], rd[k], r1[n]);
            }
        }
        float r2[8];
        nf_load_bias(B + nf_lin_b(NF_L_RGB1), r2);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            r1[k] = nf_elu(r1[k]);
            ROW(NF_R_R1, k) = r1[k];
#pragma unroll
            for (int n = 0; n < 8; ++n) r2[n] = fmaf(B[nf_lin_wt(NF_L_RGB1) + k * 8 + n], r1[k], r2[n]);
        }
        float y = B[nf_lin_b(NF_L_RGB2)];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            r2[k] = nf_elu(r2[k]);
            ROW(NF_R_R2, k) = r2[k];
            y = fmaf(B[nf_lin_wt(NF_L_RGB2) + k], r2[k], y);
        }
        if (mk == 0.f) y = -1e9f;
        ROW(NF_R_BETA, 0) = y;
        ymax = fmaxf(ymax, y);
    }
    float ysum = 0.f;
    for (int v = 0; v < V; ++v) {
        float e = expf(ROW(NF_R_BETA, 0) - ymax);
        ROW(NF_R_BETA, 0) = e;
        ysum += e;
    }
    rgb[0] = rgb[1] = rgb[2] = 0.f;
    for (int v = 0; v < V; ++v) {
        float beta = ROW(NF_R_BETA, 0) / ysum;
        ROW(NF_R_BETA, 0) = beta;
        const float* rf = c.rgb_feat + ((size_t)c.s * V + v) * 35;
        rgb[0] += rf[0] * beta;
        rgb[1] += rf[1] * beta;
        rgb[2] += rf[2] * beta;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// forward, ray attention for query sample c.s + LayerNorm + out_geometry_fc.  Requires a barrier after ibr_fwd_sample.
// ---------------------------------------------------------------------------------------------------------------
__device__ float ibr_fwd_attention(const IbrCtx& c) {
    const float* B = c.blob;
    const int S = c.S;
    float nval = SMP(NF_S_NVAL, 0);
    bool row_on = nval > 1.f;
    float o[16];
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
        float q0 = SMP(NF_S_Q, hd * 4 + 0) / 2.f, q1 = SMP(NF_S_Q, hd * 4 + 1) / 2.f, q2 = SMP(NF_S_Q, hd * 4 + 2) / 2.f,
              q3 = SMP(NF_S_Q, hd * 4 + 3) / 2.f;
        const float* Kb = c.ws_smp + (size_t)(NF_S_K + hd * 4) * S;
        const float* Vb = c.ws_smp + (size_t)(NF_S_V + hd * 4) * S;
        float m = -3.0e38f;
        if (row_on) {
            for (int k = 0; k < S; ++k) {
                float sc = fmaf(q3, Kb[3 * S + k], fmaf(q2, Kb[2 * S + k], fmaf(q1, Kb[S + k], q0 * Kb[k])));
                m = fmaxf(m, sc);
            }
        } else {
            m = -1e9f;   // masked_fill(mask == 0, -1e9) on the whole QUERY row => uniform attention (:36)
        }
        float l = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int k = 0; k < S; ++k) {
            float sc = row_on ? fmaf(q3, Kb[3 * S + k], fmaf(q2, Kb[2 * S + k], fmaf(q1, Kb[S + k], q0 * Kb[k]))) : -1e9f;
            float p = expf(sc - m);
            l += p;
            a0 = fmaf(p, Vb[k], a0);
            a1 = fmaf(p, Vb[S + k], a1);
            a2 = fmaf(p, Vb[2 * S + k], a2);
            a3 = fmaf(p, Vb[3 * S + k], a3);
        }
        o[hd * 4 + 0] = a0 / l; o[hd * 4 + 1] = a1 / l; o[hd * 4 + 2] = a2 / l; o[hd * 4 + 3] = a3 / l;
        SMP(NF_S_M, hd) = m;
        SMP(NF_S_L, hd) = l;
    }
    float pre[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        SMP(NF_S_O, n) = o[n];
        pre[n] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int n = 0; n < 16; ++n) pre[n] = fmaf(B[nf_att_wt(3) + k * 16 + n], o[k], pre[n]);
    }
    float mu = 0.f;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        pre[n] += SMP(NF_S_GPE, n);
        mu += pre[n];
    }
    mu = mu / 16.f;
    float var = 0.f;
#pragma unroll
    for (int n = 0; n < 16; ++n) var += (pre[n] - mu) * (pre[n] - mu);
    float rstd = 1.f / sqrtf(var / 16.f + 1e-6f);
    SMP(NF_S_RSTD, 0) = rstd;
    float gat[16], og1[16];
    nf_load_bias(B + nf_lin_b(NF_L_OG0), og1);
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        float xh = (pre[n] - mu) * rstd;
        SMP(NF_S_XHAT, n) = xh;
        gat[n] = xh * B[NF_LN_W + n] + B[NF_LN_B + n];
        SMP(NF_S_GAT, n) = gat[n];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int n = 0; n < 16; ++n) og1[n] = fmaf(B[nf_lin_wt(NF_L_OG0) + k * 16 + n], gat[k], og1[n]);
    }
    float sp = B[nf_lin_b(NF_L_OG1)];
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        og1[n] = nf_elu(og1[n]);
        SMP(NF_S_OG1, n) = og1[n];
        sp = fmaf(B[nf_lin_wt(NF_L_OG1) + n], og1[n], sp);
    }
    SMP(NF_S_SIGPRE, 0) = sp;
    float sigma = fmaxf(sp, 0.f);
    if (nval < 1.f) sigma = 0.f;   // masked_fill(num_valid_obs < 1, 0)  (:266)
    return sigma;
}

template <int MAXT>
__global__ void __launch_bounds__(MAXT) k_ibrnet_fwd(const float* __restrict__ blob, const float* __restrict__ pos_enc,
                                                             const float* __restrict__ rgb_feat,
                                                             const float* __restrict__ ray_diff,
                                                             const float* __restrict__ mask, int S, int V, int aa,
                                                             float* __restrict__ raw, float* __restrict__ ws, int row_floats,
                                                             int smp_floats) {
    int64_t ray = blockIdx.x;
    IbrCtx c;
    c.blob = blob; c.pos_enc = pos_enc; c.S = S; c.V = V; c.aa = aa; c.s = threadIdx.x;
    c.rgb_feat = rgb_feat + ray * S * V * 35;
    c.ray_diff = ray_diff + ray * S * V * 4;
    c.mask = mask + ray * S * V;
    size_t per_ray = (size_t)S * V * row_floats + (size_t)S * smp_floats;
    c.ws_row = ws + ray * per_ray;
    c.ws_smp = c.ws_row + (size_t)S * V * row_floats;
    bool active = c.s < S;
    float rgb[3] = {0.f, 0.f, 0.f};
    if (active) ibr_fwd_sample(c, rgb);
    __syncthreads();
    if (active) {
        float sigma = ibr_fwd_attention(c);
        float* o = raw + (ray * S + c.s) * 4;
        o[0] = rgb[0]; o[1] = rgb[1]; o[2] = rgb[2]; o[3] = sigma;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward (d/d rgb_feat), blueprint: oracle/ibrnet_manual_bwd.py
// ---------------------------------------------------------------------------------------------------------------

// B1: density head, LayerNorm, fc of the attention -> d_o, D_h = <d_o_h, o_h>, and the residual part of d_gpe (kept
// in GS[0..15]).  Thread = query sample.
__device__ void ibr_bwd_density_head(const IbrCtx& c, float d_sigma) {
    const float* B = c.blob;
    float nval = SMP(NF_S_NVAL, 0);
    float live = (nval >= 1.f && SMP(NF_S_SIGPRE, 0) > 0.f) ? 1.f : 0.f;
    float d_sp = d_sigma * live;
    float d_og1[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) d_og1[n] = B[nf_lin_w(NF_L_OG1) + n] * d_sp * nf_elu_grad(SMP(NF_S_OG1, n));
    float d_gat[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) d_gat[k] = 0.f;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
#pragma unroll
        for (int k = 0; k < 16; ++k) d_gat[k] = fmaf(B[nf_lin_w(NF_L_OG0) + n * 16 + k], d_og1[n], d_gat[k]);
    }
    float m1 = 0.f, m2 = 0.f, d_xh[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        d_xh[n] = d_gat[n] * B[NF_LN_W + n];
        m1 += d_xh[n];
        m2 += d_xh[n] * SMP(NF_S_XHAT, n);
    }
    m1 = m1 / 16.f;
    m2 = m2 / 16.f;
    float rstd = SMP(NF_S_RSTD, 0);
    float d_pre[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        d_pre[n] = rstd * (d_xh[n] - m1 - SMP(NF_S_XHAT, n) * m2);
        SMP(NF_S_GS, n) = d_pre[n];   // residual branch of d_gpe
    }
    float d_o[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) d_o[j] = 0.f;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
#pragma unroll
        for (int j = 0; j < 16; ++j) d_o[j] = fmaf(B[nf_att_w(3) + n * 16 + j], d_pre[n], d_o[j]);
    }
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
        float D = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            SMP(NF_S_DO, hd * 4 + d) = d_o[hd * 4 + d];
            D = fmaf(d_o[hd * 4 + d], SMP(NF_S_O, hd * 4 + d), D);
        }
        SMP(NF_S_DH, hd) = D;
    }
}

// B2: thread = query: dQ.  B3: thread = key: dK, dV.  Both re-derive p[q,k] from the stored (max, sum) of the row.
__device__ void ibr_bwd_attention_q(const IbrCtx& c) {
    const int S = c.S;
    bool row_on = SMP(NF_S_NVAL, 0) > 1.f;
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
        float dq0 = 0.f, dq1 = 0.f, dq2 = 0.f, dq3 = 0.f;
        if (row_on) {
            float q0 = SMP(NF_S_Q, hd * 4 + 0) / 2.f, q1 = SMP(NF_S_Q, hd * 4 + 1) / 2.f, q2 = SMP(NF_S_Q, hd * 4 + 2) / 2.f,
                  q3 = SMP(NF_S_Q, hd * 4 + 3) / 2.f;
            float g0 = SMP(NF_S_DO, hd * 4 + 0), g1 = SMP(NF_S_DO, hd * 4 + 1), g2 = SMP(NF_S_DO, hd * 4 + 2),
                  g3 = SMP(NF_S_DO, hd * 4 + 3);
            float m = SMP(NF_S_M, hd), l = SMP(NF_S_L, hd), D = SMP(NF_S_DH, hd);
            const float* Kb = c.ws_smp + (size_t)(NF_S_K + hd * 4) * S;
            const float* Vb = c.ws_smp + (size_t)(NF_S_V + hd * 4) * S;
            for (int k = 0; k < S; ++k) {
                float k0 = Kb[k], k1 = Kb[S + k], k2 = Kb[2 * S + k], k3 = Kb[3 * S + k];
                float sc = fmaf(q3, k3, fmaf(q2, k2, fmaf(q1, k1, q0 * k0)));
                float p = expf(sc - m) / l;
                float dA = fmaf(g3, Vb[3 * S + k], fmaf(g2, Vb[2 * S + k], fmaf(g1, Vb[S + k], g0 * Vb[k])));
                float dS = p * (dA - D);
                dq0 = fmaf(dS, k0, dq0); dq1 = fmaf(dS, k1, dq1); dq2 = fmaf(dS, k2, dq2); dq3 = fmaf(dS, k3, dq3);
            }
        }
        SMP(NF_S_DQ, hd * 4 + 0) = dq0 / 2.f;
        SMP(NF_S_DQ, hd * 4 + 1) = dq1 / 2.f;
        SMP(NF_S_DQ, hd * 4 + 2) = dq2 / 2.f;
        SMP(NF_S_DQ, hd * 4 + 3) = dq3 / 2.f;
    }
}

__device__ void ibr_bwd_attention_kv(const IbrCtx& c, float (&d_gpe)[16]) {
    const float* B = c.blob;
    const int S = c.S;
    float dk[16], dv[16];
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
        float k0 = SMP(NF_S_K, hd * 4 + 0), k1 = SMP(NF_S_K, hd * 4 + 1), k2 = SMP(NF_S_K, hd * 4 + 2), k3 = SMP(NF_S_K, hd * 4 + 3);
        float v0 = SMP(NF_S_V, hd * 4 + 0), v1 = SMP(NF_S_V, hd * 4 + 1), v2 = SMP(NF_S_V, hd * 4 + 2), v3 = SMP(NF_S_V, hd * 4 + 3);
        const float* Qb = c.ws_smp + (size_t)(NF_S_Q + hd * 4) * S;
        const float* Gb = c.ws_smp + (size_t)(NF_S_DO + hd * 4) * S;
        const float* Mb = c.ws_smp + (size_t)(NF_S_M + hd) * S;
        const float* Lb = c.ws_smp + (size_t)(NF_S_L + hd) * S;
        const float* Db = c.ws_smp + (size_t)(NF_S_DH + hd) * S;
        const float* Nb = c.ws_smp + (size_t)NF_S_NVAL * S;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        for (int q = 0; q < S; ++q) {
            bool on = Nb[q] > 1.f;
            float q0 = Qb[q] / 2.f, q1 = Qb[S + q] / 2.f, q2 = Qb[2 * S + q] / 2.f, q3 = Qb[3 * S + q] / 2.f;
            float g0 = Gb[q], g1 = Gb[S + q], g2 = Gb[2 * S + q], g3 = Gb[3 * S + q];
            float sc = on ? fmaf(q3, k3, fmaf(q2, k2, fmaf(q1, k1, q0 * k0))) : -1e9f;
            float p = expf(sc - Mb[q]) / Lb[q];
            b0 = fmaf(p, g0, b0); b1 = fmaf(p, g1, b1); b2 = fmaf(p, g2, b2); b3 = fmaf(p, g3, b3);
            if (on) {
                float dA = fmaf(g3, v3, fmaf(g2, v2, fmaf(g1, v1, g0 * v0)));
                float dS = p * (dA - Db[q]);
                a0 = fmaf(dS, q0, a0); a1 = fmaf(dS, q1, a1); a2 = fmaf(dS, q2, a2); a3 = fmaf(dS, q3, a3);
            }
        }
        dk[hd * 4 + 0] = a0; dk[hd * 4 + 1] = a1; dk[hd * 4 + 2] = a2; dk[hd * 4 + 3] = a3;
        dv[hd * 4 + 0] = b0; dv[hd * 4 + 1] = b1; dv[hd * 4 + 2] = b2; dv[hd * 4 + 3] = b3;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) d_gpe[j] = SMP(NF_S_GS, j);
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        float dqn = SMP(NF_S_DQ, n);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            d_gpe[j] = fmaf(B[nf_att_w(0) + n * 16 + j], dqn, d_gpe[j]);
            d_gpe[j] = fmaf(B[nf_att_w(1) + n * 16 + j], dk[n], d_gpe[j]);
            d_gpe[j] = fmaf(B[nf_att_w(2) + n * 16 + j], dv[n], d_gpe[j]);
        }
    }
}

// B4: everything per sample after the attention: geometry_fc, second pooling, vis_fc2, vis_fc, base_fc, first pooling,
// colour head.  Writes d_rgb_feat[s, :, 0:35].
__device__ void ibr_bwd_sample(const IbrCtx& c, const float (&d_gpe)[16], const float (&d_rgb)[3], float* __restrict__ d_rgb_feat) {
    const float* B = c.blob;
    const int V = c.V;
    // ---- geometry_fc backward: d_gin[65]
    {
#pragma unroll
        for (int n = 0; n < 16; ++n) SMP(NF_S_GS, n) = d_gpe[n] * nf_elu_grad(SMP(NF_S_G, n));
        float d_g1[64];
#pragma unroll
        for (int k = 0; k < 64; ++k) d_g1[k] = 0.f;
        nf_dense_bwd_ws<64>(B + nf_lin_w(NF_L_GEO1), 16, 64, SMPP(NF_S_GS), (size_t)c.S, d_g1);
#pragma unroll
        for (int k = 0; k < 64; ++k) SMP(NF_S_GS, k) = d_g1[k] * nf_elu_grad(SMP(NF_S_G1, k));
        float d_gin[65];
#pragma unroll
        for (int k = 0; k < 65; ++k) d_gin[k] = 0.f;
        nf_dense_bwd_ws<65>(B + nf_lin_w(NF_L_GEO0), 64, 65, SMPP(NF_S_GS), (size_t)c.S, d_gin);
#pragma unroll
        for (int k = 0; k < 65; ++k) SMP(NF_S_DGIN, k) = d_gin[k];
    }
    const float vsum = SMP(NF_S_VSUM, 0);
    const float d_wmean = SMP(NF_S_DGIN, 64);
    // ---- colour head backward per view -> DX2, DVIS2
    float sbd = 0.f;
    for (int v = 0; v < V; ++v) {
        const float* rf = c.rgb_feat + ((size_t)c.s * V + v) * 35;
        float d_beta = rf[0] * d_rgb[0] + rf[1] * d_rgb[1] + rf[2] * d_rgb[2];
        sbd = fmaf(ROW(NF_R_BETA, 0), d_beta, sbd);
    }
    for (int v = 0; v < V; ++v) {
        const float* rf = c.rgb_feat + ((size_t)c.s * V + v) * 35;
        float mk = c.mask[(size_t)c.s * V + v];
        float d_beta = rf[0] * d_rgb[0] + rf[1] * d_rgb[1] + rf[2] * d_rgb[2];
        float d_y = ROW(NF_R_BETA, 0) * (d_beta - sbd);
        if (mk == 0.f) d_y = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) ROW(NF_R_GR, k) = B[nf_lin_w(NF_L_RGB2) + k] * d_y * nf_elu_grad(ROW(NF_R_R2, k));
        float d_r1[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) d_r1[k] = 0.f;
        nf_dense_bwd_ws<16>(B + nf_lin_w(NF_L_RGB1), 8, 16, ROWP(NF_R_GR), ROWSTRIDE, d_r1);
#pragma unroll
        for (int k = 0; k < 16; ++k) ROW(NF_R_GR, k) = d_r1[k] * nf_elu_grad(ROW(NF_R_R1, k));
        float d_yin[33];
#pragma unroll
        for (int k = 0; k < 33; ++k) d_yin[k] = 0.f;
        nf_dense_bwd_ws<33>(B + nf_lin_w(NF_L_RGB0), 16, 37, ROWP(NF_R_GR), ROWSTRIDE, d_yin);
#pragma unroll
        for (int k = 0; k < 32; ++k) ROW(NF_R_DX2, k) = d_yin[k];
        ROW(NF_R_DVIS2, 0) = d_yin[32];
    }
    // ---- second pooling: d_mean2_tot, d_w2
    float dm2[32], dv2[32];
    {
        float s2[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) s2[j] = 0.f;
        for (int v = 0; v < V; ++v) {
            float w2 = ROW(NF_R_W2, 0);
#pragma unroll
            for (int j = 0; j < 32; ++j) s2[j] = fmaf(w2, ROW(NF_R_X2, j) - SMP(NF_S_GIN, j), s2[j]);
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            dv2[j] = SMP(NF_S_DGIN, 32 + j);
            dm2[j] = SMP(NF_S_DGIN, j) + dv2[j] * (-2.f * s2[j]);
        }
    }
    float sdw = 0.f;
    for (int v = 0; v < V; ++v) {
        float acc = d_wmean / (float)V;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            float x2 = ROW(NF_R_X2, j);
            float dev = x2 - SMP(NF_S_GIN, j);
            acc = fmaf(x2, dm2[j], acc);
            acc = fmaf(dev * dev, dv2[j], acc);
        }
        ROW(NF_R_DW2, 0) = acc;
        sdw = fmaf(acc, ROW(NF_R_W2, 0), sdw);
    }
    // ---- per view: vis_fc2, vis_fc, base_fc backward; accumulate d_mean / d_var of the first pooling
    float d_mean[35], d_var[35];
#pragma unroll
    for (int j = 0; j < 35; ++j) d_mean[j] = d_var[j] = 0.f;
    for (int v = 0; v < V; ++v) {
        float mk = c.mask[(size_t)c.s * V + v];
        float w = ROW(NF_R_W, 0);
        float w2 = ROW(NF_R_W2, 0);
        float d_x2[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            float dev = ROW(NF_R_X2, j) - SMP(NF_S_GIN, j);
            d_x2[j] = ROW(NF_R_DX2, j) + w2 * (dm2[j] + 2.f * dev * dv2[j]);
        }
        float d_vis2 = ROW(NF_R_DVIS2, 0) + (ROW(NF_R_DW2, 0) - sdw) / vsum;
        float sig2 = ROW(NF_R_SIG2, 0);
        float d_z2 = d_vis2 * mk * sig2 * (1.f - sig2);
#pragma unroll
        for (int k = 0; k < 32; ++k) ROW(NF_R_GR, k) = B[nf_lin_w(NF_L_VISB1) + k] * d_z2 * nf_elu_grad(ROW(NF_R_U, k));
        float d_xvis[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) d_xvis[k] = 0.f;
        nf_dense_bwd_ws<32>(B + nf_lin_w(NF_L_VISB0), 32, 32, ROWP(NF_R_GR), ROWSTRIDE, d_xvis);
        float vis1 = ROW(NF_R_VIS1, 0);
        float d_vis1 = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            d_x2[k] = fmaf(d_xvis[k], vis1, d_x2[k]);
            d_vis1 = fmaf(d_xvis[k], ROW(NF_R_X2, k), d_vis1);
        }
        float sig1 = ROW(NF_R_SIG1, 0);
#pragma unroll
        for (int k = 0; k < 32; ++k) ROW(NF_R_GR, k) = d_x2[k] * nf_elu_grad(ROW(NF_R_XV, k));
        ROW(NF_R_GR, 32) = d_vis1 * mk * sig1 * (1.f - sig1) * nf_elu_grad(ROW(NF_R_XV, 32));
        float d_v1[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) d_v1[k] = 0.f;
        nf_dense_bwd_ws<32>(B + nf_lin_w(NF_L_VIS1), 33, 32, ROWP(NF_R_GR), ROWSTRIDE, d_v1);
#pragma unroll
        for (int k = 0; k < 32; ++k) ROW(NF_R_GR, k) = d_v1[k] * nf_elu_grad(ROW(NF_R_V1, k));
        float d_t[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) d_t[k] = 0.f;
        nf_dense_bwd_ws<32>(B + nf_lin_w(NF_L_VIS0), 32, 32, ROWP(NF_R_GR), ROWSTRIDE, d_t);
#pragma unroll
        for (int k = 0; k < 32; ++k) ROW(NF_R_GR, k) = (d_x2[k] + d_t[k] * w) * nf_elu_grad(ROW(NF_R_H, k));
        float d_h1[64];
#pragma unroll
        for (int k = 0; k < 64; ++k) d_h1[k] = 0.f;
        nf_dense_bwd_ws<64>(B + nf_lin_w(NF_L_BASE1), 32, 64, ROWP(NF_R_GR), ROWSTRIDE, d_h1);
#pragma unroll
        for (int k = 0; k < 64; ++k) ROW(NF_R_GR, k) = d_h1[k] * nf_elu_grad(ROW(NF_R_H1, k));
        nf_dense_bwd_ws<35>(B + nf_lin_w(NF_L_BASE0), 64, 105, ROWP(NF_R_GR), ROWSTRIDE, d_mean);
        nf_dense_bwd_ws<35>(B + nf_lin_w(NF_L_BASE0) + 35, 64, 105, ROWP(NF_R_GR), ROWSTRIDE, d_var);
        float d_f[35];
#pragma unroll
        for (int k = 0; k < 35; ++k) d_f[k] = 0.f;
        nf_dense_bwd_ws<35>(B + nf_lin_w(NF_L_BASE0) + 70, 64, 105, ROWP(NF_R_GR), ROWSTRIDE, d_f);
#pragma unroll
        for (int k = 0; k < 35; ++k) ROW(NF_R_DF, k) = d_f[k];
    }
    // ---- first pooling (weights constant): d_f += w (d_mean_tot + 2 dev d_var); colour taps: + beta * d_rgb
    {
        float s1[35];
#pragma unroll
        for (int j = 0; j < 35; ++j) s1[j] = 0.f;
        for (int v = 0; v < V; ++v) {
            float w = ROW(NF_R_W, 0);
#pragma unroll
            for (int j = 0; j < 35; ++j) s1[j] = fmaf(w, ROW(NF_R_F, j) - SMP(NF_S_MEAN, j), s1[j]);
        }
#pragma unroll
        for (int j = 0; j < 35; ++j) d_mean[j] = d_mean[j] + d_var[j] * (-2.f * s1[j]);
    }
    for (int v = 0; v < V; ++v) {
        float w = ROW(NF_R_W, 0);
        float beta = ROW(NF_R_BETA, 0);
        float* out = d_rgb_feat + ((size_t)c.s * V + v) * 35;
#pragma unroll
        for (int j = 0; j < 35; ++j) {
            float dev = ROW(NF_R_F, j) - SMP(NF_S_MEAN, j);
            float g = ROW(NF_R_DF, j) + w * (d_mean[j] + 2.f * dev * d_var[j]);
            if (j < 3) g = fmaf(beta, d_rgb[j], g);
            out[j] = g;
        }
    }
}

template <int MAXT>
__global__ void __launch_bounds__(MAXT) k_ibrnet_bwd(const float* __restrict__ blob, const float* __restrict__ pos_enc,
                                                             const float* __restrict__ rgb_feat,
                                                             const float* __restrict__ ray_diff,
                                                             const float* __restrict__ mask, const float* __restrict__ d_raw,
                                                             int S, int V, int aa, float* __restrict__ d_rgb_feat,
                                                             float* __restrict__ ws) {
    int64_t ray = blockIdx.x;
    IbrCtx c;
    c.blob = blob; c.pos_enc = pos_enc; c.S = S; c.V = V; c.aa = aa; c.s = threadIdx.x;
    c.rgb_feat = rgb_feat + ray * S * V * 35;
    c.ray_diff = ray_diff + ray * S * V * 4;
    c.mask = mask + ray * S * V;
    size_t per_ray = (size_t)S * V * NF_ROW_BWD + (size_t)S * NF_SMP_BWD;
    c.ws_row = ws + ray * per_ray;
    c.ws_smp = c.ws_row + (size_t)S * V * NF_ROW_BWD;
    bool active = c.s < S;
    float rgb[3];
    if (active) ibr_fwd_sample(c, rgb);
    __syncthreads();
    float d_rgb[3] = {0.f, 0.f, 0.f};
    if (active) {
        (void)ibr_fwd_attention(c);
        const float* g = d_raw + (ray * S + c.s) * 4;
        d_rgb[0] = g[0]; d_rgb[1] = g[1]; d_rgb[2] = g[2];
        ibr_bwd_density_head(c, g[3]);
    }
    __syncthreads();
    if (active) ibr_bwd_attention_q(c);
    __syncthreads();
    if (active) {
        float d_gpe[16];
        ibr_bwd_attention_kv(c, d_gpe);
        ibr_bwd_sample(c, d_gpe, d_rgb, d_rgb_feat + ray * S * V * 35);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
extern "C" int64_t nf_ibrnet_blob_floats(void) { return NF_BLOB_FLOATS; }

extern "C" int nf_ibrnet_blob_entry(int idx, char* name, int name_cap, int64_t* offset, int* rows, int* cols, int* transposed) {
    if (idx < 0 || idx >= NF_BLOB_ENTRIES) return 1;
    char key[64];
    int64_t off = 0;
    int r = 1, cdim = 1, tr = 0;
    static const char* att[4] = {"ray_attention.w_qs.weight", "ray_attention.w_ks.weight", "ray_attention.w_vs.weight",
                                 "ray_attention.fc.weight"};
    if (idx == 0) {
        snprintf(key, sizeof(key), "s");
    } else if (idx <= NF_N_LIN * 3) {
        int l = (idx - 1) / 3, part = (idx - 1) % 3;
        if (part == 2) {
            snprintf(key, sizeof(key), "%s.bias", NF_LIN[l].key);
            off = nf_lin_b(l); r = 1; cdim = NF_LIN[l].out;
        } else {
            snprintf(key, sizeof(key), "%s.weight", NF_LIN[l].key);
            off = part == 0 ? nf_lin_wt(l) : nf_lin_w(l);
            r = NF_LIN[l].out; cdim = NF_LIN[l].in; tr = part == 0 ? 1 : 0;
        }
    } else if (idx <= NF_N_LIN * 3 + 8) {
        int j = idx - 1 - NF_N_LIN * 3;
        snprintf(key, sizeof(key), "%s", att[j / 2]);
        off = (j % 2 == 0) ? nf_att_wt(j / 2) : nf_att_w(j / 2);
        r = 16; cdim = 16; tr = (j % 2 == 0) ? 1 : 0;
    } else {
        int j = idx - 1 - NF_N_LIN * 3 - 8;
        snprintf(key, sizeof(key), j == 0 ? "ray_attention.layer_norm.weight" : "ray_attention.layer_norm.bias");
        off = j == 0 ? NF_LN_W : NF_LN_B; r = 1; cdim = 16;
    }
    if (name && name_cap > 0) {
        strncpy(name, key, (size_t)name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (offset) *offset = off;
    if (rows) *rows = r;
    if (cols) *cols = cdim;
    if (transposed) *transposed = tr;
    return 0;
}

static int64_t nf_ibr_per_ray(int S, int V, int backward) {
    return backward ? ((int64_t)S * V * NF_ROW_BWD + (int64_t)S * NF_SMP_BWD)
                    : ((int64_t)S * V * NF_ROW_FWD + (int64_t)S * NF_SMP_FWD);
}

extern "C" int64_t nf_ibrnet_workspace_floats(int64_t n_rays, int n_samples, int n_views, int backward) {
    int64_t rays = n_rays < NF_IBR_RAYS_PER_LAUNCH ? n_rays : NF_IBR_RAYS_PER_LAUNCH;
    if (rays < 1) rays = 1;
    return rays * nf_ibr_per_ray(n_samples, n_views, backward);
}

static int nf_ibr_check(const char* who, int64_t n_rays, int S, int V) {
    NF_REQUIRE(n_rays >= 0 && S >= 1 && S <= NF_IBR_MAX_S && V >= 1 && V <= 64,
               "%s: need 1 <= S <= %d and 1 <= V <= 64 (got R %lld, S %d, V %d)", who, NF_IBR_MAX_S, (long long)n_rays, S, V);
    return 0;
}

extern "C" int nf_ibrnet_fwd(const float* blob, const float* pos_enc, const float* rgb_feat, const float* ray_diff,
                             const float* mask, int64_t n_rays, int n_samples, int n_views, int anti_alias_pooling,
                             float* raw, float* workspace, nf_stream_t stream) {
    if (nf_ibr_check("nf_ibrnet_fwd", n_rays, n_samples, n_views)) return 1;
    const int S = n_samples, V = n_views;
    int threads = ((S + 63) / 64) * 64;
    for (int64_t r0 = 0; r0 < n_rays; r0 += NF_IBR_RAYS_PER_LAUNCH) {
        int64_t nr = n_rays - r0 < NF_IBR_RAYS_PER_LAUNCH ? n_rays - r0 : NF_IBR_RAYS_PER_LAUNCH;
        if (threads <= 256)
            hipLaunchKernelGGL(k_ibrnet_fwd<256>, dim3((unsigned)nr), dim3(threads), 0, (hipStream_t)stream, blob, pos_enc,
                               rgb_feat + r0 * S * V * 35, ray_diff + r0 * S * V * 4, mask + r0 * S * V, S, V,
                               anti_alias_pooling, raw + r0 * S * 4, workspace, (int)NF_ROW_FWD, (int)NF_SMP_FWD);
        else
            hipLaunchKernelGGL(k_ibrnet_fwd<NF_IBR_MAX_S>, dim3((unsigned)nr), dim3(threads), 0, (hipStream_t)stream, blob,
                               pos_enc, rgb_feat + r0 * S * V * 35, ray_diff + r0 * S * V * 4, mask + r0 * S * V, S, V,
                               anti_alias_pooling, raw + r0 * S * 4, workspace, (int)NF_ROW_FWD, (int)NF_SMP_FWD);
        NF_LAUNCH_CHECK("nf_ibrnet_fwd");
    }
    return 0;
}

extern "C" int nf_ibrnet_bwd(const float* blob, const float* pos_enc, const float* rgb_feat, const float* ray_diff,
                             const float* mask, const float* d_raw, int64_t n_rays, int n_samples, int n_views,
                             int anti_alias_pooling, float* d_rgb_feat, float* workspace, nf_stream_t stream) {
    if (nf_ibr_check("nf_ibrnet_bwd", n_rays, n_samples, n_views)) return 1;
    const int S = n_samples, V = n_views;
    int threads = ((S + 63) / 64) * 64;
    for (int64_t r0 = 0; r0 < n_rays; r0 += NF_IBR_RAYS_PER_LAUNCH) {
        int64_t nr = n_rays - r0 < NF_IBR_RAYS_PER_LAUNCH ? n_rays - r0 : NF_IBR_RAYS_PER_LAUNCH;
        if (threads <= 256)
            hipLaunchKernelGGL(k_ibrnet_bwd<256>, dim3((unsigned)nr), dim3(threads), 0, (hipStream_t)stream, blob, pos_enc,
                               rgb_feat + r0 * S * V * 35, ray_diff + r0 * S * V * 4, mask + r0 * S * V, d_raw + r0 * S * 4,
                               S, V, anti_alias_pooling, d_rgb_feat + r0 * S * V * 35, workspace);
        else
            hipLaunchKernelGGL(k_ibrnet_bwd<NF_IBR_MAX_S>, dim3((unsigned)nr), dim3(threads), 0, (hipStream_t)stream, blob,
                               pos_enc, rgb_feat + r0 * S * V * 35, ray_diff + r0 * S * V * 4, mask + r0 * S * V,
                               d_raw + r0 * S * 4, S, V, anti_alias_pooling, d_rgb_feat + r0 * S * V * 35, workspace);
        NF_LAUNCH_CHECK("nf_ibrnet_bwd");
    }
    return 0;
}
