// GNT per-ray network (a15): view transformers (subtraction attention over the V source views) interleaved with ray
// transformers (4-head attention over the S samples), forward and backward w.r.t. rgb_feat.
// ref: gnt/transformer_network.py:270-309 (GNT.forward, ret_alpha = False, eval mode), :55-89, :93-113, :121-171, :175-202.
// The backward follows oracle/gnt_manual_bwd.py.
//
// Shape-generic kernels (any V, S <= 256, any depth): one workgroup per ray, FOUR lanes per sample -- each lane owns a
// 16-channel slice of every 64-wide vector and one of the four attention heads -- and walks the V views serially.
// Vectors are exchanged through a per-ray global workspace [slot][view][sample] / [slot][sample] (sample fastest =>
// coalesced; L2-resident), so every stage ends with a workgroup barrier; weights are read with wave-uniform addresses
// (scalar loads).  With `save` the forward keeps every activation the backward needs (one slot set per layer), otherwise
// one slot set is recycled.
#include "nf_dense.h"

#include <stdio.h>
#include <string.h>

#include "nf_gnt.h"

extern "C" int64_t nf_gnt_blob_floats(int depth) { return GNT_STEM_FLOATS + (int64_t)depth * GNT_LAYER_FLOATS + GNT_FINAL_FLOATS; }

// idx enumerates (state-dict key, offset, rows, cols, transposed); returns 1 past the end.  Entries of q_fcs on odd layers
// are reported with an empty name (nn.Identity in the reference: nothing to pack).
extern "C" int nf_gnt_blob_entry(int depth, int idx, char* name, int name_cap, int64_t* offset, int* rows, int* cols,
                                 int* transposed) {
    char key[96] = "";
    int64_t off = 0;
    int r = 1, c = 1, tr = 0;
    const int per_lin = 3, stem_n = 2 * per_lin, layer_n = G_NLIN * per_lin + 8;
    if (idx < 0) return 1;
    if (idx < stem_n) {
        const GntLin& l = GNT_STEM[idx / 3];
        int part = idx % 3;
        int64_t base = idx / 3 == 0 ? 0 : GNT_STEM1;
        snprintf(key, sizeof(key), "%s.%s", l.key, part == 2 ? "bias" : "weight");
        if (part == 2) { off = base + 2 * l.in * l.out; r = 1; c = l.out; }
        else { off = base + (part == 0 ? 0 : l.in * l.out); r = l.out; c = l.in; tr = part == 0; }
    } else if (idx < stem_n + depth * layer_n) {
        int li = (idx - stem_n) / layer_n, j = (idx - stem_n) % layer_n;
        int64_t base = gnt_layer_base(li);
        if (j < G_NLIN * per_lin) {
            int l = j / 3, part = j % 3;
            const GntLin& L = GNT_LAYER[l];
            char fmt[96];
            snprintf(fmt, sizeof(fmt), "%s.%s", L.key, part == 2 ? "bias" : "weight");
            bool skip = (part == 2 && !L.bias) || ((l == GQ_0 || l == GQ_2) && (li % 2 == 1));
            if (!skip) snprintf(key, sizeof(key), fmt, li);
            if (part == 2) { off = base + gnt_b(l); r = 1; c = L.out; }
            else { off = base + (part == 0 ? gnt_wt(l) : gnt_w(l)); r = L.out; c = L.in; tr = part == 0; }
        } else {
            int k = j - G_NLIN * per_lin;      // 0..7: 4 LayerNorms x (weight, bias)
            char fmt[96];
            snprintf(fmt, sizeof(fmt), "%s.%s", GNT_LN_KEYS[k / 2], k % 2 == 0 ? "weight" : "bias");
            snprintf(key, sizeof(key), fmt, li);
            off = base + (k % 2 == 0 ? gnt_ln_w(k / 2) : gnt_ln_b(k / 2)); r = 1; c = 64;
        }
    } else {
        int j = idx - stem_n - depth * layer_n;
        int64_t base = gnt_layer_base(depth);
        if (j == 0) { snprintf(key, sizeof(key), "norm.weight"); off = base; c = 64; }
        else if (j == 1) { snprintf(key, sizeof(key), "norm.bias"); off = base + 64; c = 64; }
        else if (j == 2) { snprintf(key, sizeof(key), "rgb_fc.weight"); off = base + 128; r = 3; c = 64; tr = 1; }
        else if (j == 3) { snprintf(key, sizeof(key), "rgb_fc.weight"); off = base + 320; r = 3; c = 64; }
        else if (j == 4) { snprintf(key, sizeof(key), "rgb_fc.bias"); off = base + 512; c = 3; }
        else return 1;
    }
    if (name && name_cap > 0) { strncpy(name, key, (size_t)name_cap - 1); name[name_cap - 1] = 0; }
    if (offset) *offset = off;
    if (rows) *rows = r;
    if (cols) *cols = c;
    if (transposed) *transposed = tr;
    return 0;
}


extern "C" int64_t nf_gnt_workspace_floats(int64_t n_rays, int n_samples, int n_views, int depth, int save) {
    int64_t rays = save ? n_rays : (n_rays < GNT_RAYS_PER_LAUNCH ? n_rays : GNT_RAYS_PER_LAUNCH);
    if (rays < 1) rays = 1;
    return rays * ((int64_t)n_samples * n_views * gnt_row_floats(depth, save) + (int64_t)n_samples * gnt_smp_floats(depth, save));
}

struct GntCtx {
    const float* __restrict__ blob;
    const float* __restrict__ rgb_feat;   // this ray [S,V,35]
    const float* __restrict__ ray_diff;   // [S,V,4]
    const float* __restrict__ mask;       // [S,V]
    float* ws_row;
    float* ws_smp;
    int S, V, s, part, P0;                // sample, lane-part (0..3) and its 16-channel slice offset
};

// Four lanes cooperate on one sample: each owns a 16-channel slice of every 64-wide vector (and one attention head).
// Vectors are exchanged through the workspace, so every stage ends with a workgroup barrier; all threads run every stage.
#define GNT_PARTS 4
#define STAGE_BEGIN if (act) {
#define STAGE_END } __syncthreads();

// per-(sample, view) slots: [view][slot][sample] -- all slots of one view within one base + compile-time offsets (nf_gnt.h)
#define ROWP(slot) (c.ws_row + ((size_t)v * row_floats + (slot)) * c.S + c.s)
#define ROWSTRIDE ((size_t)c.S)
#define ROW(slot, j) ROWP(slot)[(size_t)(j) * ROWSTRIDE]
#define SMPP(slot) (c.ws_smp + (size_t)(slot) * c.S + c.s)
#define SMP(slot, j) SMPP(slot)[(size_t)(j) * c.S]

// y[j] (j < 16) = bias[P0 + j] + sum_k Wt[k][col0 + j] x[k]   (Wt has leading dimension ld)
__device__ __forceinline__ void gnt_slice(const float* __restrict__ Wt, int ld, int col0, const float* __restrict__ bias, int K,
                                          const float* x, size_t xstride, float (&y)[16]) {
#pragma unroll
    for (int j = 0; j < 16; ++j) y[j] = bias ? bias[col0 + j] : 0.f;
    nf_dense_ws_ld<16>(Wt + col0, ld, K, x, xstride, y);
}

// dx[j] (j < 16) += sum_n W[n][col0 + j] dy[n]   (W native [N][ldw])
__device__ __forceinline__ void gnt_slice_bwd(const float* __restrict__ W, int N, int ldw, int col0, const float* dy, size_t dystride,
                                              float (&dx)[16]) {
    nf_dense_bwd_ws<16>(W + col0, N, ldw, dy, dystride, dx);
}

// LayerNorm over the 64 channels at SMPP(src): every part recomputes the statistics, writes ITS slice of xhat and of the
// affine output (dst); part 0 stores rstd.  Needs a barrier before (src complete) and after.
__device__ __forceinline__ void gnt_ln_slice(const GntCtx& c, int src, const float* __restrict__ w, const float* __restrict__ b,
                                             float eps, int xh_slot, int rstd_slot, int dst) {
    float mu = 0.f;
    for (int j = 0; j < 64; ++j) mu += SMP(src, j);
    mu = mu / 64.f;
    float var = 0.f;
    for (int j = 0; j < 64; ++j) {
        float d = SMP(src, j) - mu;
        var += d * d;
    }
    float rstd = 1.f / sqrtf(var / 64.f + eps);
    if (c.part == 0) SMP(rstd_slot, 0) = rstd;
    for (int j = c.P0; j < c.P0 + 16; ++j) {
        float xh = (SMP(src, j) - mu) * rstd;
        SMP(xh_slot, j) = xh;
        SMP(dst, j) = xh * w[j] + b[j];
    }
}

// dst slice += rstd (dxh - mean(dxh) - xh mean(dxh xh)), dxh = dy w; dy at SMPP(dy_slot) must be complete (barrier before)
__device__ __forceinline__ void gnt_ln_bwd_slice(const GntCtx& c, int dy_slot, const float* __restrict__ w, int xh_slot,
                                                 int rstd_slot, int dst) {
    float m1 = 0.f, m2 = 0.f;
    for (int j = 0; j < 64; ++j) {
        float dxh = SMP(dy_slot, j) * w[j];
        m1 += dxh;
        m2 += dxh * SMP(xh_slot, j);
    }
    m1 = m1 / 64.f;
    m2 = m2 / 64.f;
    float rstd = SMP(rstd_slot, 0);
    for (int j = c.P0; j < c.P0 + 16; ++j) {
        float dxh = SMP(dy_slot, j) * w[j];
        SMP(dst, j) += rstd * (dxh - m1 - SMP(xh_slot, j) * m2);
    }
}

__device__ __forceinline__ float gnt_score(const float* __restrict__ base, size_t S, int k, const float (&q)[16]) {
    float sc = 0.f;
#pragma unroll
    for (int d = 0; d < 16; ++d) sc = fmaf(q[d], base[(size_t)d * S + k], sc);
    return sc;
}

#define GNT_SETUP()                                                                                   \
    const int64_t ray = blockIdx.x;                                                                   \
    const int Sp = blockDim.x / GNT_PARTS;                                                            \
    GntCtx c;                                                                                         \
    c.blob = blob; c.S = S; c.V = V; c.part = threadIdx.x / Sp; c.s = threadIdx.x - c.part * Sp; c.P0 = c.part * 16; \
    c.ray_diff = ray_diff + ray * S * V * 4;                                                          \
    c.mask = mask + ray * S * V;                                                                      \
    const size_t per_ray = (size_t)S * V * row_floats + (size_t)S * smp_floats;                      \
    c.ws_row = ws + ray * per_ray;                                                                    \
    c.ws_smp = c.ws_row + (size_t)S * V * row_floats;                                                \
    const bool act = c.s < S;                                                                         \
    const int P0 = c.P0;

// ---- shared stages -------------------------------------------------------------------------------------------------
// feed-forward 64 -> 256 -> 64 with residual: y at SW_T, hidden saved at f_slot, CUR += FF(y).  Two barriers inside.
// (training mode: the hidden units and the output pass through Dropout sites `site` and `site + 1`, transformer_network.py:47-48; the
//  hidden units are SAVED dropped -- they are what fc2 multiplies)
#define GNT_FF(L, l1, l2, f_slot, site)                                                                        \
    STAGE_BEGIN                                                                                                \
    for (int ch = 0; ch < 4; ++ch) {                                                                           \
        float f[16];                                                                                           \
        gnt_slice((L) + gnt_wt(l1), 256, ch * 64 + P0, (L) + gnt_b(l1), 64, SMPP(SW_T), (size_t)c.S, f);       \
        for (int j = 0; j < 16; ++j)                                                                           \
            SMP(f_slot, ch * 64 + P0 + j) = fmaxf(f[j], 0.f) * gnt_keep(dr, (site), smp_idx * 256 + ch * 64 + P0 + j); \
    }                                                                                                          \
    STAGE_END                                                                                                  \
    STAGE_BEGIN                                                                                                \
    float o[16];                                                                                               \
    gnt_slice((L) + gnt_wt(l2), 64, P0, (L) + gnt_b(l2), 256, SMPP(f_slot), (size_t)c.S, o);                   \
    for (int j = 0; j < 16; ++j) SMP(SW_CUR, P0 + j) += o[j] * gnt_keep(dr, (site) + 1, smp_idx * 64 + P0 + j); \
    STAGE_END

// backward of FF + its LayerNorm: DCUR holds d(q2) on entry, d(q1) on exit
// (training mode: the output's Dropout mask scales the incoming gradient -- a masked copy at SW_U, which this macro overwrites only
//  behind the next barrier -- and a hidden unit that survived both ReLU and Dropout, i.e. whose SAVED value is positive, passes its
//  gradient scaled by 1 / (1 - p))
#define GNT_FF_BWD(L, l1, l2, f_slot, ln_idx, xh_slot, rstd_slot, site)                                        \
    if (dr.on) {                                                                                               \
        STAGE_BEGIN                                                                                            \
        for (int j = 0; j < 16; ++j) SMP(SW_U, P0 + j) = SMP(SW_DCUR, P0 + j) * gnt_keep(dr, (site) + 1, smp_idx * 64 + P0 + j); \
        STAGE_END                                                                                              \
    }                                                                                                          \
    STAGE_BEGIN                                                                                                \
    for (int ch = 0; ch < 4; ++ch) {                                                                           \
        float df[16];                                                                                          \
        for (int j = 0; j < 16; ++j) df[j] = 0.f;                                                              \
        gnt_slice_bwd((L) + gnt_w(l2), 64, 256, ch * 64 + P0, SMPP(dr.on ? SW_U : SW_DCUR), (size_t)c.S, df);  \
        for (int j = 0; j < 16; ++j)                                                                           \
            SMP(SW_T, ch * 64 + P0 + j) = SMP(f_slot, ch * 64 + P0 + j) > 0.f ? df[j] * (dr.on ? dr.scale : 1.f) : 0.f; \
    }                                                                                                          \
    STAGE_END                                                                                                  \
    STAGE_BEGIN                                                                                                \
    float dy[16];                                                                                              \
    for (int j = 0; j < 16; ++j) dy[j] = 0.f;                                                                  \
    gnt_slice_bwd((L) + gnt_w(l1), 256, 64, P0, SMPP(SW_T), (size_t)c.S, dy);                                  \
    for (int j = 0; j < 16; ++j) SMP(SW_U, P0 + j) = dy[j];                                                    \
    STAGE_END                                                                                                  \
    STAGE_BEGIN                                                                                                \
    gnt_ln_bwd_slice(c, SW_U, (L) + gnt_ln_w(ln_idx), xh_slot, rstd_slot, SW_DCUR);                            \
    STAGE_END

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_gnt_fwd(const float* __restrict__ blob, const float* __restrict__ rgb_feat_all,
                                                  const float* __restrict__ ray_diff, const float* __restrict__ mask,
                                                  const float* __restrict__ pts, const float* __restrict__ ray_d, int S, int V,
                                                  int depth, int save, float* __restrict__ rgb_out, float* __restrict__ ws,
                                                  int64_t row_floats, int64_t smp_floats, float* __restrict__ alpha_out, GntDrop dr,
                                                  int64_t ray0) {
    __shared__ float mean_h[64];
    GNT_SETUP()
    c.rgb_feat = rgb_feat_all + ray * S * V * 35;
    const unsigned long long gray = (unsigned long long)(ray0 + ray);          // the ray's index in the whole batch: Dropout indices
    const unsigned long long smp_idx = gray * (unsigned long long)S + (unsigned long long)c.s;
    // ---- stem: X_v = W2 relu(W1 rgb_feat_v + b1) + b2 ; q = max over views (first maximum wins, like torch.max)
    for (int v = 0; v < V; ++v) {
        STAGE_BEGIN
        const float* rf = c.rgb_feat + ((size_t)c.s * V + v) * 35;
        float r1[16];
        gnt_slice(blob, 64, P0, blob + 2 * 35 * 64, 35, rf, 1, r1);
        for (int j = 0; j < 16; ++j) ROW(RW_R1, P0 + j) = fmaxf(r1[j], 0.f);
        STAGE_END
        STAGE_BEGIN
        float x[16];
        gnt_slice(blob + GNT_STEM1, 64, P0, blob + GNT_STEM1 + 2 * 64 * 64, 64, ROWP(RW_R1), ROWSTRIDE, x);
        for (int j = 0; j < 16; ++j) {
            ROW(RW_X, P0 + j) = x[j];
            if (v == 0 || x[j] > SMP(SW_CUR, P0 + j)) {
                SMP(SW_CUR, P0 + j) = x[j];
                SMP(SW_AMAX, P0 + j) = (float)v;
            }
        }
        STAGE_END
    }
    // ---- positional encodings of the sample position (part 0) and of the unit view direction (part 1): 3 -> 63 each
    STAGE_BEGIN
    if (c.part < 2) {
        const float* p3 = pts + (ray * S + c.s) * 3;
        const float* d3 = ray_d + ray * 3;
        float dn = sqrtf(d3[0] * d3[0] + d3[1] * d3[1] + d3[2] * d3[2]);
        float x[3];
        for (int a = 0; a < 3; ++a) x[a] = c.part == 0 ? p3[a] : d3[a] / dn;
        for (int a = 0; a < 3; ++a) SMP(SW_PE, c.part * 63 + a) = x[a];
        float freq = 1.f;
        for (int k = 0; k < 10; ++k) {
            for (int a = 0; a < 3; ++a) {
                SMP(SW_PE, c.part * 63 + 3 + k * 6 + a) = sinf(x[a] * freq);
                SMP(SW_PE, c.part * 63 + 3 + k * 6 + 3 + a) = cosf(x[a] * freq);
            }
            freq *= 2.f;
        }
    }
    STAGE_END
    for (int i = 0; i < depth; ++i) {
        const float* L = blob + gnt_layer_base(i);
        const int ls = SW_BASE + (save ? i : 0) * SW_LAYER;
        const int lr = RW_BASE + (save ? i : 0) * RW_LAYER;
        // ================= view transformer =================
        STAGE_BEGIN
        gnt_ln_slice(c, SW_CUR, L + gnt_ln_w(0), L + gnt_ln_b(0), 1e-6f, ls + SL_XH1, ls + SL_RSTD1, SW_T);
        STAGE_END
        STAGE_BEGIN
        float Q[16];
        gnt_slice(L + gnt_wt(GV_Q), 64, P0, nullptr, 64, SMPP(SW_T), (size_t)c.S, Q);
        for (int j = 0; j < 16; ++j) SMP(SW_QV, P0 + j) = Q[j];
        STAGE_END
        float m[16];
        for (int j = 0; j < 16; ++j) m[j] = -3.0e38f;
        for (int v = 0; v < V; ++v) {
            STAGE_BEGIN
            float K[16];
            gnt_slice(L + gnt_wt(GV_K), 64, P0, nullptr, 64, ROWP(RW_X), ROWSTRIDE, K);
            for (int j = 0; j < 16; ++j) ROW(RW_T, P0 + j) = K[j];
            STAGE_END
            STAGE_BEGIN
            const float* rd = c.ray_diff + ((size_t)c.s * V + v) * 4;
            float h0[8];
            for (int n = 0; n < 8; ++n) {
                float t = L[gnt_b(GV_POS0) + n];
                for (int k = 0; k < 4; ++k) t = fmaf(L[gnt_wt(GV_POS0) + k * 8 + n], rd[k], t);
                h0[n] = fmaxf(t, 0.f);
            }
            float Vv[16];
            gnt_slice(L + gnt_wt(GV_V), 64, P0, nullptr, 64, ROWP(RW_T), ROWSTRIDE, Vv);
            for (int j = 0; j < 16; ++j) {
                float pos = L[gnt_b(GV_POS2) + P0 + j];
                for (int k = 0; k < 8; ++k) pos = fmaf(L[gnt_wt(GV_POS2) + k * 64 + P0 + j], h0[k], pos);
                ROW(lr + RWL_VP, P0 + j) = Vv[j] + pos;
                ROW(RW_T2, P0 + j) = ROW(RW_T, P0 + j) - SMP(SW_QV, P0 + j) + pos;      // k - q + pos
            }
            STAGE_END
            STAGE_BEGIN
            for (int j = 0; j < 2; ++j) {          // 8 hidden units of attn_fc, two per part
                int n = c.part * 2 + j;
                float t = L[gnt_b(GV_ATT0) + n];
                for (int k = 0; k < 64; ++k) t = fmaf(L[gnt_wt(GV_ATT0) + k * 8 + n], ROW(RW_T2, k), t);
                ROW(lr + RWL_H, n) = fmaxf(t, 0.f);
            }
            STAGE_END
            STAGE_BEGIN
            const float mk = c.mask[(size_t)c.s * V + v];
            for (int j = 0; j < 16; ++j) {
                float t = L[gnt_b(GV_ATT2) + P0 + j];
                for (int k = 0; k < 8; ++k) t = fmaf(L[gnt_wt(GV_ATT2) + k * 64 + P0 + j], ROW(lr + RWL_H, k), t);
                if (mk == 0.f) t = -1e9f;
                ROW(lr + RWL_PROB, P0 + j) = t;
                m[j] = fmaxf(m[j], t);
            }
            STAGE_END
        }
        STAGE_BEGIN      // softmax over the views (per channel) and the attention-weighted sum, own slice only
        float sum[16], u[16];
        for (int j = 0; j < 16; ++j) sum[j] = u[j] = 0.f;
        for (int v = 0; v < V; ++v)
            for (int j = 0; j < 16; ++j) {
                float e = expf(ROW(lr + RWL_PROB, P0 + j) - m[j]);
                ROW(lr + RWL_PROB, P0 + j) = e;
                sum[j] += e;
            }
        for (int v = 0; v < V; ++v)
            for (int j = 0; j < 16; ++j) {
                float pr = ROW(lr + RWL_PROB, P0 + j) / sum[j];
                ROW(lr + RWL_PROB, P0 + j) = pr;           // saved BEFORE Dropout site 0 (:85); the backward regenerates the mask
                u[j] = fmaf(ROW(lr + RWL_VP, P0 + j), pr * gnt_keep(dr, 8 * i + 0, (smp_idx * V + v) * 64 + P0 + j), u[j]);
            }
        for (int j = 0; j < 16; ++j) SMP(SW_U, P0 + j) = u[j];
        STAGE_END
        STAGE_BEGIN
        float o[16];
        gnt_slice(L + gnt_wt(GV_OUT), 64, P0, L + gnt_b(GV_OUT), 64, SMPP(SW_U), (size_t)c.S, o);
        for (int j = 0; j < 16; ++j) SMP(SW_CUR, P0 + j) += o[j] * gnt_keep(dr, 8 * i + 1, smp_idx * 64 + P0 + j);       // :88
        STAGE_END
        STAGE_BEGIN
        gnt_ln_slice(c, SW_CUR, L + gnt_ln_w(1), L + gnt_ln_b(1), 1e-6f, ls + SL_XH2, ls + SL_RSTD2, SW_T);
        STAGE_END
        GNT_FF(L, GV_FF1, GV_FF2, ls + SL_F, 8 * i + 2)
        // ================= positional MLP on even layers =================
        if ((i & 1) == 0) {
            STAGE_BEGIN
            float g[16];
            gnt_slice(L + gnt_wt(GQ_0), 64, P0, L + gnt_b(GQ_0), 64, SMPP(SW_CUR), (size_t)c.S, g);
            nf_dense_ws_ld<16>(L + gnt_wt(GQ_0) + 64 * 64 + P0, 64, 126, SMPP(SW_PE), (size_t)c.S, g);
            for (int j = 0; j < 16; ++j) SMP(ls + SL_G, P0 + j) = fmaxf(g[j], 0.f);
            STAGE_END
            STAGE_BEGIN
            float q3[16];
            gnt_slice(L + gnt_wt(GQ_2), 64, P0, L + gnt_b(GQ_2), 64, SMPP(ls + SL_G), (size_t)c.S, q3);
            for (int j = 0; j < 16; ++j) SMP(SW_CUR, P0 + j) = q3[j];
            STAGE_END
        }
        // ================= ray transformer =================
        STAGE_BEGIN
        gnt_ln_slice(c, SW_CUR, L + gnt_ln_w(2), L + gnt_ln_b(2), 1e-6f, ls + SL_RXH1, ls + SL_RRSTD1, SW_T);
        STAGE_END
        STAGE_BEGIN
        for (int which = 0; which < 3; ++which) {
            float o[16];
            gnt_slice(L + gnt_wt(GR_Q + which), 64, P0, nullptr, 64, SMPP(SW_T), (size_t)c.S, o);
            const int slot = ls + (which == 0 ? SL_QH : (which == 1 ? SL_KH : SL_VH));
            for (int j = 0; j < 16; ++j) SMP(slot, P0 + j) = o[j];
        }
        STAGE_END
        STAGE_BEGIN      // attention over the S samples: part = head
        {
            float q[16];
            for (int d = 0; d < 16; ++d) q[d] = SMP(ls + SL_QH, P0 + d) * 0.25f;     // 1 / sqrt(16)
            const float* Kb = c.ws_smp + (size_t)(ls + SL_KH + P0) * S;
            const float* Vb = c.ws_smp + (size_t)(ls + SL_VH + P0) * S;
            float mx = -3.0e38f;
            for (int k = 0; k < S; ++k) mx = fmaxf(mx, gnt_score(Kb, (size_t)S, k, q));
            float l = 0.f, acc[16];
            for (int d = 0; d < 16; ++d) acc[d] = 0.f;
            const unsigned long long arow = ((gray * 4 + (unsigned long long)c.part) * S + (unsigned long long)c.s) * S;     // [R,4,S,S]
            for (int k = 0; k < S; ++k) {
                float p = expf(gnt_score(Kb, (size_t)S, k, q) - mx);
                l += p;                                            // the softmax normalises BEFORE Dropout site 4 (:162)
                p *= gnt_keep(dr, 8 * i + 4, arow + k);
                for (int d = 0; d < 16; ++d) acc[d] = fmaf(p, Vb[(size_t)d * S + k], acc[d]);
            }
            for (int d = 0; d < 16; ++d) {
                float o = acc[d] / l;
                SMP(SW_U, P0 + d) = o;
                SMP(ls + SL_OUTA, P0 + d) = o;
            }
            SMP(ls + SL_ML, c.part) = mx;
            SMP(ls + SL_ML, 4 + c.part) = l;
            // ret_alpha (transformer_network.py:196-200, :303-309): attention of the LAST ray transformer, row of sample 0,
            // averaged over the heads -- head `part` parks its row at SW_T + part, summed below
            if (alpha_out && i == depth - 1 && c.s == 0)
                for (int k = 0; k < S; ++k)          // (the attention returned with ret_attn is the dropped one, :162-169)
                    c.ws_smp[(size_t)(SW_T + c.part) * S + k] = expf(gnt_score(Kb, (size_t)S, k, q) - mx) / l * gnt_keep(dr, 8 * i + 4, arow + k);
        }
        STAGE_END
        if (alpha_out && i == depth - 1) {
            STAGE_BEGIN
            if (c.part == 0) {
                float t = 0.f;
                for (int hd = 0; hd < 4; ++hd) t += c.ws_smp[(size_t)(SW_T + hd) * S + c.s];
                alpha_out[ray * S + c.s] = t / 4.f;
            }
            STAGE_END
        }
        STAGE_BEGIN
        float o[16];
        gnt_slice(L + gnt_wt(GR_OUT), 64, P0, L + gnt_b(GR_OUT), 64, SMPP(SW_U), (size_t)c.S, o);
        for (int j = 0; j < 16; ++j) SMP(SW_CUR, P0 + j) += o[j] * gnt_keep(dr, 8 * i + 5, smp_idx * 64 + P0 + j);       // :166
        STAGE_END
        STAGE_BEGIN
        gnt_ln_slice(c, SW_CUR, L + gnt_ln_w(3), L + gnt_ln_b(3), 1e-6f, ls + SL_RXH2, ls + SL_RRSTD2, SW_T);
        STAGE_END
        GNT_FF(L, GR_FF1, GR_FF2, ls + SL_F2, 8 * i + 6)
    }
    // ---- final LayerNorm (eps 1e-5), mean over the samples, rgb_fc
    const float* Fp = blob + gnt_layer_base(depth);
    STAGE_BEGIN
    gnt_ln_slice(c, SW_CUR, Fp, Fp + 64, 1e-5f, SW_XHF, SW_RSTDF, SW_HF);
    STAGE_END
    if (threadIdx.x < 64) {
        float t = 0.f;
        const float* col = c.ws_smp + (size_t)(SW_HF + threadIdx.x) * S;
        for (int k = 0; k < S; ++k) t += col[k];
        mean_h[threadIdx.x] = t / (float)S;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float t = Fp[512 + threadIdx.x];
        for (int j = 0; j < 64; ++j) t = fmaf(Fp[128 + j * 3 + threadIdx.x], mean_h[j], t);
        rgb_out[ray * 3 + threadIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward (reads the activations saved by the forward with save = 1)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_gnt_bwd(const float* __restrict__ blob, const float* __restrict__ ray_diff,
                                                  const float* __restrict__ mask, const float* __restrict__ d_rgb, int S, int V,
                                                  int depth, float* __restrict__ d_rgb_feat, float* __restrict__ ws,
                                                  int64_t row_floats, int64_t smp_floats, GntDrop dr) {
    GNT_SETUP()
    c.rgb_feat = nullptr;
    const unsigned long long gray = (unsigned long long)ray;
    const unsigned long long smp_idx = gray * (unsigned long long)S + (unsigned long long)c.s;
    const float* Fp = blob + gnt_layer_base(depth);
    // ---- rgb_fc, mean over samples, final LayerNorm
    STAGE_BEGIN
    const float* g = d_rgb + ray * 3;
    for (int j = P0; j < P0 + 16; ++j) {
        float dm = Fp[320 + j] * g[0] + Fp[320 + 64 + j] * g[1] + Fp[320 + 128 + j] * g[2];
        SMP(SW_U, j) = dm / (float)S;
        SMP(SW_DCUR, j) = 0.f;
    }
    for (int v = 0; v < V; ++v)
        for (int j = P0; j < P0 + 16; ++j) ROW(RW_DX, j) = 0.f;
    STAGE_END
    STAGE_BEGIN
    gnt_ln_bwd_slice(c, SW_U, Fp, SW_XHF, SW_RSTDF, SW_DCUR);
    STAGE_END
    for (int i = depth - 1; i >= 0; --i) {
        const float* L = blob + gnt_layer_base(i);
        const int ls = SW_BASE + i * SW_LAYER;
        const int lr = RW_BASE + i * RW_LAYER;
        // ================= ray transformer backward =================
        GNT_FF_BWD(L, GR_FF1, GR_FF2, ls + SL_F2, 3, ls + SL_RXH2, ls + SL_RRSTD2, 8 * i + 6)        // DCUR = d q1
        if (dr.on) {         // Dropout site 5 on out_fc's output: a masked copy of the incoming gradient
            STAGE_BEGIN
            for (int j = 0; j < 16; ++j) SMP(SW_T, P0 + j) = SMP(SW_DCUR, P0 + j) * gnt_keep(dr, 8 * i + 5, smp_idx * 64 + P0 + j);
            STAGE_END
        }
        STAGE_BEGIN
        float d_out[16];
        for (int j = 0; j < 16; ++j) d_out[j] = 0.f;
        gnt_slice_bwd(L + gnt_w(GR_OUT), 64, 64, P0, SMPP(dr.on ? SW_T : SW_DCUR), (size_t)c.S, d_out);
        float D = 0.f;
        for (int d = 0; d < 16; ++d) {
            SMP(SW_GO, P0 + d) = d_out[d];
            D = fmaf(d_out[d], SMP(ls + SL_OUTA, P0 + d), D);
        }
        SMP(SW_DH, c.part) = D;
        STAGE_END
        STAGE_BEGIN      // attention backward for head = part: dQ as query, dK / dV as key
        {
            const float* Qb = c.ws_smp + (size_t)(ls + SL_QH + P0) * S;
            const float* Kb = c.ws_smp + (size_t)(ls + SL_KH + P0) * S;
            const float* Vb = c.ws_smp + (size_t)(ls + SL_VH + P0) * S;
            const float* Gb = c.ws_smp + (size_t)(SW_GO + P0) * S;
            const float* Mb = c.ws_smp + (size_t)(ls + SL_ML + c.part) * S;
            const float* Lb = c.ws_smp + (size_t)(ls + SL_ML + 4 + c.part) * S;
            const float* Db = c.ws_smp + (size_t)(SW_DH + c.part) * S;
            {
                float q[16], go[16], dq[16];
                for (int d = 0; d < 16; ++d) {
                    q[d] = Qb[(size_t)d * S + c.s] * 0.25f;
                    go[d] = Gb[(size_t)d * S + c.s];
                    dq[d] = 0.f;
                }
                float mx = Mb[c.s], rl = 1.f / Lb[c.s], D = Db[c.s];
                // Dropout site 4 on the probabilities: O = sum_k P_k m_k V_k, so d P_k = m_k (go . V_k) and D = sum_k P_k d P_k = go . O
                // with the DROPPED output O the forward saved
                const unsigned long long arow = ((gray * 4 + (unsigned long long)c.part) * S + (unsigned long long)c.s) * S;
                for (int k = 0; k < S; ++k) {
                    float p = expf(gnt_score(Kb, (size_t)S, k, q) - mx) * rl;
                    float dS = p * (gnt_score(Vb, (size_t)S, k, go) * gnt_keep(dr, 8 * i + 4, arow + k) - D);
                    for (int d = 0; d < 16; ++d) dq[d] = fmaf(dS, Kb[(size_t)d * S + k], dq[d]);
                }
                for (int d = 0; d < 16; ++d) SMP(SW_DQ, P0 + d) = dq[d] * 0.25f;
            }
            {
                float kk[16], vv[16], dk[16], dv[16];
                for (int d = 0; d < 16; ++d) {
                    kk[d] = Kb[(size_t)d * S + c.s];
                    vv[d] = Vb[(size_t)d * S + c.s];
                    dk[d] = dv[d] = 0.f;
                }
                for (int qi = 0; qi < S; ++qi) {
                    float sc = 0.f, dA = 0.f;
                    for (int d = 0; d < 16; ++d) {
                        sc = fmaf(Qb[(size_t)d * S + qi] * 0.25f, kk[d], sc);
                        dA = fmaf(Gb[(size_t)d * S + qi], vv[d], dA);
                    }
                    float p = expf(sc - Mb[qi]) / Lb[qi];
                    const float mk = gnt_keep(dr, 8 * i + 4, ((gray * 4 + (unsigned long long)c.part) * S + (unsigned long long)qi) * S + c.s);
                    dA *= mk;
                    float dS = p * (dA - Db[qi]);
                    for (int d = 0; d < 16; ++d) {
                        dv[d] = fmaf(p * mk, Gb[(size_t)d * S + qi], dv[d]);
                        dk[d] = fmaf(dS, Qb[(size_t)d * S + qi] * 0.25f, dk[d]);
                    }
                }
                for (int d = 0; d < 16; ++d) {
                    SMP(SW_DK, P0 + d) = dk[d];
                    SMP(SW_DV, P0 + d) = dv[d];
                }
            }
        }
        STAGE_END
        STAGE_BEGIN
        float dx[16];
        for (int j = 0; j < 16; ++j) dx[j] = 0.f;
        gnt_slice_bwd(L + gnt_w(GR_Q), 64, 64, P0, SMPP(SW_DQ), (size_t)c.S, dx);
        gnt_slice_bwd(L + gnt_w(GR_K), 64, 64, P0, SMPP(SW_DK), (size_t)c.S, dx);
        gnt_slice_bwd(L + gnt_w(GR_V), 64, 64, P0, SMPP(SW_DV), (size_t)c.S, dx);
        for (int j = 0; j < 16; ++j) SMP(SW_U, P0 + j) = dx[j];
        STAGE_END
        STAGE_BEGIN
        gnt_ln_bwd_slice(c, SW_U, L + gnt_ln_w(2), ls + SL_RXH1, ls + SL_RRSTD1, SW_DCUR);
        STAGE_END
        // ================= positional MLP backward (even layers) =================
        if ((i & 1) == 0) {
            STAGE_BEGIN
            float dg[16];
            for (int j = 0; j < 16; ++j) dg[j] = 0.f;
            gnt_slice_bwd(L + gnt_w(GQ_2), 64, 64, P0, SMPP(SW_DCUR), (size_t)c.S, dg);
            for (int j = 0; j < 16; ++j) SMP(SW_T, P0 + j) = SMP(ls + SL_G, P0 + j) > 0.f ? dg[j] : 0.f;
            STAGE_END
            STAGE_BEGIN
            float dq[16];
            for (int j = 0; j < 16; ++j) dq[j] = 0.f;
            gnt_slice_bwd(L + gnt_w(GQ_0), 64, 190, P0, SMPP(SW_T), (size_t)c.S, dq);
            for (int j = 0; j < 16; ++j) SMP(SW_DCUR, P0 + j) = dq[j];
            STAGE_END
        }
        // ================= view transformer backward =================
        GNT_FF_BWD(L, GV_FF1, GV_FF2, ls + SL_F, 1, ls + SL_XH2, ls + SL_RSTD2, 8 * i + 2)           // DCUR = d q1
        float du[16], sp[16], dqs[16];
        if (dr.on) {         // Dropout site 1 on out_fc's output
            STAGE_BEGIN
            for (int j = 0; j < 16; ++j) SMP(SW_T, P0 + j) = SMP(SW_DCUR, P0 + j) * gnt_keep(dr, 8 * i + 1, smp_idx * 64 + P0 + j);
            STAGE_END
        }
        STAGE_BEGIN
        for (int j = 0; j < 16; ++j) du[j] = sp[j] = dqs[j] = 0.f;
        gnt_slice_bwd(L + gnt_w(GV_OUT), 64, 64, P0, SMPP(dr.on ? SW_T : SW_DCUR), (size_t)c.S, du);
        // Dropout site 0 on the probabilities: u = sum_v (V_v + pos_v) pr_v m_v, so d pr_v = m_v (V_v + pos_v) du
        for (int v = 0; v < V; ++v)
            for (int j = 0; j < 16; ++j)
                sp[j] = fmaf(ROW(lr + RWL_PROB, P0 + j) * gnt_keep(dr, 8 * i + 0, (smp_idx * V + v) * 64 + P0 + j) * ROW(lr + RWL_VP, P0 + j), du[j], sp[j]);
        STAGE_END
        for (int v = 0; v < V; ++v) {
            STAGE_BEGIN
            const float mk = c.mask[(size_t)c.s * V + v];
            for (int j = 0; j < 16; ++j) {
                float pr = ROW(lr + RWL_PROB, P0 + j);
                const float dm = gnt_keep(dr, 8 * i + 0, (smp_idx * V + v) * 64 + P0 + j);
                ROW(RW_T2, P0 + j) = mk == 0.f ? 0.f : pr * (ROW(lr + RWL_VP, P0 + j) * du[j] * dm - sp[j]);    // d logit
                ROW(RW_T, P0 + j) = pr * dm * du[j];                                                              // d Vv
            }
            STAGE_END
            STAGE_BEGIN
            for (int j = 0; j < 2; ++j) {
                int n = c.part * 2 + j;
                float t = 0.f;
                for (int k = 0; k < 64; ++k) t = fmaf(L[gnt_w(GV_ATT2) + k * 8 + n], ROW(RW_T2, k), t);
                ROW(RW_R1H, n) = ROW(lr + RWL_H, n) > 0.f ? t : 0.f;
            }
            STAGE_END
            STAGE_BEGIN
            float dk[16];
            for (int j = 0; j < 16; ++j) {
                float t = 0.f;
                for (int n = 0; n < 8; ++n) t = fmaf(L[gnt_w(GV_ATT0) + n * 64 + P0 + j], ROW(RW_R1H, n), t);
                dqs[j] += t;
                dk[j] = t;                                        // d a
            }
            gnt_slice_bwd(L + gnt_w(GV_V), 64, 64, P0, ROWP(RW_T), ROWSTRIDE, dk);      // d K = d a + Wv^T d Vv
            for (int j = 0; j < 16; ++j) ROW(RW_T3, P0 + j) = dk[j];
            STAGE_END
            STAGE_BEGIN
            float dx[16];
            for (int j = 0; j < 16; ++j) dx[j] = 0.f;
            gnt_slice_bwd(L + gnt_w(GV_K), 64, 64, P0, ROWP(RW_T3), ROWSTRIDE, dx);
            for (int j = 0; j < 16; ++j) ROW(RW_DX, P0 + j) += dx[j];
            STAGE_END
        }
        STAGE_BEGIN
        for (int j = 0; j < 16; ++j) SMP(SW_T, P0 + j) = -dqs[j];        // d Q = - sum_v d a_v
        STAGE_END
        STAGE_BEGIN
        float dx[16];
        for (int j = 0; j < 16; ++j) dx[j] = 0.f;
        gnt_slice_bwd(L + gnt_w(GV_Q), 64, 64, P0, SMPP(SW_T), (size_t)c.S, dx);
        for (int j = 0; j < 16; ++j) SMP(SW_U, P0 + j) = dx[j];
        STAGE_END
        STAGE_BEGIN
        gnt_ln_bwd_slice(c, SW_U, L + gnt_ln_w(0), ls + SL_XH1, ls + SL_RSTD1, SW_DCUR);
        STAGE_END
    }
    // ---- q0 = max over views routes its gradient to the arg-max view; then the stem
    STAGE_BEGIN
    for (int j = P0; j < P0 + 16; ++j) {
        int v = (int)SMP(SW_AMAX, j);
        ROW(RW_DX, j) += SMP(SW_DCUR, j);
    }
    STAGE_END
    for (int v = 0; v < V; ++v) {
        STAGE_BEGIN
        float dr[16];
        for (int j = 0; j < 16; ++j) dr[j] = 0.f;
        gnt_slice_bwd(blob + GNT_STEM1 + 64 * 64, 64, 64, P0, ROWP(RW_DX), ROWSTRIDE, dr);
        for (int j = 0; j < 16; ++j) ROW(RW_T, P0 + j) = ROW(RW_R1, P0 + j) > 0.f ? dr[j] : 0.f;
        STAGE_END
        STAGE_BEGIN
        float* o = d_rgb_feat + ((ray * S + c.s) * V + v) * 35;
        for (int n = c.part * 9; n < min(35, c.part * 9 + 9); ++n) {       // 35 input channels split 9 / 9 / 9 / 8
            float t = 0.f;
            for (int k = 0; k < 64; ++k) t = fmaf(blob[35 * 64 + k * 35 + n], ROW(RW_T, k), t);
            o[n] = t;
        }
        STAGE_END
    }
}

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
static int gnt_check(const char* who, int64_t R, int S, int V, int depth) {
    NF_REQUIRE(R >= 0 && S >= 1 && S <= 256 && V >= 1 && V <= 64 && depth >= 1 && depth <= 16,
               "%s: need 1 <= S <= 256, 1 <= V <= 64, 1 <= depth <= 16 (got R %lld S %d V %d depth %d)", who, (long long)R, S, V, depth);
    return 0;
}

/* rgb_feat [R,S,V,35], ray_diff [R,S,V,4], mask [R,S,V], pts [R,S,3], ray_d [R,3] -> rgb [R,3].
 * save != 0 keeps the activations of ALL rays in `workspace` (nf_gnt_workspace_floats(R,S,V,depth,1)) for nf_gnt_bwd. */
static int gnt_fwd_impl(const char* who, const float* blob, const float* rgb_feat, const float* ray_diff, const float* mask, const float* pts,
                        const float* ray_d, int64_t n_rays, int n_samples, int n_views, int depth, int save, float* rgb,
                        float* alpha, float* workspace, const GntDrop& dr, nf_stream_t stream) {
    if (gnt_check(who, n_rays, n_samples, n_views, depth)) return 1;
    const int S = n_samples, V = n_views;
    const int threads = GNT_PARTS * (((S + 15) / 16) * 16);       // 4 lanes per sample, multiple of 64
    const int64_t rf = gnt_row_floats(depth, save), sf = gnt_smp_floats(depth, save);
    const int64_t per_ray = (int64_t)S * V * rf + (int64_t)S * sf;
    const int64_t step = save ? n_rays : GNT_RAYS_PER_LAUNCH;
    for (int64_t r0 = 0; r0 < n_rays; r0 += step) {
        int64_t nr = n_rays - r0 < step ? n_rays - r0 : step;
        hipLaunchKernelGGL(k_gnt_fwd, dim3((unsigned)nr), dim3(threads), 0, (hipStream_t)stream, blob, rgb_feat + r0 * S * V * 35,
                           ray_diff + r0 * S * V * 4, mask + r0 * S * V, pts + r0 * S * 3, ray_d + r0 * 3, S, V, depth, save ? 1 : 0,
                           rgb + r0 * 3, workspace + (save ? r0 * per_ray : 0), rf, sf, alpha ? alpha + r0 * S : nullptr, dr, r0);
        NF_LAUNCH_CHECK("nf_gnt_fwd");
    }
    return 0;
}

extern "C" int nf_gnt_fwd(const float* blob, const float* rgb_feat, const float* ray_diff, const float* mask, const float* pts,
                          const float* ray_d, int64_t n_rays, int n_samples, int n_views, int depth, int save, float* rgb,
                          float* alpha, float* workspace, nf_stream_t stream) {
    return gnt_fwd_impl("nf_gnt_fwd", blob, rgb_feat, ray_diff, mask, pts, ray_d, n_rays, n_samples, n_views, depth, save, rgb, alpha,
                        workspace, gnt_drop_make(0, 0u, 0.0), stream);
}

/* nf_gnt_fwd in TRAINING mode: the eight Dropout(p) sites of every layer live, masks from the counter-based generator of nf_gnt.h
 * (seed = the call's seed; the backward of this forward must be given the same seed and p). */
extern "C" int nf_gnt_fwd_train(const float* blob, const float* rgb_feat, const float* ray_diff, const float* mask, const float* pts,
                                const float* ray_d, int64_t n_rays, int n_samples, int n_views, int depth, int save, float* rgb,
                                float* alpha, float* workspace, uint32_t seed, double p, nf_stream_t stream) {
    NF_REQUIRE(p >= 0.0 && p < 1.0, "nf_gnt_fwd_train: dropout rate %g outside [0, 1)", p);
    return gnt_fwd_impl("nf_gnt_fwd_train", blob, rgb_feat, ray_diff, mask, pts, ray_d, n_rays, n_samples, n_views, depth, save, rgb, alpha,
                        workspace, gnt_drop_make(1, seed, p), stream);
}

/* d_rgb [R,3] -> d_rgb_feat [R,S,V,35]; `workspace` is the buffer a forward with save = 1 filled */
static int gnt_bwd_impl(const char* who, const float* blob, const float* ray_diff, const float* mask, const float* d_rgb, int64_t n_rays,
                        int n_samples, int n_views, int depth, float* d_rgb_feat, float* workspace, const GntDrop& dr, nf_stream_t stream) {
    if (gnt_check(who, n_rays, n_samples, n_views, depth)) return 1;
    if (n_rays == 0) return 0;
    const int threads = GNT_PARTS * (((n_samples + 15) / 16) * 16);
    hipLaunchKernelGGL(k_gnt_bwd, dim3((unsigned)n_rays), dim3(threads), 0, (hipStream_t)stream, blob, ray_diff, mask, d_rgb,
                       n_samples, n_views, depth, d_rgb_feat, workspace, gnt_row_floats(depth, 1), gnt_smp_floats(depth, 1), dr);
    NF_LAUNCH_CHECK("nf_gnt_bwd");
    return 0;
}

extern "C" int nf_gnt_bwd(const float* blob, const float* ray_diff, const float* mask, const float* d_rgb, int64_t n_rays,
                          int n_samples, int n_views, int depth, float* d_rgb_feat, float* workspace, nf_stream_t stream) {
    return gnt_bwd_impl("nf_gnt_bwd", blob, ray_diff, mask, d_rgb, n_rays, n_samples, n_views, depth, d_rgb_feat, workspace,
                        gnt_drop_make(0, 0u, 0.0), stream);
}

/* backward of nf_gnt_fwd_train(save = 1, seed, p): regenerates the forward's masks */
extern "C" int nf_gnt_bwd_train(const float* blob, const float* ray_diff, const float* mask, const float* d_rgb, int64_t n_rays,
                                int n_samples, int n_views, int depth, float* d_rgb_feat, float* workspace, uint32_t seed, double p,
                                nf_stream_t stream) {
    NF_REQUIRE(p >= 0.0 && p < 1.0, "nf_gnt_bwd_train: dropout rate %g outside [0, 1)", p);
    return gnt_bwd_impl("nf_gnt_bwd_train", blob, ray_diff, mask, d_rgb, n_rays, n_samples, n_views, depth, d_rgb_feat, workspace,
                        gnt_drop_make(1, seed, p), stream);
}
