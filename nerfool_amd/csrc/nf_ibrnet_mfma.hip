// IBRNet on the matrix cores (a4/a5), V in {1,2,4,8,16,32}.
//
// Forward = two kernels per level:
//   A  k_ibr_rows_fwd  -- everything that lives on a (sample, view) ROW: direction MLP, first pooling, base_fc, vis_fc,
//                         vis_fc2, second pooling, colour head + blending softmax.  A wave owns a tile of 32 rows
//                         (32/V samples x V views); rows sit on the MFMA lane (lane & 31), so every cross-view
//                         reduction is a butterfly over V adjacent lanes.
//   B  k_ibr_ray_fwd   -- everything per SAMPLE along a ray: geometry_fc, ray self-attention over S samples, LayerNorm,
//                         density head (one workgroup per ray, K/V staged in LDS).
//
// GEMM formulation (exact fp32, v_mfma_f32_32x32x2_f32): Y^T[N x 32 rows] = W[N x K] . X^T[K x 32 rows].
//   A operand = weights (lane l: W[n = 32*nt + (l&31)][k(step, l>>5)]), pre-arranged on the host into "records" of 64
//   floats (one per lane) in exactly the order the kernel consumes them and staged in LDS (conflict-free ds_read_b32).
//   B operand = activations: lane l = (row l&31, half h = l>>5).
//   The accumulator of a 32-output tile holds feature n(r,h) = (r&3) + 8*(r>>2) + 4*h in register r -- and that is
//   precisely a valid B operand for k-step r of the NEXT layer (k pair = n(r,0), n(r,1)), so activations never leave
//   registers between layers: no LDS round trip, no transposes.  Bias = accumulator init, ELU = in-register VALU.
//
// ref: ibrnet/mlp_network.py:222-274.  Cross-checked on device against the generic kernels of nf_ibrnet.hip and against
// the oracle in tests/.
#include "nf_ibrnet.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__host__ __device__ constexpr int nf_nidx(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- record layout of the MFMA-order weight blob (units: records of 64 floats) -------------------------------------
enum {
    MR_DIR0 = 0,                 // 2
    MR_DIR1 = MR_DIR0 + 2,       // 8   (outputs = feature channels 3..34)
    MR_BASE0 = MR_DIR1 + 8,      // 2 tiles x 3 parts x (16 + 2)
    MR_BASE1 = MR_BASE0 + 108,   // 32
    MR_VIS0 = MR_BASE1 + 32,     // 16
    MR_VIS1 = MR_VIS0 + 16,      // 16  (outputs 0..31 of 33)
    MR_VISB0 = MR_VIS1 + 16,     // 16
    MR_RGB0 = MR_VISB0 + 16,     // 16 + 3
    MR_RECORDS = MR_RGB0 + 19    // 217
};
// ---- small VALU-side tables behind the records (units: floats) ------------------------------------------------------
enum {
    MS_BASE = MR_RECORDS * 64,
    MS_BIAS = MS_BASE,               // 9 tiles x [h][16]: DIR0, DIR1, BASE0 t0, BASE0 t1, BASE1, VIS0, VIS1, VISB0, RGB0
    MS_DIR1C = MS_BIAS + 9 * 32,     // 3 x [h][8] weights of ray_dir_fc.2 rows 0..2, then 3 biases
    MS_VIS1L = MS_DIR1C + 3 * 16 + 3 + 1,   // [h][16] row 32 of vis_fc.2, bias (+pad)
    MS_VISB1 = MS_VIS1L + 32 + 4,    // [h][16] vis_fc2.2, bias
    MS_RGB1 = MS_VISB1 + 32 + 4,     // 8 x [h][8], then 8 biases
    MS_RGB2 = MS_RGB1 + 128 + 8,     // 8 weights, bias, |s|
    MS_END = MS_RGB2 + 12,
    NF_MFMA_BLOB_FLOATS = MS_END
};
enum { BT_DIR0, BT_DIR1, BT_BASE0A, BT_BASE0B, BT_BASE1, BT_VIS0, BT_VIS1, BT_VISB0, BT_RGB0 };

#define NF_SMP_STRIDE 72   // per-sample record written by kernel A: mean2 32 | var2 32 | wmean | rgb 3 | nvalid | vsum | pad

// ---------------------------------------------------------------------------------------------------------------
// host: natural blob (nf_ibrnet.h layout) -> MFMA-order blob
// ---------------------------------------------------------------------------------------------------------------
static void emit_record(float* rec, const float* W, int N, int K, int row_base, int nt, int k0, int k1) {
    for (int lane = 0; lane < 64; ++lane) {
        int i = lane & 31, h = lane >> 5;
        int n = nt * 32 + i, k = h ? k1 : k0;
        rec[lane] = (n < N && k >= 0 && k < K) ? W[(size_t)(row_base + n) * K + k] : 0.f;
    }
}

static void emit_frag_block(float*& rec, const float* W, int N, int K, int row_base, int nt, int kbase, int nsteps) {
    for (int r = 0; r < nsteps; ++r, rec += 64) emit_record(rec, W, N, K, row_base, nt, kbase + nf_nidx(r, 0), kbase + nf_nidx(r, 1));
}

static void emit_bias_tile(float* dst, const float* b, int N, int base) {
    for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 16; ++r) {
            int n = base + nf_nidx(r, h);
            dst[h * 16 + r] = (n >= 0 && n - base < 32 && n < N) ? b[n] : 0.f;
        }
}

extern "C" int64_t nf_ibrnet_mfma_blob_floats(void) { return NF_MFMA_BLOB_FLOATS; }

extern "C" int nf_ibrnet_pack_mfma(const float* nat, float* out) {
    for (int i = 0; i < NF_MFMA_BLOB_FLOATS; ++i) out[i] = 0.f;
    float* rec = out;
    const float* W;
    // DIR0 4 -> 16
    W = nat + nf_lin_w(NF_L_DIR0);
    emit_record(rec, W, 16, 4, 0, 0, 0, 1); rec += 64;
    emit_record(rec, W, 16, 4, 0, 0, 2, 3); rec += 64;
    // DIR1 16 -> channels 3..34 (rows 3..34 of the [35][16] matrix)
    W = nat + nf_lin_w(NF_L_DIR1);
    emit_frag_block(rec, W, 32, 16, 3, 0, 0, 8);
    // BASE0 105 -> 64: per tile, parts mean / var / f: feature block (inputs p+3..p+34) then colour inputs p+0..p+2
    W = nat + nf_lin_w(NF_L_BASE0);
    for (int nt = 0; nt < 2; ++nt)
        for (int p = 0; p < 105; p += 35) {
            emit_frag_block(rec, W, 64, 105, 0, nt, p + 3, 16);
            emit_record(rec, W, 64, 105, 0, nt, p + 0, p + 1); rec += 64;
            emit_record(rec, W, 64, 105, 0, nt, p + 2, -1); rec += 64;
        }
    W = nat + nf_lin_w(NF_L_BASE1);
    emit_frag_block(rec, W, 32, 64, 0, 0, 0, 16);
    emit_frag_block(rec, W, 32, 64, 0, 0, 32, 16);
    W = nat + nf_lin_w(NF_L_VIS0);
    emit_frag_block(rec, W, 32, 32, 0, 0, 0, 16);
    W = nat + nf_lin_w(NF_L_VIS1);
    emit_frag_block(rec, W, 32, 32, 0, 0, 0, 16);           // rows 0..31 of the [33][32] matrix
    W = nat + nf_lin_w(NF_L_VISB0);
    emit_frag_block(rec, W, 32, 32, 0, 0, 0, 16);
    W = nat + nf_lin_w(NF_L_RGB0);
    emit_frag_block(rec, W, 16, 37, 0, 0, 0, 16);
    emit_record(rec, W, 16, 37, 0, 0, 32, 33); rec += 64;   // vis2 | ray_diff[0]
    emit_record(rec, W, 16, 37, 0, 0, 34, 35); rec += 64;   // ray_diff[1] | ray_diff[2]
    emit_record(rec, W, 16, 37, 0, 0, 36, -1); rec += 64;   // ray_diff[3] | -
    if (rec - out != MR_RECORDS * 64) return 1;
    // bias tiles
    float* bt = out + MS_BIAS;
    emit_bias_tile(bt + BT_DIR0 * 32, nat + nf_lin_b(NF_L_DIR0), 16, 0);
    emit_bias_tile(bt + BT_DIR1 * 32, nat + nf_lin_b(NF_L_DIR1) + 3, 32, 0);
    emit_bias_tile(bt + BT_BASE0A * 32, nat + nf_lin_b(NF_L_BASE0), 32, 0);
    emit_bias_tile(bt + BT_BASE0B * 32, nat + nf_lin_b(NF_L_BASE0) + 32, 32, 0);
    emit_bias_tile(bt + BT_BASE1 * 32, nat + nf_lin_b(NF_L_BASE1), 32, 0);
    emit_bias_tile(bt + BT_VIS0 * 32, nat + nf_lin_b(NF_L_VIS0), 32, 0);
    emit_bias_tile(bt + BT_VIS1 * 32, nat + nf_lin_b(NF_L_VIS1), 32, 0);
    emit_bias_tile(bt + BT_VISB0 * 32, nat + nf_lin_b(NF_L_VISB0), 32, 0);
    emit_bias_tile(bt + BT_RGB0 * 32, nat + nf_lin_b(NF_L_RGB0), 16, 0);
    // VALU-side vectors in [h][r] order
    const float* Wd = nat + nf_lin_w(NF_L_DIR1);             // [35][16]
    for (int c = 0; c < 3; ++c) {
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 8; ++r) out[MS_DIR1C + c * 16 + h * 8 + r] = Wd[c * 16 + nf_nidx(r, h)];
        out[MS_DIR1C + 48 + c] = nat[nf_lin_b(NF_L_DIR1) + c];
    }
    const float* Wv = nat + nf_lin_w(NF_L_VIS1) + 32 * 32;   // row 32 of [33][32]
    const float* Wu = nat + nf_lin_w(NF_L_VISB1);            // [1][32]
    for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 16; ++r) {
            out[MS_VIS1L + h * 16 + r] = Wv[nf_nidx(r, h)];
            out[MS_VISB1 + h * 16 + r] = Wu[nf_nidx(r, h)];
        }
    out[MS_VIS1L + 32] = nat[nf_lin_b(NF_L_VIS1) + 32];
    out[MS_VISB1 + 32] = nat[nf_lin_b(NF_L_VISB1)];
    const float* W1 = nat + nf_lin_w(NF_L_RGB1);             // [8][16]
    for (int j = 0; j < 8; ++j) {
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 8; ++r) out[MS_RGB1 + j * 16 + h * 8 + r] = W1[j * 16 + nf_nidx(r, h)];
        out[MS_RGB1 + 128 + j] = nat[nf_lin_b(NF_L_RGB1) + j];
        out[MS_RGB2 + j] = nat[nf_lin_w(NF_L_RGB2) + j];
    }
    out[MS_RGB2 + 8] = nat[nf_lin_b(NF_L_RGB2)];
    out[MS_RGB2 + 9] = fabsf(nat[0]);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float mf_elu(float x) { return x > 0.f ? x : expm1f(x); }
__device__ __forceinline__ float mf_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

template <int V>
__device__ __forceinline__ float grp_sum(float x) {
#pragma unroll
    for (int m = 1; m < V; m <<= 1) x += __shfl_xor(x, m, NF_WAVE);
    return x;
}
template <int V>
__device__ __forceinline__ float grp_min(float x) {
#pragma unroll
    for (int m = 1; m < V; m <<= 1) x = fminf(x, __shfl_xor(x, m, NF_WAVE));
    return x;
}
template <int V>
__device__ __forceinline__ float grp_max(float x) {
#pragma unroll
    for (int m = 1; m < V; m <<= 1) x = fmaxf(x, __shfl_xor(x, m, NF_WAVE));
    return x;
}
__device__ __forceinline__ float half_sum(float x) { return x + __shfl_xor(x, 32, NF_WAVE); }

__device__ __forceinline__ f32x16 bias_tile(const float* lds, int tile, int h) {
    f32x16 a;
    const float* b = lds + MS_BIAS + tile * 32 + h * 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = b[r];
    return a;
}

// acc += W(records rec .. rec+NSTEPS-1) . x  (k-step r consumes register r of the fragment x)
template <int NSTEPS>
__device__ __forceinline__ f32x16 gemm_frag(const float* lds, int rec, int lane, const f32x16& x, f32x16 acc) {
#pragma unroll
    for (int r = 0; r < NSTEPS; ++r) acc = NF_MFMA(lds[(rec + r) * 64 + lane], x[r], acc);
    return acc;
}

__device__ __forceinline__ f32x16 elu16(f32x16 a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = mf_elu(a[r]);
    return a;
}

// dot of a fragment with an [h][16] VALU vector, summed over both lane halves (all 32 features of the row)
__device__ __forceinline__ float dot_frag16(const float* vec_h, const f32x16& x) {
    float d = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) d = fmaf(vec_h[r], x[r], d);
    return half_sum(d);
}

// ---------------------------------------------------------------------------------------------------------------
// kernel A (forward)
// ---------------------------------------------------------------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(256, 2) k_ibr_rows_fwd(const float* __restrict__ wblob, const float* __restrict__ rgb_feat,
                                                      const float* __restrict__ ray_diff, const float* __restrict__ mask,
                                                      int64_t n_samples, int aa, float* __restrict__ smp) {
    HIP_DYNAMIC_SHARED(float, lds)
    for (int i = threadIdx.x; i < NF_MFMA_BLOB_FLOATS; i += blockDim.x) lds[i] = wblob[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int64_t n_rows = n_samples * V;
    const int64_t n_tiles = (n_rows + 31) / 32;
    const float s_abs = lds[MS_RGB2 + 9];
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        // compiler barrier: keeps the (tile-invariant) weight reads from being hoisted out of the loop into 200+ VGPRs
        asm volatile("" ::: "memory");
        int64_t row = tile * 32 + m;
        const bool live = row < n_rows;
        if (!live) row = n_rows - 1;
        const int64_t sample = row / V;
        const int v = (int)(row - sample * V);
        const float* rf = rgb_feat + row * 35;
        const float* rd = ray_diff + row * 4;
        f32x16 F;
#pragma unroll
        for (int r = 0; r < 16; ++r) F[r] = rf[3 + nf_nidx(r, h)];
        const float c0 = rf[0], c1 = rf[1], c2 = rf[2];
        const float rd0 = rd[0], rd1 = rd[1], rd2 = rd[2], rd3 = rd[3];
        const float mk = mask[row];

        // ---- direction MLP 4 -> 16 -> 35, f = rgb_feat + dir_feat
        f32x16 d1 = bias_tile(lds, BT_DIR0, h);
        d1 = NF_MFMA(lds[(MR_DIR0 + 0) * 64 + lane], h ? rd1 : rd0, d1);
        d1 = NF_MFMA(lds[(MR_DIR0 + 1) * 64 + lane], h ? rd3 : rd2, d1);
#pragma unroll
        for (int r = 0; r < 8; ++r) d1[r] = mf_elu(d1[r]);
        {
            f32x16 df = gemm_frag<8>(lds, MR_DIR1, lane, d1, bias_tile(lds, BT_DIR1, h));
#pragma unroll
            for (int r = 0; r < 16; ++r) F[r] += mf_elu(df[r]);
        }
        float fc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* wv = lds + MS_DIR1C + c * 16 + h * 8;
            float d = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) d = fmaf(wv[r], d1[r], d);
            d = half_sum(d) + lds[MS_DIR1C + 48 + c];
            fc[c] = (c == 0 ? c0 : (c == 1 ? c1 : c2)) + mf_elu(d);
        }
        // ---- first pooling weight (mlp_network.py:234-241)
        float w;
        if (aa) {
            float e = expf(s_abs * (rd3 - 1.f));
            w = (e - grp_min<V>(e)) * mk;
        } else {
            w = mk;
        }
        w = w / (grp_sum<V>(w) + 1e-8f);
        // ---- weighted mean / variance over the V views of the sample
        f32x16 MEAN, VAR;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float mu = grp_sum<V>(F[r] * w);
            float d = F[r] - mu;
            MEAN[r] = mu;
            VAR[r] = grp_sum<V>(w * (d * d));
        }
        float mc[3], vc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            mc[c] = grp_sum<V>(fc[c] * w);
            float d = fc[c] - mc[c];
            vc[c] = grp_sum<V>(w * (d * d));
        }
        // ---- base_fc.0 (105 -> 64) as two 32-output tiles, base_fc.2 (64 -> 32)
        f32x16 H;
        {
            f32x16 h1[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                int rec = MR_BASE0 + nt * 54;
                f32x16 acc = bias_tile(lds, BT_BASE0A + nt, h);
                acc = gemm_frag<16>(lds, rec, lane, MEAN, acc);
                acc = NF_MFMA(lds[(rec + 16) * 64 + lane], h ? mc[1] : mc[0], acc);
                acc = NF_MFMA(lds[(rec + 17) * 64 + lane], h ? 0.f : mc[2], acc);
                acc = gemm_frag<16>(lds, rec + 18, lane, VAR, acc);
                acc = NF_MFMA(lds[(rec + 34) * 64 + lane], h ? vc[1] : vc[0], acc);
                acc = NF_MFMA(lds[(rec + 35) * 64 + lane], h ? 0.f : vc[2], acc);
                acc = gemm_frag<16>(lds, rec + 36, lane, F, acc);
                acc = NF_MFMA(lds[(rec + 52) * 64 + lane], h ? fc[1] : fc[0], acc);
                acc = NF_MFMA(lds[(rec + 53) * 64 + lane], h ? 0.f : fc[2], acc);
                h1[nt] = elu16(acc);
            }
            f32x16 acc = bias_tile(lds, BT_BASE1, h);
            acc = gemm_frag<16>(lds, MR_BASE1, lane, h1[0], acc);
            acc = gemm_frag<16>(lds, MR_BASE1 + 16, lane, h1[1], acc);
            H = elu16(acc);
        }
        // ---- vis_fc on h * w, residual; vis_fc2 on x2 * vis1   (:249-254)
        f32x16 X2;
        float vis2;
        {
            f32x16 t;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = H[r] * w;
            f32x16 v1 = elu16(gemm_frag<16>(lds, MR_VIS0, lane, t, bias_tile(lds, BT_VIS0, h)));
            f32x16 xv = elu16(gemm_frag<16>(lds, MR_VIS1, lane, v1, bias_tile(lds, BT_VIS1, h)));
            float logit = mf_elu(dot_frag16(lds + MS_VIS1L + h * 16, v1) + lds[MS_VIS1L + 32]);
            float vis1 = mf_sigmoid(logit) * mk;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                X2[r] = H[r] + xv[r];
                t[r] = X2[r] * vis1;
            }
            f32x16 u = elu16(gemm_frag<16>(lds, MR_VISB0, lane, t, bias_tile(lds, BT_VISB0, h)));
            float z2 = dot_frag16(lds + MS_VISB1 + h * 16, u) + lds[MS_VISB1 + 32];
            vis2 = mf_sigmoid(z2) * mk;
        }
        const float vsum = grp_sum<V>(vis2) + 1e-8f;
        const float w2 = vis2 / vsum;
        const float wmean = grp_sum<V>(w2) / (float)V;
        const float nval = grp_sum<V>(mk);
        // ---- second pooling -> per-sample record (the lane holding view 0 writes)
        float* out = smp + sample * NF_SMP_STRIDE;
        const bool writer = live && v == 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float mu = grp_sum<V>(X2[r] * w2);
            float d = X2[r] - mu;
            float va = grp_sum<V>(w2 * (d * d));
            if (writer) {
                out[nf_nidx(r, h)] = mu;
                out[32 + nf_nidx(r, h)] = va;
            }
        }
        // ---- colour head: rgb_fc 37 -> 16 -> 8 -> 1, softmax over views, blend of the clean colours  (:268-273)
        float y;
        {
            f32x16 acc = gemm_frag<16>(lds, MR_RGB0, lane, X2, bias_tile(lds, BT_RGB0, h));
            acc = NF_MFMA(lds[(MR_RGB0 + 16) * 64 + lane], h ? rd0 : vis2, acc);
            acc = NF_MFMA(lds[(MR_RGB0 + 17) * 64 + lane], h ? rd2 : rd1, acc);
            acc = NF_MFMA(lds[(MR_RGB0 + 18) * 64 + lane], h ? 0.f : rd3, acc);
            float r1[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) r1[r] = mf_elu(acc[r]);
            y = lds[MS_RGB2 + 8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float* wv = lds + MS_RGB1 + j * 16 + h * 8;
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) t = fmaf(wv[r], r1[r], t);
                t = mf_elu(half_sum(t) + lds[MS_RGB1 + 128 + j]);
                y = fmaf(lds[MS_RGB2 + j], t, y);
            }
        }
        if (mk == 0.f) y = -1e9f;
        float p = expf(y - grp_max<V>(y));
        float beta = p / grp_sum<V>(p);
        float o0 = grp_sum<V>(beta * c0), o1 = grp_sum<V>(beta * c1), o2 = grp_sum<V>(beta * c2);
        if (writer && h == 0) {
            out[64] = wmean;
            out[65] = o0;
            out[66] = o1;
            out[67] = o2;
            out[68] = nval;
            out[69] = vsum;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// kernel B (forward): geometry_fc, ray attention, LayerNorm, density head.  One workgroup per ray, one thread per sample.
// ---------------------------------------------------------------------------------------------------------------
template <int MAXT>
__global__ void __launch_bounds__(MAXT) k_ibr_ray_fwd(const float* __restrict__ B, const float* __restrict__ pos_enc,
                                                      const float* __restrict__ smp, int S, float* __restrict__ raw) {
    HIP_DYNAMIC_SHARED(float, kv)            // [S][8 floats per head-pair...] K then V: 2 * S * 16
    float* Ks = kv;
    float* Vs = kv + (size_t)S * 16;
    const int64_t ray = blockIdx.x;
    const int s = threadIdx.x;
    const bool active = s < S;
    const float* rec = smp + (ray * S + (active ? s : 0)) * NF_SMP_STRIDE;
    float q[16], gpe[16];
    float nval = rec[68];
    if (active) {
        float g1[64];
#pragma unroll
        for (int n = 0; n < 64; ++n) g1[n] = B[nf_lin_b(NF_L_GEO0) + n];
        for (int k = 0; k < 65; ++k) {
            float xk = rec[k];
            const float* wr = B + nf_lin_wt(NF_L_GEO0) + k * 64;
#pragma unroll
            for (int n = 0; n < 64; ++n) g1[n] = fmaf(wr[n], xk, g1[n]);
        }
        float g[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) g[n] = B[nf_lin_b(NF_L_GEO1) + n];
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            float xk = mf_elu(g1[k]);
#pragma unroll
            for (int n = 0; n < 16; ++n) g[n] = fmaf(B[nf_lin_wt(NF_L_GEO1) + k * 16 + n], xk, g[n]);
        }
#pragma unroll
        for (int n = 0; n < 16; ++n) gpe[n] = mf_elu(g[n]) + pos_enc[(size_t)s * 16 + n];
        float kk[16], vv[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) q[n] = kk[n] = vv[n] = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                q[n] = fmaf(B[nf_att_wt(0) + k * 16 + n], gpe[k], q[n]);
                kk[n] = fmaf(B[nf_att_wt(1) + k * 16 + n], gpe[k], kk[n]);
                vv[n] = fmaf(B[nf_att_wt(2) + k * 16 + n], gpe[k], vv[n]);
            }
        }
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            Ks[s * 16 + n] = kk[n];
            Vs[s * 16 + n] = vv[n];
        }
    }
    __syncthreads();
    if (!active) return;
    const bool row_on = nval > 1.f;
    float o[16];
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
        float q0 = q[hd * 4] / 2.f, q1 = q[hd * 4 + 1] / 2.f, q2 = q[hd * 4 + 2] / 2.f, q3 = q[hd * 4 + 3] / 2.f;
        float mx = -1e9f;
        if (row_on) {
            mx = -3.0e38f;
            for (int k = 0; k < S; ++k) {
                const float* kp = Ks + k * 16 + hd * 4;
                mx = fmaxf(mx, fmaf(q3, kp[3], fmaf(q2, kp[2], fmaf(q1, kp[1], q0 * kp[0]))));
            }
        }
        float l = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int k = 0; k < S; ++k) {
            const float* kp = Ks + k * 16 + hd * 4;
            const float* vp = Vs + k * 16 + hd * 4;
            float sc = row_on ? fmaf(q3, kp[3], fmaf(q2, kp[2], fmaf(q1, kp[1], q0 * kp[0]))) : -1e9f;
            float p = expf(sc - mx);
            l += p;
            a0 = fmaf(p, vp[0], a0); a1 = fmaf(p, vp[1], a1); a2 = fmaf(p, vp[2], a2); a3 = fmaf(p, vp[3], a3);
        }
        o[hd * 4] = a0 / l; o[hd * 4 + 1] = a1 / l; o[hd * 4 + 2] = a2 / l; o[hd * 4 + 3] = a3 / l;
    }
    float pre[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) pre[n] = gpe[n];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int n = 0; n < 16; ++n) pre[n] = fmaf(B[nf_att_wt(3) + k * 16 + n], o[k], pre[n]);
    }
    float mu = 0.f;
#pragma unroll
    for (int n = 0; n < 16; ++n) mu += pre[n];
    mu = mu / 16.f;
    float var = 0.f;
#pragma unroll
    for (int n = 0; n < 16; ++n) var += (pre[n] - mu) * (pre[n] - mu);
    float rstd = 1.f / sqrtf(var / 16.f + 1e-6f);
    float og1[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) og1[n] = B[nf_lin_b(NF_L_OG0) + n];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float gat = (pre[k] - mu) * rstd * B[NF_LN_W + k] + B[NF_LN_B + k];
#pragma unroll
        for (int n = 0; n < 16; ++n) og1[n] = fmaf(B[nf_lin_wt(NF_L_OG0) + k * 16 + n], gat, og1[n]);
    }
    float sp = B[nf_lin_b(NF_L_OG1)];
#pragma unroll
    for (int n = 0; n < 16; ++n) sp = fmaf(B[nf_lin_wt(NF_L_OG1) + n], mf_elu(og1[n]), sp);
    float sigma = fmaxf(sp, 0.f);
    if (nval < 1.f) sigma = 0.f;
    float* ro = raw + (ray * S + s) * 4;
    ro[0] = rec[65]; ro[1] = rec[66]; ro[2] = rec[67]; ro[3] = sigma;
}

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
extern "C" int nf_ibrnet_mfma_supported(int n_samples, int n_views) {
    bool v_ok = n_views >= 1 && n_views <= 32 && (n_views & (n_views - 1)) == 0;
    return (v_ok && n_samples >= 1 && n_samples <= NF_IBR_MAX_S) ? 1 : 0;
}

extern "C" int64_t nf_ibrnet_mfma_workspace_floats(int64_t n_rays, int n_samples) {
    return n_rays * n_samples * NF_SMP_STRIDE;
}

template <int V>
static void launch_rows_fwd(const float* wblob, const float* rgb_feat, const float* ray_diff, const float* mask,
                            int64_t n_samples, int aa, float* smp, hipStream_t st) {
    int64_t tiles = (n_samples * V + 31) / 32;
    int64_t blocks = (tiles + 3) / 4;
    if (blocks > 1024) blocks = 1024;     // persistent-ish: 2 workgroups per CU hold the 58 KB weight image each
    hipLaunchKernelGGL(k_ibr_rows_fwd<V>, dim3((unsigned)blocks), dim3(256), NF_MFMA_BLOB_FLOATS * sizeof(float), st, wblob,
                       rgb_feat, ray_diff, mask, n_samples, aa, smp);
}

extern "C" int nf_ibrnet_fwd_mfma(const float* mfma_blob, const float* blob, const float* pos_enc, const float* rgb_feat,
                                  const float* ray_diff, const float* mask, int64_t n_rays, int n_samples, int n_views,
                                  int anti_alias_pooling, float* raw, float* workspace, nf_stream_t stream) {
    NF_REQUIRE(nf_ibrnet_mfma_supported(n_samples, n_views), "nf_ibrnet_fwd_mfma: V must be a power of two <= 32 (got %d)",
               n_views);
    if (n_rays == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int64_t ns = n_rays * n_samples;
    switch (n_views) {
        case 1: launch_rows_fwd<1>(mfma_blob, rgb_feat, ray_diff, mask, ns, anti_alias_pooling, workspace, st); break;
        case 2: launch_rows_fwd<2>(mfma_blob, rgb_feat, ray_diff, mask, ns, anti_alias_pooling, workspace, st); break;
        case 4: launch_rows_fwd<4>(mfma_blob, rgb_feat, ray_diff, mask, ns, anti_alias_pooling, workspace, st); break;
        case 8: launch_rows_fwd<8>(mfma_blob, rgb_feat, ray_diff, mask, ns, anti_alias_pooling, workspace, st); break;
        case 16: launch_rows_fwd<16>(mfma_blob, rgb_feat, ray_diff, mask, ns, anti_alias_pooling, workspace, st); break;
        default: launch_rows_fwd<32>(mfma_blob, rgb_feat, ray_diff, mask, ns, anti_alias_pooling, workspace, st); break;
    }
    NF_LAUNCH_CHECK("nf_ibrnet_fwd_mfma (rows)");
    int threads = ((n_samples + 63) / 64) * 64;
    size_t smem = (size_t)n_samples * 32 * sizeof(float);
    if (threads <= 256)
        hipLaunchKernelGGL(k_ibr_ray_fwd<256>, dim3((unsigned)n_rays), dim3(threads), smem, st, blob, pos_enc, workspace,
                           n_samples, raw);
    else
        hipLaunchKernelGGL(k_ibr_ray_fwd<NF_IBR_MAX_S>, dim3((unsigned)n_rays), dim3(threads), smem, st, blob, pos_enc,
                           workspace, n_samples, raw);
    NF_LAUNCH_CHECK("nf_ibrnet_fwd_mfma (ray)");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// diagnostics: raw MFMA probe, used by the GPU tests to pin the fragment layout the kernels (and the CPU stand-in) assume
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_debug_mfma32(const float* a, const float* b, const float* c, float* d) {
    int lane = threadIdx.x;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = c[lane * 16 + r];
    acc = NF_MFMA(a[lane], b[lane], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) d[lane * 16 + r] = acc[r];
}

extern "C" int nf_debug_mfma32(const float* a, const float* b, const float* c, float* d, nf_stream_t stream) {
    hipLaunchKernelGGL(k_debug_mfma32, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c, d);
    NF_LAUNCH_CHECK("nf_debug_mfma32");
    return 0;
}
