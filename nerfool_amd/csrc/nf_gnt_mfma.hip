// GNT per-ray network (a15), forward on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
// ref: gnt/transformer_network.py:270-309 (GNT.forward, ret_alpha = False, eval mode), :55-89, :93-113, :121-171, :175-202.
//
// One workgroup per ray, one wave per 32 samples (S in {32, 64, 96, 128}); sample = MFMA column (lane & 31), so every
// 64-wide per-sample vector is TWO accumulator tiles: register r of tile t in lane (m, h) holds feature 32 t + n(r, h),
// n(r, h) = (r & 3) + 8 (r >> 2) + 4 h -- which is exactly the B operand of the next layer's k-step r, so activations are
// chained through registers (same scheme as nf_ibrnet_mfma.hip).  Weights are host-packed into "records" of 64 floats in
// consumption order and streamed from L2 (a layer is 471 KB: no LDS image), one coalesced 256-byte load per MFMA.
//   * view transformer: the V source views are walked serially per wave with an online softmax (running max / sum /
//     weighted sum per channel in registers), any V; X_v comes back from the workspace the stem wrote.
//   * ray transformer: K and V^T of the ray are laid out in LDS as MFMA A-operand records by the lanes that produced them;
//     scores^T = K Q^T (keys on rows => the softmax over keys is in-lane + one cross-half shuffle), O^T = V^T P^T.
// One wave per SIMD (512 rays x 2 waves fill the chip exactly once): nothing hides a memory latency but the code itself.  So
//   * the layer's small records and bias / LayerNorm tables live in LDS, q and v + pos of the view loop are parked in LDS,
//   * the inputs of view v + 1 are fetched in front of the softmax arithmetic of view v,
//   * the weight stream's chunk loads are pinned (GM_PIN_CHAIN) where they are written -- the scheduler otherwise sinks them to
//     their use -- and the stream pointer keeps its address space (no flat_load: gm_opaque_zero).
// With save != 0 the workspace (slots of nf_gnt.h) receives what k_gnt_bwd_mfma below reads -- NOT what the shape-generic
// nf_gnt_bwd reads: the view softmax is saved as masked logits + running maximum + reciprocal sum (no normalisation pass over
// the views), ReLU layers as sign words (gm_sign_word), plus the view-attention output u.
#include "nf_gnt.h"

#include <string.h>

typedef float g16 __attribute__((ext_vector_type(16)));
#define GM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__host__ __device__ constexpr int gm_nidx(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- record layout (units: records of 64 floats), per layer: the few records of the tiny GEMMs first, then the STREAM --
//      every 64-input GEMM tile is two chunks of 16 records, laid out in exactly the order the kernel consumes them, so the
//      kernel can run a two-chunk-deep software pipeline over one monotonically advancing pointer (GmW below).
enum {
    MG_POS0 = 0, MG_POS2 = 2, MG_ATT2 = 10, MG_STREAM = 32,
    ST_VQ = 0, ST_VK = 64, ST_VV = 128, ST_ATT0 = 192, ST_VOUT = 224,
    ST_VFF = 288,            // 8 x [fc1 tile j: 2 chunks][fc2 k-tile j: out tile 0, out tile 1]
    ST_Q0 = 800,             // 2 x [cur: 2 chunks | positional: 4 chunks (63 k-steps + 1 zero record)]
    ST_Q2 = 992, ST_RQ = 1056, ST_RK = 1120, ST_RV = 1184, ST_ROUT = 1248, ST_RFF = 1312, ST_END = 1824,
    MG_LAYER_RECORDS = MG_STREAM + ST_END
};
#define GM_CHUNK 1024        // floats per chunk (16 records)
#define GM_BLOB_PAD 4096     // the pipeline reads up to two chunks past the last record it needs
// ---- per-layer tables behind the records (floats): bias tiles [h][16], LayerNorm vectors in fragment order [t][h][16]
enum {
    MB_POS0 = 0, MB_POS2 = 32, MB_ATT0 = 96, MB_ATT2 = 128, MB_VOUT = 192, MB_VFF1 = 256, MB_VFF2 = 512, MB_Q0 = 576, MB_Q2 = 640,
    MB_ROUT = 704, MB_RFF1 = 768, MB_RFF2 = 1024, MB_LN = 1088,       // 4 x (weight 64, bias 64)
    MB_FLOATS = MB_LN + 512
};
static constexpr int64_t GM_LAYER_FLOATS = (int64_t)MG_LAYER_RECORDS * 64 + MB_FLOATS;
// stem: rgbfeat_fc.0 (35 -> 64, k order 2 s + h, 18 steps per tile), rgbfeat_fc.2 (64 -> 64); then 2 + 2 bias tiles
enum { MS_L1 = 0, MS_L2 = 36, MS_RECORDS = 100, MS_B1 = MS_RECORDS * 64, MS_B2 = MS_B1 + 64, GM_STEM_FLOATS = MS_B2 + 64 };
// final: norm weight, bias (fragment order), rgb_fc weight [3] x fragment order, bias 3 (+1 pad)
enum { MF_LNW = 0, MF_LNB = 64, MF_W = 128, MF_B = 320, GM_FINAL_FLOATS = 384 };      // (whole records: every stream starts on one)

NF_HD int64_t gm_layer_base(int i) { return GM_STEM_FLOATS + (int64_t)i * GM_LAYER_FLOATS; }

// ---- backward section (behind the forward section): transposed records (dX^T = W^T dY^T) in the order the backward
//      consumes them, layers stored in forward order but walked from the last to the first
enum {
    BG_ATT0T = 0, BG_STREAM = 16,      // 8 records of attn_fc.0^T (64 <- 8), then the stream (units: records)
    BS_RFF = 0, BS_ROUT = 512, BS_RQ = 576, BS_RK = 640, BS_RV = 704, BS_Q2 = 768, BS_Q0 = 832, BS_VFF = 896, BS_VOUT = 1408,
    BS_VIEW = 1472,                    // attn_fc.2^T (8 <- 64: one tile), v_fc^T, k_fc^T = 32 + 64 + 64 records
    BS_VQ = 1632, BS_END = 1696,
    BG_LAYER_RECORDS = BG_STREAM + BS_END,
    BSTEM_RECORDS = 128                // rgbfeat_fc.2^T (64 <- 64), rgbfeat_fc.0^T (35 <- 64: two tiles)
};
static constexpr int64_t GM_BWD_LAYER_FLOATS = (int64_t)BG_LAYER_RECORDS * 64;
NF_HD int64_t gm_fwd_floats(int depth) { return GM_STEM_FLOATS + (int64_t)depth * GM_LAYER_FLOATS + GM_FINAL_FLOATS + GM_BLOB_PAD; }
NF_HD int64_t gm_bwd_layer_base(int depth, int i) { return gm_fwd_floats(depth) + (int64_t)i * GM_BWD_LAYER_FLOATS; }
NF_HD int64_t gm_bwd_stem_base(int depth) { return gm_bwd_layer_base(depth, depth); }

// ---- bf16x3 image of the STREAMED records (behind the fp32 blob): the streamed GEMMs run on v_mfma_f32_32x32x16_bf16 with every fp32
//      operand as three bf16 parts (six cross terms, fp32-grade; DESIGN "fp32-grade products on the bf16 matrix cores").  A k-block of the
//      bf16 instruction = 8 consecutive fp32 k-steps: lane (i, h) holds the eight weights it held in records Y .. Y + 7, so the B operand
//      is the lane's OWN eight activations of those k-steps.  The block of records [Y, Y + 8) lives at image offset 96 Y floats as
//      [part 3][lane 64][8 bf16] (768 floats): any stream pointer p of the fp32 blob maps to image + (p - blob) * 3 / 2.
NF_HD int64_t gm_fp32_floats(int depth) { return gm_bwd_stem_base(depth) + BSTEM_RECORDS * 64 + GM_BLOB_PAD; }
#define GM_X3_CHUNK 1536     // floats per chunk of the image (two k-blocks of three 1 KB parts)

extern "C" int64_t nf_gnt_mfma_blob_floats(int depth) { return gm_fp32_floats(depth) + gm_fp32_floats(depth) / 2 * 3; }

extern "C" int nf_gnt_mfma_supported(int n_samples, int n_views) {
    return n_samples >= 32 && n_samples <= 128 && n_samples % 32 == 0 && n_views >= 1 && n_views <= 64;
}

// ---------------------------------------------------------------------------------------------------------------
// host: natural blob (nf_gnt.h) -> MFMA-order blob
// ---------------------------------------------------------------------------------------------------------------
// record: lane (i, h) holds W[n = 32 nt + i][k = h ? k1 : k0]   (W is [N][K] row-major; out-of-range -> 0)
static void gm_record(float*& rec, const float* W, int N, int K, int nt, int k0, int k1) {
    for (int lane = 0; lane < 64; ++lane) {
        int n = nt * 32 + (lane & 31), k = (lane >> 5) ? k1 : k0;
        rec[lane] = (n < N && k >= 0 && k < K) ? W[(size_t)n * K + k] : 0.f;
    }
    rec += 64;
}
// k-steps over fragment-ordered inputs kbase + n(r, h), r < nsteps
static void gm_frag(float*& rec, const float* W, int N, int K, int nt, int kbase, int nsteps) {
    for (int r = 0; r < nsteps; ++r) gm_record(rec, W, N, K, nt, kbase + gm_nidx(r, 0), kbase + gm_nidx(r, 1));
}
// k-steps over memory-ordered inputs kbase + 2 s + h, s < nsteps
static void gm_seq(float*& rec, const float* W, int N, int K, int nt, int kbase, int nsteps) {
    for (int s = 0; s < nsteps; ++s) gm_record(rec, W, N, K, nt, kbase + 2 * s, kbase + 2 * s + 1);
}
static void gm_lin64(float*& rec, const float* W, int N, int K) {      // N outputs (tiles of 32) x K = 64 inputs
    for (int nt = 0; nt < (N + 31) / 32; ++nt) {
        gm_frag(rec, W, N, K, nt, 0, 16);
        gm_frag(rec, W, N, K, nt, 32, 16);
    }
}
static void gm_vec_tiles(float* dst, const float* b, int N) {          // fragment order [t][h][16]
    for (int t = 0; t < (N + 31) / 32; ++t)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) {
                int n = 32 * t + gm_nidx(r, h);
                dst[t * 32 + h * 16 + r] = n < N ? b[n] : 0.f;
            }
}

extern "C" int nf_gnt_pack_mfma(int depth, const float* nat, float* out) {
    const int64_t total = nf_gnt_mfma_blob_floats(depth);
    for (int64_t i = 0; i < total; ++i) out[i] = 0.f;
    {   // stem
        float* rec = out;
        const float* W1 = nat + 35 * 64;                        // native [64][35]
        for (int nt = 0; nt < 2; ++nt) gm_seq(rec, W1, 64, 35, nt, 0, 18);
        gm_lin64(rec, nat + GNT_STEM1 + 64 * 64, 64, 64);
        if (rec - out != MS_RECORDS * 64) return 1;
        gm_vec_tiles(out + MS_B1, nat + 2 * 35 * 64, 64);
        gm_vec_tiles(out + MS_B2, nat + GNT_STEM1 + 2 * 64 * 64, 64);
    }
    for (int i = 0; i < depth; ++i) {
        const float* L = nat + gnt_layer_base(i);
        float* base = out + gm_layer_base(i);
        float* rec = base;
        gm_seq(rec, L + gnt_w(GV_POS0), 8, 4, 0, 0, 2);
        for (int nt = 0; nt < 2; ++nt) gm_frag(rec, L + gnt_w(GV_POS2), 64, 8, nt, 0, 4);
        for (int nt = 0; nt < 2; ++nt) gm_frag(rec, L + gnt_w(GV_ATT2), 64, 8, nt, 0, 4);
        if (rec - base != 18 * 64) return 2;
        rec = base + MG_STREAM * 64;
        float* st0 = rec;
        gm_lin64(rec, L + gnt_w(GV_Q), 64, 64);
        gm_lin64(rec, L + gnt_w(GV_K), 64, 64);
        gm_lin64(rec, L + gnt_w(GV_V), 64, 64);
        gm_lin64(rec, L + gnt_w(GV_ATT0), 8, 64);
        gm_lin64(rec, L + gnt_w(GV_OUT), 64, 64);
        for (int pass = 0; pass < 2; ++pass) {                  // view FF, then ray FF: hidden tile j chained into fc2
            const int l1 = pass == 0 ? GV_FF1 : GR_FF1, l2 = pass == 0 ? GV_FF2 : GR_FF2;
            if (rec - st0 != (pass == 0 ? ST_VFF : ST_RFF) * 64) return 3;
            for (int j = 0; j < 8; ++j) {
                gm_frag(rec, L + gnt_w(l1), 256, 64, j, 0, 16);
                gm_frag(rec, L + gnt_w(l1), 256, 64, j, 32, 16);
                for (int nt = 0; nt < 2; ++nt) gm_frag(rec, L + gnt_w(l2), 64, 256, nt, 32 * j, 16);
            }
            if (pass == 0) {
                if ((i & 1) == 0) {
                    const float* W0 = L + gnt_w(GQ_0);          // [64][190]: 64 current features, 126 positional
                    for (int nt = 0; nt < 2; ++nt) {
                        gm_frag(rec, W0, 64, 190, nt, 0, 16);
                        gm_frag(rec, W0, 64, 190, nt, 32, 16);
                        gm_seq(rec, W0, 64, 190, nt, 64, 64);   // step 63 reads inputs 190, 191: out of range -> zero record
                    }
                    gm_lin64(rec, L + gnt_w(GQ_2), 64, 64);
                } else {
                    rec += (192 + 64) * 64;
                }
                if (rec - st0 != ST_RQ * 64) return 4;
                gm_lin64(rec, L + gnt_w(GR_Q), 64, 64);
                gm_lin64(rec, L + gnt_w(GR_K), 64, 64);
                gm_lin64(rec, L + gnt_w(GR_V), 64, 64);
                gm_lin64(rec, L + gnt_w(GR_OUT), 64, 64);
            }
        }
        if (rec - st0 != ST_END * 64) return 5;
        float* tb = base + (int64_t)MG_LAYER_RECORDS * 64;
        gm_vec_tiles(tb + MB_POS0, L + gnt_b(GV_POS0), 8);
        gm_vec_tiles(tb + MB_POS2, L + gnt_b(GV_POS2), 64);
        gm_vec_tiles(tb + MB_ATT0, L + gnt_b(GV_ATT0), 8);
        gm_vec_tiles(tb + MB_ATT2, L + gnt_b(GV_ATT2), 64);
        gm_vec_tiles(tb + MB_VOUT, L + gnt_b(GV_OUT), 64);
        gm_vec_tiles(tb + MB_VFF1, L + gnt_b(GV_FF1), 256);
        gm_vec_tiles(tb + MB_VFF2, L + gnt_b(GV_FF2), 64);
        if ((i & 1) == 0) {
            gm_vec_tiles(tb + MB_Q0, L + gnt_b(GQ_0), 64);
            gm_vec_tiles(tb + MB_Q2, L + gnt_b(GQ_2), 64);
        }
        gm_vec_tiles(tb + MB_ROUT, L + gnt_b(GR_OUT), 64);
        gm_vec_tiles(tb + MB_RFF1, L + gnt_b(GR_FF1), 256);
        gm_vec_tiles(tb + MB_RFF2, L + gnt_b(GR_FF2), 64);
        for (int j = 0; j < 4; ++j) {
            gm_vec_tiles(tb + MB_LN + j * 128, L + gnt_ln_w(j), 64);
            gm_vec_tiles(tb + MB_LN + j * 128 + 64, L + gnt_ln_b(j), 64);
        }
    }
    {   // final
        const float* F = nat + gnt_layer_base(depth);
        float* fo = out + gm_layer_base(depth);
        gm_vec_tiles(fo + MF_LNW, F, 64);
        gm_vec_tiles(fo + MF_LNB, F + 64, 64);
        for (int c = 0; c < 3; ++c) gm_vec_tiles(fo + MF_W + c * 64, F + 320 + c * 64, 64);      // native W [3][64]
        for (int c = 0; c < 3; ++c) fo[MF_B + c] = F[512 + c];
    }
    // ================= backward section: M = W^T, i.e. the blob's [in][out] arrays =================
    for (int i = 0; i < depth; ++i) {
        const float* L = nat + gnt_layer_base(i);
        float* base = out + gm_bwd_layer_base(depth, i);
        float* rec = base;
        for (int nt = 0; nt < 2; ++nt) gm_frag(rec, L + gnt_wt(GV_ATT0), 64, 8, nt, 0, 4);
        rec = base + BG_STREAM * 64;
        float* st0 = rec;
        for (int pass = 0; pass < 2; ++pass) {      // ray FF first (the backward walks a layer from its end), then view FF
            const int l1 = pass == 0 ? GR_FF1 : GV_FF1, l2 = pass == 0 ? GR_FF2 : GV_FF2;
            if (rec - st0 != (pass == 0 ? BS_RFF : BS_VFF) * 64) return 6;
            for (int j = 0; j < 8; ++j) {
                gm_frag(rec, L + gnt_wt(l2), 256, 64, j, 0, 16);       // d hidden tile j  <- d out (64)
                gm_frag(rec, L + gnt_wt(l2), 256, 64, j, 32, 16);
                for (int nt = 0; nt < 2; ++nt) gm_frag(rec, L + gnt_wt(l1), 64, 256, nt, 32 * j, 16);      // d y += W1^T d hidden_j
            }
            if (pass == 0) {
                gm_lin64(rec, L + gnt_wt(GR_OUT), 64, 64);
                gm_lin64(rec, L + gnt_wt(GR_Q), 64, 64);
                gm_lin64(rec, L + gnt_wt(GR_K), 64, 64);
                gm_lin64(rec, L + gnt_wt(GR_V), 64, 64);
                if ((i & 1) == 0) {
                    gm_lin64(rec, L + gnt_wt(GQ_2), 64, 64);
                    gm_lin64(rec, L + gnt_wt(GQ_0), 64, 64);           // rows 0..63 of the [190][64] array: d of the 64 current features
                } else {
                    rec += 128 * 64;
                }
            } else {
                gm_lin64(rec, L + gnt_wt(GV_OUT), 64, 64);
                gm_lin64(rec, L + gnt_wt(GV_ATT2), 8, 64);
                gm_lin64(rec, L + gnt_wt(GV_V), 64, 64);
                gm_lin64(rec, L + gnt_wt(GV_K), 64, 64);
                gm_lin64(rec, L + gnt_wt(GV_Q), 64, 64);
            }
        }
        if (rec - st0 != BS_END * 64) return 7;
    }
    {
        float* rec = out + gm_bwd_stem_base(depth);
        gm_lin64(rec, nat + GNT_STEM1, 64, 64);                        // rgbfeat_fc.2 [in][out]
        gm_lin64(rec, nat, 35, 64);                                    // rgbfeat_fc.0 [in = 35][out = 64]: two output tiles
        if (rec - (out + gm_bwd_stem_base(depth)) != BSTEM_RECORDS * 64) return 8;
    }
    // the bf16x3 image of the streamed regions
    {
        const int64_t f32 = gm_fp32_floats(depth);
        float* img = out + f32;
        memset(img, 0, sizeof(float) * (size_t)(f32 / 2 * 3));
        auto region = [&](int64_t base_floats, int records) {
            for (int y = 0; y < records; y += 8) {
                const float* src = out + base_floats + (int64_t)y * 64;
                uint16_t* dst = reinterpret_cast<uint16_t*>(img + (base_floats + (int64_t)y * 64) / 2 * 3);
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        float rem = src[j * 64 + lane];
                        for (int part = 0; part < 3; ++part) {
                            uint32_t u;
                            memcpy(&u, &rem, 4);
                            const uint16_t q = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);        // round to nearest even (finite weights)
                            dst[(size_t)part * 512 + lane * 8 + j] = q;
                            const uint32_t up = (uint32_t)q << 16;
                            float f;
                            memcpy(&f, &up, 4);
                            rem -= f;
                        }
                    }
            }
        };
        for (int i = 0; i < depth; ++i) {
            region(gm_layer_base(i) + MG_STREAM * 64, ST_END);
            region(gm_bwd_layer_base(depth, i) + BG_STREAM * 64, BS_END);
        }
        region(gm_bwd_stem_base(depth), BSTEM_RECORDS);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------------------------
struct V64 { g16 t[2]; };       // a 64-wide per-sample vector in fragment order

__device__ __forceinline__ g16 gm_zero() {
    g16 a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
    return a;
}
__device__ __forceinline__ g16 gm_tile(const float* __restrict__ tb, int h) {      // one [h][16] tile of a table (16-byte aligned)
    g16 a;
    const float4* b = reinterpret_cast<const float4*>(tb + h * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 x = b[q];
        a[4 * q] = x.x; a[4 * q + 1] = x.y; a[4 * q + 2] = x.z; a[4 * q + 3] = x.w;
    }
    return a;
}
// ---- LDS parking of a wave's own V64 (lane-private values, [t][q][lane][4]: 16-byte accesses of consecutive lanes)
#define GM_PARK_FLOATS 2048
__device__ __forceinline__ void gm_park4(float* p, int lane, int t, int q, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p + ((t * 4 + q) * 64 + lane) * 4) = make_float4(a, b, c, d);
}
__device__ __forceinline__ float4 gm_parked4(const float* p, int lane, int t, int q) {
    return *reinterpret_cast<const float4*>(p + ((t * 4 + q) * 64 + lane) * 4);
}
__device__ __forceinline__ void gm_park(float* p, int lane, const V64& x) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) gm_park4(p, lane, t, q, x.t[t][4 * q], x.t[t][4 * q + 1], x.t[t][4 * q + 2], x.t[t][4 * q + 3]);
}
// the two lane halves swap through v_permlane32_swap (nf_common.h): no LDS round trip
__device__ __forceinline__ float gm_half_sum(float x) { return nf_half_sum(x); }
__device__ __forceinline__ float gm_half_max(float x) { return nf_half_max(x); }

// a wave-uniform zero the optimiser cannot see through.  Added to a pointer inside the view loop it keeps the loop-invariant
// record reads -- 178 registers' worth -- from being hoisted out of the loop and spilled; unlike passing the POINTER through an
// asm statement it keeps the pointer's address space (a laundered pointer is accessed with flat_load, whose waits are
// vmcnt(0) lgkmcnt(0): no load stays in flight across one)
__device__ __forceinline__ int gm_opaque_zero() {
    int z = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(z));
#else
    asm volatile("" : "+r"(z));
#endif
    return z;
}
// acc += sum over NSTEPS k-steps: records at rec (L2 or LDS), B operand = x[r]
template <int NSTEPS>
__device__ __forceinline__ g16 gm_gemm(const float* __restrict__ rec, int lane, const g16& x, g16 acc) {
#pragma unroll
    for (int r = 0; r < NSTEPS; ++r) acc = GM_MFMA(rec[r * 64 + lane], x[r], acc);
    return acc;
}
// one 32-output tile of a 64-input linear: records [k-tile 0: 16][k-tile 1: 16]
__device__ __forceinline__ g16 gm_lin_tile(const float* __restrict__ rec, int lane, const V64& x, g16 acc) {
    acc = gm_gemm<16>(rec, lane, x.t[0], acc);
    return gm_gemm<16>(rec + 16 * 64, lane, x.t[1], acc);
}
// 64 -> 64 linear; bias tiles at tb (nullptr: none)
__device__ __forceinline__ V64 gm_lin64(const float* __restrict__ rec, const float* __restrict__ tb, int lane, int h, const V64& x) {
    V64 y;
#pragma unroll
    for (int t = 0; t < 2; ++t) y.t[t] = gm_lin_tile(rec + t * 32 * 64, lane, x, tb ? gm_tile(tb + t * 32, h) : gm_zero());
    return y;
}

// ---- weight stream: two chunks (2 x 16 records) are always in flight; consuming a chunk immediately issues the loads of the
//      chunk two ahead (w.p).  With one wave per SIMD (512 rays x 2 waves = one wave per SIMD of the chip) nothing else hides
//      the ~1 us L2 latency of the weight loads.  Non-contiguous transitions set w.p before the last TWO chunks of a segment.
typedef unsigned gm_u4 __attribute__((ext_vector_type(4)));
typedef __bf16 gm_bf8 __attribute__((ext_vector_type(8)));
struct GmChunk { gm_u4 q[6]; };      // [k-block 2][part 3]: one 32 x 32 weight block on bf16x3 operands
struct GmW { GmChunk a, b; const float* p; const float* blob; const float* img; };      // p walks the fp32 blob's record space
__device__ __forceinline__ GmChunk gm_ld16(const GmW& w, const float* __restrict__ p, int lane) {
    const float* __restrict__ s = w.img + ((p - w.blob) >> 1) * 3 + 4 * lane;      // (uniform arithmetic; p - blob is a multiple of 64)
    GmChunk c;
#pragma unroll
    for (int i = 0; i < 6; ++i) c.q[i] = *reinterpret_cast<const gm_u4*>(s + i * 256);
    return c;
}
__device__ __forceinline__ void gm_w_start(GmW& w, const float* blob, int depth, const float* p, int lane) {
    w.blob = blob;
    w.img = blob + gm_fp32_floats(depth);
    w.a = gm_ld16(w, p, lane);
    w.b = gm_ld16(w, p + GM_CHUNK, lane);
    w.p = p + 2 * GM_CHUNK;
}
// two fp32 values -> their bf16 roundings (packed) and the exact remainders
__device__ __forceinline__ unsigned gm_split_pair(float& x0, float& x1) { return nf_split_pair_bf16(x0, x1); }
// acc += (32 x 32 block) x (the lane's 16 activations of the k-tile): per k-block the six cross terms of order <= 2^-16
__device__ __forceinline__ g16 gm_mfma16(const GmChunk& wv, const g16& x, g16 acc) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        gm_u4 bv[3];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v0 = x[8 * blk + 2 * q], v1 = x[8 * blk + 2 * q + 1];
#pragma unroll
            for (int part = 0; part < 3; ++part) bv[part][q] = gm_split_pair(v0, v1);
        }
#define GM_PROD(pa, pb) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gm_bf8, wv.q[3 * blk + pa]), __builtin_bit_cast(gm_bf8, bv[pb]), acc, 0, 0, 0)
        GM_PROD(0, 0);
        GM_PROD(0, 1);
        GM_PROD(1, 0);
        GM_PROD(0, 2);
        GM_PROD(2, 0);
        GM_PROD(1, 1);
#undef GM_PROD
    }
    return acc;
}
// memory operations stay on their side of this point: under register pressure the instruction scheduler otherwise sinks the
// loads of a chunk down to the MFMAs that consume them (load, s_waitcnt vmcnt(0), MFMA -- the whole L2 latency per record)
#define GM_PIN() asm volatile("" ::: "memory")
// ... and the MFMAs of the NEXT chunk stay behind it (they start with `next0`, the first part of the other buffer): hoisted
// above the pin they would run right behind the loads issued for the chunk after them, which shortens the distance between a
// load and its use from two chunks to a few records
#if defined(__HIP_DEVICE_COMPILE__)
#define GM_PIN_CHAIN(next0) asm volatile("" : "+v"(next0) : : "memory")
#else
#define GM_PIN_CHAIN(next0) asm volatile("" ::: "memory")
#endif
__device__ __forceinline__ g16 gm_take_a(GmW& w, int lane, const g16& x, g16 acc) {
    acc = gm_mfma16(w.a, x, acc);
    w.a = gm_ld16(w, w.p, lane);
    GM_PIN_CHAIN(w.b.q[0]);
    w.p += GM_CHUNK;
    return acc;
}
__device__ __forceinline__ g16 gm_take_b(GmW& w, int lane, const g16& x, g16 acc) {
    acc = gm_mfma16(w.b, x, acc);
    w.b = gm_ld16(w, w.p, lane);
    GM_PIN_CHAIN(w.a.q[0]);
    w.p += GM_CHUNK;
    return acc;
}
// one 32-output tile of a 64-input linear = chunk a (k-tile 0) then chunk b (k-tile 1)
__device__ __forceinline__ g16 gm_tile_s(GmW& w, int lane, const V64& x, g16 acc) {
    acc = gm_take_a(w, lane, x.t[0], acc);
    return gm_take_b(w, lane, x.t[1], acc);
}
__device__ __forceinline__ V64 gm_lin64_s(GmW& w, const float* __restrict__ tb, int lane, int h, const V64& x) {
    V64 y;
#pragma unroll
    for (int t = 0; t < 2; ++t) y.t[t] = gm_tile_s(w, lane, x, tb ? gm_tile(tb + t * 32, h) : gm_zero());
    return y;
}

// Workspace addressing: feature 32 t + n(r, h) = [32 t + n(r, 0)] + 4 h, so every access is a WAVE-UNIFORM base (scalar
// registers, scalar arithmetic) plus ONE per-lane 32-bit offset computed once, 4 h S + s, plus a compile-time multiple of S:
// the per-(sample, view) slots are laid out [view][slot][sample], so one base per view serves all of a view's slots.  (Per-element 64-bit vector address arithmetic, hoisted out of the view loop
// by the optimiser, was worth ~450 spilled registers.)
struct GmCtx {
    float* ws_row;      // [V][slot][S], uniform
    float* ws_smp;      // [slot][S], uniform
    int64_t view_floats;        // floats of one view's slots = row_floats * S
    int S, V, s, h;
    unsigned row_lane, smp_lane;
};
__device__ __forceinline__ float* gm_smp_at(const GmCtx& c, int slot_feature) { return c.ws_smp + (size_t)slot_feature * c.S; }
__device__ __forceinline__ float* gm_row_at(const GmCtx& c, int slot_feature, int v) {
    return c.ws_row + (size_t)v * c.view_floats + (size_t)slot_feature * c.S;
}
__device__ __forceinline__ void gm_store_smp(const GmCtx& c, int slot, const V64& x) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) gm_smp_at(c, slot + 32 * t + gm_nidx(r, 0))[c.smp_lane] = x.t[t][r];
}
// What the forward saves for the backward is not read again by this kernel: non-temporal stores (streamed past the caches the weight
// stream and the X_v re-reads live in; measured on the two per-view stores alone: forward 1.76 -> 1.69 ms)
#define GM_SAVE(ptr, val) __builtin_nontemporal_store((val), (ptr))
__device__ __forceinline__ void gm_save_smp(const GmCtx& c, int slot, const V64& x) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) GM_SAVE(gm_smp_at(c, slot + 32 * t + gm_nidx(r, 0)) + c.smp_lane, x.t[t][r]);
}
__device__ __forceinline__ V64 gm_load_smp(const GmCtx& c, int slot) {
    V64 x;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) x.t[t][r] = gm_smp_at(c, slot + 32 * t + gm_nidx(r, 0))[c.smp_lane];
    return x;
}
__device__ __forceinline__ void gm_store_row(const GmCtx& c, int slot, int v, const V64& x) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) gm_row_at(c, slot + 32 * t + gm_nidx(r, 0), v)[c.row_lane] = x.t[t][r];
}
__device__ __forceinline__ V64 gm_load_row(const GmCtx& c, int slot, int v) {
    V64 x;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) x.t[t][r] = gm_row_at(c, slot + 32 * t + gm_nidx(r, 0), v)[c.row_lane];
    return x;
}

// LayerNorm over the 64 features of the lane's sample: y = xhat * w + b; xhat -> xh (also saved when xh_slot >= 0)
__device__ __forceinline__ V64 gm_layernorm(const GmCtx& c, const V64& x, const float* __restrict__ w, const float* __restrict__ b,
                                            float eps, bool save, int xh_slot, int rstd_slot, V64* xh_out = nullptr) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += x.t[t][r];
    const float mu = gm_half_sum(s) / 64.f;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) q += (x.t[t][r] - mu) * (x.t[t][r] - mu);
    const float rstd = 1.f / sqrtf(gm_half_sum(q) / 64.f + eps);
    V64 xh, y;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            xh.t[t][r] = (x.t[t][r] - mu) * rstd;
            y.t[t][r] = xh.t[t][r] * w[t * 32 + c.h * 16 + r] + b[t * 32 + c.h * 16 + r];
        }
    if (save) {
        gm_save_smp(c, xh_slot, xh);
        if (c.h == 0) GM_SAVE(gm_smp_at(c, rstd_slot) + c.s, rstd);
    }
    if (xh_out) *xh_out = xh;
    return y;
}

// What the backward needs of a ReLU layer is the sign of its 16 outputs per lane and tile: ONE word instead of 16 floats
// (the feed-forward hidden layers alone were 512 of the 1168 floats saved per sample and layer).  Word j of lane half h lands in
// slot gm_sign_slot(j) + 4 h (the half offset is part of the lane offset), so words 0..7 occupy 16 slots.
__device__ __forceinline__ float gm_sign_word(const g16& f) {
    unsigned bits = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) bits |= (f[r] > 0.f ? 1u : 0u) << r;
    return __uint_as_float(bits);
}
__host__ __device__ constexpr int gm_sign_slot(int j) { return (j & 3) + 8 * (j >> 2); }

// ---- TRAINING mode (round 6): the eight Dropout sites of a layer on the matrix-core kernels, masks from the counter-based generator
//      of nf_gnt.h (gnt_keep: the same masks as k_gnt_fwd / k_gnt_bwd, the oracle and the fixture's reference run).  TRAIN is a template
//      flag: the eval-mode kernels carry none of it.  Element (t, r) of a fragment-ordered V64 of the lane's sample is channel
//      32 t + n(r, h) of the [.., 64] tensor the reference hands to nn.Dropout.
template <bool TRAIN>
__device__ __forceinline__ float gm_keep(const GntDrop& dr, unsigned site, unsigned long long idx) {
    return TRAIN ? gnt_keep(dr, site, idx) : 1.f;
}
// x .* keep(site, base + channel) over the 64 channels of the lane's sample
template <bool TRAIN>
__device__ __forceinline__ V64 gm_drop64(const GntDrop& dr, unsigned site, unsigned long long base, int h, const V64& x) {
    if (!TRAIN) return x;
    V64 y;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) y.t[t][r] = x.t[t][r] * gnt_keep(dr, site, base + (unsigned)(32 * t + gm_nidx(r, h)));
    return y;
}

// feed-forward 64 -> 256 (ReLU) -> 64, hidden tile j chained straight into fc2; returns FF(y) (bias included).
// jump (nullable): where the stream continues behind this block when that is not the next record in memory.
// TRAIN: the hidden units pass Dropout site `site` (index smp_idx * 256 + unit); the sign word then marks the units that survived BOTH
// the ReLU and the Dropout -- what the backward scales by 1 / (1 - p).  (The output's site + 1 is the caller's: gm_drop64.)
template <bool TRAIN>
__device__ __forceinline__ V64 gm_ff(const GmCtx& c, GmW& w, const float* __restrict__ b1, const float* __restrict__ b2, int lane,
                                     const V64& y, bool save, int f_slot, const float* jump, const GntDrop& dr, unsigned site,
                                     unsigned long long smp_idx) {
    V64 o;
    o.t[0] = gm_tile(b2, c.h);
    o.t[1] = gm_tile(b2 + 32, c.h);
#pragma unroll 1
    for (int j = 0; j < 8; ++j) {
        g16 f = gm_tile_s(w, lane, y, gm_tile(b1 + j * 32, c.h));
        if (j == 7 && jump) w.p = jump;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            f[r] = fmaxf(f[r], 0.f);
            if (TRAIN) f[r] *= gnt_keep(dr, site, smp_idx * 256 + (unsigned)(32 * j + gm_nidx(r, c.h)));
        }
        if (save) GM_SAVE(gm_smp_at(c, f_slot + gm_sign_slot(j)) + c.smp_lane, gm_sign_word(f));      // the backward needs relu' only
        o.t[0] = gm_take_a(w, lane, f, o.t[0]);
        o.t[1] = gm_take_b(w, lane, f, o.t[1]);
    }
    return o;
}

// LDS image of a ray's keys / values as MFMA A-operand records
#define GM_VT_STRIDE 65      // odd record stride: the transposing V^T writes of a wave spread over the banks
// In front of it: the layer's small records (pos_fc, attn_fc.2) and its bias / LayerNorm tables -- read inside the view loop, where a
// load from L2 would sit fully exposed (one wave per SIMD).  The K / V^T image is dead while the views are walked: each wave
// parks ITS q and v + pos there (2 x GM_PARK_FLOATS).
#define GM_TAB_REC (18 * 64)
#define GM_TAB_FLOATS (GM_TAB_REC + MB_FLOATS)
template <int NW> struct GmLds {
    static constexpr int K_FLOATS = 4 * NW * 8 * 64;
    static constexpr int VT_FLOATS = 4 * NW * 16 * GM_VT_STRIDE;
    static constexpr int RED_FLOATS = NW * 4;
    static constexpr int FLOATS = GM_TAB_FLOATS + K_FLOATS + VT_FLOATS + RED_FLOATS;
    static_assert(K_FLOATS + VT_FLOATS >= NW * 2 * GM_PARK_FLOATS, "the parked q / v + pos alias the K / V^T image");
};


// SAVE (write what the backward reads) is a template flag, not an argument: inside the view loop the count of stores that follow the
// next view's loads then is a compile-time fact, and the wait in front of the loop's back edge is vmcnt(32) instead of a vmcnt(0) that
// sits out the acknowledgements of the stores
template <int NW, bool SAVE, bool TRAIN>
__global__ void __launch_bounds__(64 * NW) k_gnt_fwd_mfma(const float* __restrict__ wb, const float* __restrict__ rgb_feat_all,
                                                          const float* __restrict__ ray_diff_all, const float* __restrict__ mask_all,
                                                          const float* __restrict__ pts, const float* __restrict__ ray_d, int V,
                                                          int depth, float* __restrict__ rgb_out, float* __restrict__ ws,
                                                          int64_t row_floats, int64_t smp_floats, float* __restrict__ alpha_out,
                                                          GntDrop dr, const unsigned* __restrict__ seed_dev, int64_t ray0) {
    HIP_DYNAMIC_SHARED(float, lds)
    if (TRAIN && seed_dev) dr.seed = *seed_dev;         // (a captured step reads its seed from a device word: a replay must not freeze it)
    constexpr int S = 32 * NW;
    float* tab = lds;
    float* Kl = lds + GM_TAB_FLOATS;
    float* Vl = Kl + GmLds<NW>::K_FLOATS;
    float* red = Vl + GmLds<NW>::VT_FLOATS;
    const int64_t ray = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    float* Qp = Kl + wave * 2 * GM_PARK_FLOATS;       // view phase: this wave's q | v + pos
    float* VPp = Qp + GM_PARK_FLOATS;
    GmCtx c;
    c.S = S; c.V = V; c.s = wave * 32 + m; c.h = h;
    c.smp_lane = (unsigned)(4 * h * S + c.s);
    c.row_lane = (unsigned)(4 * h * S + c.s);
    c.view_floats = row_floats * S;
    const size_t per_ray = (size_t)S * V * row_floats + (size_t)S * smp_floats;
    c.ws_row = ws + ray * per_ray;
    c.ws_smp = c.ws_row + (size_t)S * V * row_floats;
    const float* rgb_feat = rgb_feat_all + ray * S * V * 35;
    const float* ray_diff = ray_diff_all + ray * S * V * 4;
    const float* mask = mask_all + ray * S * V;
    constexpr bool sv = SAVE;
    const unsigned long long gray = (unsigned long long)(ray0 + ray);          // the ray's index in the call's tensors (Dropout indices)
    const unsigned long long smp_idx = gray * S + (unsigned)c.s;

    // ---- stem: X_v = W2 relu(W1 rgb_feat_v + b1) + b2 (kept in the workspace), q = max over the views (first maximum wins)
    V64 cur, amax;
    cur.t[0] = cur.t[1] = amax.t[0] = amax.t[1] = gm_zero();
    for (int v = 0; v < V; ++v) {
        const float* rf = rgb_feat + ((size_t)c.s * V + v) * 35;
        g16 in0, in1;                     // k order 2 s + h: 18 steps, feature 35 is padding
#pragma unroll
        for (int st = 0; st < 16; ++st) in0[st] = rf[2 * st + h];
        in1 = gm_zero();
        in1[0] = rf[32 + h];
        in1[1] = h == 0 ? rf[34] : 0.f;
        V64 r1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            g16 acc = gm_tile(wb + MS_B1 + t * 32, h);
            acc = gm_gemm<16>(wb + (MS_L1 + t * 18) * 64, lane, in0, acc);
            acc = gm_gemm<2>(wb + (MS_L1 + t * 18 + 16) * 64, lane, in1, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
            r1.t[t] = acc;
        }
        if (sv) {       // relu' of the stem's hidden layer: the two tiles' sign bits in one word
            const unsigned bits = __float_as_uint(gm_sign_word(r1.t[0])) | (__float_as_uint(gm_sign_word(r1.t[1])) << 16);
            GM_SAVE(gm_row_at(c, RW_R1, v) + c.row_lane, __uint_as_float(bits));
        }
        V64 x = gm_lin64(wb + MS_L2 * 64, wb + MS_B2, lane, h, r1);
        gm_store_row(c, RW_X, v, x);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (v == 0 || x.t[t][r] > cur.t[t][r]) {
                    cur.t[t][r] = x.t[t][r];
                    amax.t[t][r] = (float)v;
                }
    }
    if (sv) gm_save_smp(c, SW_AMAX, amax);
    // ---- positional encodings (sample position | unit view direction), 126 features at SW_PE in memory order: lane (m, h)
    //      writes the features 2 st + h
    {
        const float* p3 = pts + (ray * S + c.s) * 3;
        const float* d3 = ray_d + ray * 3;
        const float dn = sqrtf(d3[0] * d3[0] + d3[1] * d3[1] + d3[2] * d3[2]);
        for (int st = 0; st < 63; ++st) {
            const int f = 2 * st + h, part = f >= 63 ? 1 : 0, j = f - 63 * part;
            float val;
            if (j < 3) {
                val = part == 0 ? p3[j] : d3[j] / dn;
            } else {
                const int k = (j - 3) / 6, a = (j - 3) - 6 * k, ax = a >= 3 ? a - 3 : a;
                const float xa = part == 0 ? p3[ax] : d3[ax] / dn;
                const float arg = xa * (float)(1 << k);
                val = a >= 3 ? cosf(arg) : sinf(arg);
            }
            gm_smp_at(c, SW_PE + 2 * st)[h * S + c.s] = val;
        }
    }
    GmW w;
    gm_w_start(w, wb, depth, wb + gm_layer_base(0) + MG_STREAM * 64, lane);
    for (int i = 0; i < depth; ++i) {
        const float* Lbase = wb + gm_layer_base(i);
        const float* Lst = Lbase + MG_STREAM * 64;                                        // this layer's stream
        const float* next_stream = i + 1 < depth ? wb + gm_layer_base(i + 1) + MG_STREAM * 64 : nullptr;
        const int ls = SW_BASE + (sv ? i : 0) * SW_LAYER;
        const int lr = RW_BASE + (sv ? i : 0) * RW_LAYER;
        // ---- the layer's small records and tables -> LDS (the previous layer's are dead behind the first barrier)
        __syncthreads();
        {
            const float4* src_rec = reinterpret_cast<const float4*>(Lbase);
            const float4* src_tab = reinterpret_cast<const float4*>(Lbase + (size_t)MG_LAYER_RECORDS * 64);
            float4* dst = reinterpret_cast<float4*>(tab);
            for (int k = threadIdx.x; k < GM_TAB_REC / 4; k += 64 * NW) dst[k] = src_rec[k];
            for (int k = threadIdx.x; k < MB_FLOATS / 4; k += 64 * NW) dst[GM_TAB_REC / 4 + k] = src_tab[k];
        }
        __syncthreads();
        const float* tb = tab + GM_TAB_REC;
        // ================= view transformer =================
        {
            V64 y = gm_layernorm(c, cur, tb + MB_LN, tb + MB_LN + 64, 1e-6f, sv, ls + SL_XH1, ls + SL_RSTD1);
            // the running state of the online softmax needs 96 registers: the residual stream waits in the (L2-resident)
            // workspace and Q in this wave's LDS slab while the views are walked
            gm_store_smp(c, SW_CUR, cur);
            gm_park(Qp, lane, gm_lin64_s(w, nullptr, lane, h, y));
            V64 mx, sum, acc;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    mx.t[t][r] = -3.0e38f;
                    sum.t[t][r] = 0.f;
                    acc.t[t][r] = 0.f;
                }
            // X_v, the view's ray difference and mask are fetched one view ahead (issued in front of the softmax arithmetic of
            // the previous view, which covers their HBM latency)
            V64 X = gm_load_row(c, RW_X, 0);
            float rd0 = ray_diff[(size_t)c.s * V * 4 + h], rd1 = ray_diff[(size_t)c.s * V * 4 + 2 + h];
            float mk = mask[(size_t)c.s * V];
            for (int v = 0; v < V; ++v) {
                const int oz = gm_opaque_zero();
                const float* L = tab + oz;
                const float* tb = tab + GM_TAB_REC + oz;
                // register budget: the running softmax state (96) stays live across the views, so V + pos is parked in LDS
                // (and stored for the backward) and Q is read back per four channels instead of held
                V64 T2;
                {
                    const V64 K = gm_lin64_s(w, nullptr, lane, h, X);
                    g16 rin = gm_zero();
                    rin[0] = rd0;
                    rin[1] = rd1;
                    g16 h0 = gm_gemm<2>(L + MG_POS0 * 64, lane, rin, gm_tile(tb + MB_POS0, h));
#pragma unroll
                    for (int r = 0; r < 4; ++r) h0[r] = fmaxf(h0[r], 0.f);
                    const V64 Vv = gm_lin64_s(w, nullptr, lane, h, K);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const g16 pos = gm_gemm<4>(L + (MG_POS2 + 4 * t) * 64, lane, h0, gm_tile(tb + MB_POS2 + t * 32, h));
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 qv = gm_parked4(Qp, lane, t, q);
                            const float qq[4] = {qv.x, qv.y, qv.z, qv.w};
                            float vp[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int r = 4 * q + e;
                                vp[e] = Vv.t[t][r] + pos[r];
                                if (sv) GM_SAVE(gm_row_at(c, lr + RWL_VP + 32 * t + gm_nidx(r, 0), v) + c.row_lane, vp[e]);
                                T2.t[t][r] = K.t[t][r] - qq[e] + pos[r];      // k - q + pos
                            }
                            gm_park4(VPp, lane, t, q, vp[0], vp[1], vp[2], vp[3]);
                        }
                    }
                }
                if (v + 1 < V) w.p = Lst + ST_VK * 64;      // the last two chunks of the body prefetch the next view's first two
                g16 hid = gm_tile_s(w, lane, T2, gm_tile(tb + MB_ATT0, h));
#pragma unroll
                for (int r = 0; r < 4; ++r) hid[r] = fmaxf(hid[r], 0.f);
                if (sv) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) GM_SAVE(gm_row_at(c, lr + RWL_H + r, v) + c.row_lane, hid[r]);
                }
                const float mk_v = mk;
                {   // the next view's inputs (the last view re-reads itself)
                    const int vn = v + 1 < V ? v + 1 : v;
                    X = gm_load_row(c, RW_X, vn);
                    const float* rd = ray_diff + ((size_t)c.s * V + vn) * 4;
                    rd0 = rd[h];
                    rd1 = rd[2 + h];
                    mk = mask[(size_t)c.s * V + vn];
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    g16 a = gm_gemm<4>(L + (MG_ATT2 + 4 * t) * 64, lane, hid, gm_tile(tb + MB_ATT2 + t * 32, h));
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 pv = gm_parked4(VPp, lane, t, q);
                        const float vpq[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 4 * q + e;
                            const float lg = mk_v == 0.f ? -1e9f : a[r];
                            if (sv) GM_SAVE(gm_row_at(c, lr + RWL_PROB + 32 * t + gm_nidx(r, 0), v) + c.row_lane, lg);
                            const float mn = fmaxf(mx.t[t][r], lg);
                            const float sc = __expf(mx.t[t][r] - mn), p = __expf(lg - mn);
                            sum.t[t][r] = sum.t[t][r] * sc + p;         // (the softmax normalises BEFORE Dropout site 0, :85)
                            const float pk = p * gm_keep<TRAIN>(dr, 8 * i + 0, (smp_idx * V + v) * 64 + (unsigned)(32 * t + gm_nidx(r, h)));
                            acc.t[t][r] = acc.t[t][r] * sc + pk * vpq[e];
                            mx.t[t][r] = mn;
                        }
                    }
                }
            }
            V64 u;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // one reciprocal per channel (v_rcp_f32, 1 ulp), re-used by the V normalised weights below: an IEEE division is a
                    // ten-instruction vector sequence, and this block divides 32 (V + 1) times per sample and layer
                    sum.t[t][r] = nf_rcp(sum.t[t][r]);
                    u.t[t][r] = acc.t[t][r] * sum.t[t][r];
                }
            if (sv) {       // the logits stay where the loop parked them: the backward re-forms p_v = exp(logit_v - max) / sum
                gm_save_smp(c, ls + SL_MX, mx);
                gm_save_smp(c, ls + SL_RS, sum);
            }
            if (sv) gm_save_smp(c, ls + SL_U, u);
            const V64 o = gm_drop64<TRAIN>(dr, 8 * i + 1, smp_idx * 64, h, gm_lin64_s(w, tb + MB_VOUT, lane, h, u));      // :88
            cur = gm_load_smp(c, SW_CUR);
#pragma unroll
            for (int t = 0; t < 2; ++t) cur.t[t] += o.t[t];
            y = gm_layernorm(c, cur, tb + MB_LN + 128, tb + MB_LN + 192, 1e-6f, sv, ls + SL_XH2, ls + SL_RSTD2);
            const V64 f = gm_drop64<TRAIN>(dr, 8 * i + 3, smp_idx * 64, h,
                                           gm_ff<TRAIN>(c, w, tb + MB_VFF1, tb + MB_VFF2, lane, y, sv, ls + SL_F,
                                                        (i & 1) ? Lst + ST_RQ * 64 : nullptr, dr, 8 * i + 2, smp_idx));
#pragma unroll
            for (int t = 0; t < 2; ++t) cur.t[t] += f.t[t];
        }
        // ================= positional MLP on even layers =================
        if ((i & 1) == 0) {
            V64 g;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                g16 a = gm_tile_s(w, lane, cur, gm_tile(tb + MB_Q0 + t * 32, h));
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {        // 126 positional features in memory order 2 st + h (+ one zero step)
                    g16 pe;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int st = 16 * ch + r;
                        pe[r] = st < 63 ? gm_smp_at(c, SW_PE + 2 * st)[h * S + c.s] : 0.f;
                    }
                    a = (ch & 1) ? gm_take_b(w, lane, pe, a) : gm_take_a(w, lane, pe, a);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f);
                g.t[t] = a;
            }
            if (sv) gm_save_smp(c, ls + SL_G, g);
            cur = gm_lin64_s(w, tb + MB_Q2, lane, h, g);
        }
        // ================= ray transformer =================
        {
            V64 y = gm_layernorm(c, cur, tb + MB_LN + 256, tb + MB_LN + 320, 1e-6f, sv, ls + SL_RXH1, ls + SL_RRSTD1);
            V64 q = gm_lin64_s(w, nullptr, lane, h, y);
            {
                const V64 k = gm_lin64_s(w, nullptr, lane, h, y);
                const V64 vv = gm_lin64_s(w, nullptr, lane, h, y);
                if (sv) {
                    gm_save_smp(c, ls + SL_QH, q);
                    gm_save_smp(c, ls + SL_KH, k);
                    gm_save_smp(c, ls + SL_VH, vv);
                }
                // head hd = features 16 hd .. 16 hd + 15 = tile hd / 2, registers 8 (hd & 1) .. + 7, dims n(j, h)
                // K record (hd, key tile = wave, step j): lane (key m, h) -> K[key][dim n(j, h)]  == own register
                // V^T record (hd, key tile, step r): lane (dim i < 16, hh) -> V[key n(r, hh)][dim i]: this key is m = n(r, hh)
                const int rk = ((m & 3) | ((m >> 3) << 2)), hk = (m >> 2) & 1;      // inverse of n(r, hh) for key m
                __syncthreads();        // every wave is through with its parked q / v + pos, which the image overlays
#pragma unroll
                for (int hd = 0; hd < 4; ++hd) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int r = 8 * (hd & 1) + j;
                        Kl[((hd * NW + wave) * 8 + j) * 64 + lane] = k.t[hd >> 1][r];
                        Vl[((hd * NW + wave) * 16 + rk) * GM_VT_STRIDE + gm_nidx(j, h) + 32 * hk] = vv.t[hd >> 1][r];
                    }
                }
            }
            __syncthreads();
            V64 att;
#pragma unroll
            for (int hd = 0; hd < 4; ++hd) {
                g16 sc[NW];
                float mxh = -3.0e38f;
#pragma unroll
                for (int kt = 0; kt < NW; ++kt) {
                    g16 a = gm_zero();
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        a = GM_MFMA(Kl[((hd * NW + kt) * 8 + j) * 64 + lane], q.t[hd >> 1][8 * (hd & 1) + j] * 0.25f, a);
#pragma unroll
                    for (int r = 0; r < 16; ++r) mxh = fmaxf(mxh, a[r]);
                    sc[kt] = a;
                }
                mxh = gm_half_max(mxh);
                float l = 0.f;
#pragma unroll
                for (int kt = 0; kt < NW; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        sc[kt][r] = __expf(sc[kt][r] - mxh);
                        l += sc[kt][r];
                    }
                l = gm_half_sum(l);
                if (TRAIN) {        // Dropout site 4 on the normalised probabilities [R, 4, S, S] (:162): query = own sample, key = row
                    const unsigned long long arow = ((gray * 4 + (unsigned)hd) * S + (unsigned)c.s) * S;
#pragma unroll
                    for (int kt = 0; kt < NW; ++kt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sc[kt][r] *= gnt_keep(dr, 8 * i + 4, arow + (unsigned)(32 * kt + gm_nidx(r, h)));
                }
                // ret_alpha (the DROPPED attention in training mode, :162-169): attention row of sample 0 in the LAST ray transformer, averaged over the heads; the lanes of
                // query 0 (wave 0, m == 0) hold it -- keys 32 kt + n(r, h) -- and accumulate straight into the output
                if (alpha_out && i == depth - 1 && wave == 0 && m == 0) {
#pragma unroll
                    for (int kt = 0; kt < NW; ++kt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float* ap = alpha_out + ray * S + 32 * kt + gm_nidx(r, h);
                            *ap = (hd == 0 ? 0.f : *ap) + sc[kt][r] / l * 0.25f;
                        }
                }
                g16 o = gm_zero();
#pragma unroll
                for (int kt = 0; kt < NW; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float av = (lane & 31) < 16 ? Vl[((hd * NW + kt) * 16 + r) * GM_VT_STRIDE + (lane & 31) + 32 * h] : 0.f;
                        o = GM_MFMA(av, sc[kt][r], o);
                    }
                if (sv && h == 0) {
                    GM_SAVE(gm_smp_at(c, ls + SL_ML + hd) + c.s, mxh);
                    GM_SAVE(gm_smp_at(c, ls + SL_ML + 4 + hd) + c.s, l);
                }
                const float rl = nf_rcp(l);
#pragma unroll
                for (int j = 0; j < 8; ++j) att.t[hd >> 1][8 * (hd & 1) + j] = o[j] * rl;
            }
            __syncthreads();        // K / V of this layer are dead: the next layer may overwrite the LDS image
            if (sv) gm_save_smp(c, ls + SL_OUTA, att);
            const V64 o = gm_drop64<TRAIN>(dr, 8 * i + 5, smp_idx * 64, h, gm_lin64_s(w, tb + MB_ROUT, lane, h, att));     // :166
#pragma unroll
            for (int t = 0; t < 2; ++t) cur.t[t] += o.t[t];
            y = gm_layernorm(c, cur, tb + MB_LN + 384, tb + MB_LN + 448, 1e-6f, sv, ls + SL_RXH2, ls + SL_RRSTD2);
            const V64 f = gm_drop64<TRAIN>(dr, 8 * i + 7, smp_idx * 64, h,
                                           gm_ff<TRAIN>(c, w, tb + MB_RFF1, tb + MB_RFF2, lane, y, sv, ls + SL_F2, next_stream, dr,
                                                        8 * i + 6, smp_idx));
#pragma unroll
            for (int t = 0; t < 2; ++t) cur.t[t] += f.t[t];
        }
    }
    // ---- final LayerNorm (eps 1e-5), mean over the samples, rgb_fc
    {
        const float* F = wb + gm_layer_base(depth);
        const V64 hf = gm_layernorm(c, cur, F + MF_LNW, F + MF_LNB, 1e-5f, true, SW_XHF, SW_RSTDF);
        float part[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float t = 0.f;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) t = fmaf(F[MF_W + ch * 64 + tt * 32 + h * 16 + r], hf.t[tt][r], t);
            part[ch] = nf_wave_sum(t);
        }
        if (lane == 0) {
            red[wave * 4 + 0] = part[0];
            red[wave * 4 + 1] = part[1];
            red[wave * 4 + 2] = part[2];
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            float t = 0.f;
            for (int w = 0; w < NW; ++w) t += red[w * 4 + threadIdx.x];
            rgb_out[ray * 3 + threadIdx.x] = t / (float)S + F[MF_B + threadIdx.x];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward on the matrix cores: d rgb [R,3] -> d rgb_feat [R,S,V,35], from the activations k_gnt_fwd_mfma with save != 0 left in
// the workspace.  Same decomposition as the forward (one workgroup per ray, one wave per 32 samples, register-chained
// transposed GEMMs dX^T = W^T dY^T, weight stream).  The ray attention is differentiated head by head with seven record
// sets in LDS: scores are formed in BOTH orientations -- keys on rows (lane = query) for dQ, queries on rows (lane = key)
// for dK / dV -- so that every contraction runs over MFMA rows and no cross-lane reduction is needed.
// Follows oracle/gnt_manual_bwd.py and k_gnt_bwd (nf_gnt.hip), which stays the reference implementation for other shapes.
// ---------------------------------------------------------------------------------------------------------------
// 64 -> 64 transposed linear through the stream; jump (nullable) redirects the stream behind this GEMM
__device__ __forceinline__ V64 gm_lin64_sj(GmW& w, int lane, const V64& x, const V64* init, const float* jump) {
    V64 y;
    y.t[0] = gm_tile_s(w, lane, x, init ? init->t[0] : gm_zero());
    if (jump) w.p = jump;
    y.t[1] = gm_tile_s(w, lane, x, init ? init->t[1] : gm_zero());
    return y;
}

// LayerNorm backward: returns rstd (dxh - mean(dxh) - xh mean(dxh xh)), dxh = dy w   (w in fragment order)
__device__ __forceinline__ V64 gm_ln_bwd(const GmCtx& c, const V64& dy, const float* __restrict__ w, int xh_slot, int rstd_slot) {
    const V64 xh = gm_load_smp(c, xh_slot);
    const float rstd = gm_smp_at(c, rstd_slot)[c.s];
    V64 dxh;
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dxh.t[t][r] = dy.t[t][r] * w[t * 32 + c.h * 16 + r];
            m1 += dxh.t[t][r];
            m2 += dxh.t[t][r] * xh.t[t][r];
        }
    m1 = gm_half_sum(m1) / 64.f;
    m2 = gm_half_sum(m2) / 64.f;
    V64 o;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o.t[t][r] = rstd * (dxh.t[t][r] - m1 - xh.t[t][r] * m2);
    return o;
}

// feed-forward backward: d(FF input y) = W1^T (relu'(F) . W2^T d out); F read from f_slot; stream: 8 x [fc2^T tile j | fc1^T k-tile j]
// scale: 1 (eval) or 1 / (1 - p) (training mode: the sign words mark the units that survived ReLU AND Dropout)
__device__ __forceinline__ V64 gm_ff_bwd(const GmCtx& c, GmW& w, int lane, const V64& dout, int f_slot, float scale) {
    V64 dy;
    dy.t[0] = dy.t[1] = gm_zero();
    float sg[8];            // relu' of the 8 hidden tiles: one sign word each (gm_sign_word)
#pragma unroll
    for (int j = 0; j < 8; ++j) sg[j] = gm_smp_at(c, f_slot + gm_sign_slot(j))[c.smp_lane];
    GM_PIN();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        g16 df = gm_tile_s(w, lane, dout, gm_zero());
        const unsigned bits = __float_as_uint(sg[j]);
#pragma unroll
        for (int r = 0; r < 16; ++r) df[r] = (bits >> r) & 1u ? df[r] * scale : 0.f;
        dy.t[0] = gm_take_a(w, lane, df, dy.t[0]);
        dy.t[1] = gm_take_b(w, lane, df, dy.t[1]);
    }
    return dy;
}

template <int NW> struct GmBwdLds {
    static constexpr int REC = NW * 8 * 64;                    // rows x dims record set (K, V, Q, dO)
    static constexpr int TREC = NW * 16 * GM_VT_STRIDE;        // dims x rows record set (K^T, Q^T, dO^T)
    static constexpr int SCAL = 32 * NW;                       // per-query scalars (max, 1 / sum, D)
    static constexpr int TAB = 8 * 64 + 512;                   // the layer's attn_fc.0^T records | its four LayerNorm (weight, bias) pairs
    static constexpr int FLOATS = TAB + 4 * REC + 3 * TREC + 3 * SCAL;
};

template <int NW, bool TRAIN>
__global__ void __launch_bounds__(64 * NW) k_gnt_bwd_mfma(const float* __restrict__ wb, const float* __restrict__ mask_all,
                                                          const float* __restrict__ d_rgb, int V, int depth,
                                                          float* __restrict__ d_rgb_feat, float* __restrict__ ws,
                                                          int64_t row_floats, int64_t smp_floats, GntDrop dr,
                                                          const unsigned* __restrict__ seed_dev) {
    HIP_DYNAMIC_SHARED(float, lds)
    if (TRAIN && seed_dev) dr.seed = *seed_dev;         // the seed its forward read (PGDAttack stages one seed per captured forward)
    const float ff_scale = TRAIN ? dr.scale : 1.f;
    constexpr int S = 32 * NW;
    float* tab = lds;
    float* Kr = lds + GmBwdLds<NW>::TAB;
    float* Vr = Kr + GmBwdLds<NW>::REC;
    float* Qr = Vr + GmBwdLds<NW>::REC;
    float* Gr = Qr + GmBwdLds<NW>::REC;
    float* Kt = Gr + GmBwdLds<NW>::REC;
    float* Qt = Kt + GmBwdLds<NW>::TREC;
    float* Gt = Qt + GmBwdLds<NW>::TREC;
    float* Ms = Gt + GmBwdLds<NW>::TREC;
    float* Ls = Ms + GmBwdLds<NW>::SCAL;
    float* Ds = Ls + GmBwdLds<NW>::SCAL;
    const int64_t ray = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    GmCtx c;
    c.S = S; c.V = V; c.s = wave * 32 + m; c.h = h;
    c.smp_lane = (unsigned)(4 * h * S + c.s);
    c.row_lane = (unsigned)(4 * h * S + c.s);
    c.view_floats = row_floats * S;
    const size_t per_ray = (size_t)S * V * row_floats + (size_t)S * smp_floats;
    c.ws_row = ws + ray * per_ray;
    c.ws_smp = c.ws_row + (size_t)S * V * row_floats;
    const float* mask = mask_all + ray * S * V;
    const unsigned long long gray = (unsigned long long)ray, smp_idx = gray * S + (unsigned)c.s;

    // ---- rgb_fc, mean over the samples, final LayerNorm
    V64 dcur;
    {
        const float* F = wb + gm_layer_base(depth);
        const float* g = d_rgb + ray * 3;
        V64 dm;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int at = t * 32 + h * 16 + r;
                dm.t[t][r] = (F[MF_W + at] * g[0] + F[MF_W + 64 + at] * g[1] + F[MF_W + 128 + at] * g[2]) / (float)S;
            }
        dcur = gm_ln_bwd(c, dm, F + MF_LNW, SW_XHF, SW_RSTDF);
    }
    GmW w;
    gm_w_start(w, wb, depth, wb + gm_bwd_layer_base(depth, depth - 1) + BG_STREAM * 64, lane);
    for (int i = depth - 1; i >= 0; --i) {
        const float* Bl = wb + gm_bwd_layer_base(depth, i);
        const float* Bst = Bl + BG_STREAM * 64;
        // ---- the layer's attn_fc.0^T records and LayerNorm weights -> LDS (read inside the view loop / right behind a GEMM, where
        //      a load from L2 sits fully exposed with one wave per SIMD)
        __syncthreads();
        {
            const float4* src_rec = reinterpret_cast<const float4*>(Bl + BG_ATT0T * 64);
            const float4* src_ln = reinterpret_cast<const float4*>(wb + gm_layer_base(i) + (size_t)MG_LAYER_RECORDS * 64 + MB_LN);
            float4* dst = reinterpret_cast<float4*>(tab);
            for (int k = threadIdx.x; k < 128; k += 64 * NW) {
                dst[k] = src_rec[k];
                dst[128 + k] = src_ln[k];
            }
        }
        __syncthreads();
        const float* ln = tab + 512;                                                // LayerNorm tables as at MB_LN of the forward section
        const float* after = i > 0 ? wb + gm_bwd_layer_base(depth, i - 1) + BG_STREAM * 64 : wb + gm_bwd_stem_base(depth);
        const int ls = SW_BASE + i * SW_LAYER;
        const int lr = RW_BASE + i * RW_LAYER;
        // ================= ray transformer backward =================
        {
            const V64 dy = gm_ff_bwd(c, w, lane, gm_drop64<TRAIN>(dr, 8 * i + 7, smp_idx * 64, h, dcur), ls + SL_F2, ff_scale);
            const V64 dl = gm_ln_bwd(c, dy, ln + 384, ls + SL_RXH2, ls + SL_RRSTD2);
#pragma unroll
            for (int t = 0; t < 2; ++t) dcur.t[t] += dl.t[t];
        }
        V64 dq, dk, dv;
        {
            const V64 go = gm_lin64_sj(w, lane, gm_drop64<TRAIN>(dr, 8 * i + 5, smp_idx * 64, h, dcur), nullptr, nullptr);            // d (attention output)
            const int rk = ((m & 3) | ((m >> 3) << 2)), hk = (m >> 2) & 1;          // inverse of n(r, hh) for sample m
#pragma unroll
            for (int hd = 0; hd < 4; ++hd) {
                const int tt = hd >> 1, r0 = 8 * (hd & 1);
                // ---- this wave's 32 samples -> the seven record sets of head hd, and the per-query scalars; the head's 16
                //      dims of q, k, v are re-read from the workspace (features 16 hd + n(j, h)) instead of being held
                float qh[8], kh[8], vh[8];
                float D = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f0 = 16 * hd + gm_nidx(j, 0);
                    qh[j] = gm_smp_at(c, ls + SL_QH + f0)[c.smp_lane] * 0.25f;
                    kh[j] = gm_smp_at(c, ls + SL_KH + f0)[c.smp_lane];
                    vh[j] = gm_smp_at(c, ls + SL_VH + f0)[c.smp_lane];
                    const float gv = go.t[tt][r0 + j];
                    const int ar = (wave * 8 + j) * 64 + lane;
                    const int at = (wave * 16 + rk) * GM_VT_STRIDE + gm_nidx(j, h) + 32 * hk;
                    Kr[ar] = kh[j];
                    Vr[ar] = vh[j];
                    Qr[ar] = qh[j];
                    Gr[ar] = gv;
                    Kt[at] = kh[j];
                    Qt[at] = qh[j];
                    Gt[at] = gv;
                    D = fmaf(gv, gm_smp_at(c, ls + SL_OUTA + f0)[c.smp_lane], D);
                }
                D = gm_half_sum(D);
                const float mxq = gm_smp_at(c, ls + SL_ML + hd)[c.s], rlq = 1.f / gm_smp_at(c, ls + SL_ML + 4 + hd)[c.s];
                if (h == 0) {
                    Ms[c.s] = mxq;
                    Ls[c.s] = rlq;
                    Ds[c.s] = D;
                }
                __syncthreads();
                // ---- orientation A: keys on rows, lane = query (own sample) -> dQ
                g16 dqa = gm_zero();
#pragma unroll 1
                for (int kt = 0; kt < NW; ++kt) {
                    g16 sT = gm_zero(), dPT = gm_zero();
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        sT = GM_MFMA(Kr[(kt * 8 + j) * 64 + lane], qh[j], sT);
                        dPT = GM_MFMA(Vr[(kt * 8 + j) * 64 + lane], go.t[tt][r0 + j], dPT);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pT = __expf(sT[r] - mxq) * rlq;
                        // (training mode: O = sum_k keep_qk P_qk v_k, so d P_qk = keep_qk (d O_q . v_k); D = d O_q . O_q as before)
                        const float kp = gm_keep<TRAIN>(dr, 8 * i + 4, ((gray * 4 + (unsigned)hd) * S + (unsigned)c.s) * S + (unsigned)(32 * kt + gm_nidx(r, h)));
                        sT[r] = pT * (dPT[r] * kp - D);                         // d S^T
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float av = (lane & 31) < 16 ? Kt[(kt * 16 + r) * GM_VT_STRIDE + (lane & 31) + 32 * h] : 0.f;
                        dqa = GM_MFMA(av, sT[r], dqa);
                    }
                }
                // ---- orientation B: queries on rows, lane = key (own sample) -> dK, dV
                g16 dka = gm_zero(), dva = gm_zero();
#pragma unroll 1
                for (int qt = 0; qt < NW; ++qt) {
                    g16 sc = gm_zero(), dP = gm_zero();
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        sc = GM_MFMA(Qr[(qt * 8 + j) * 64 + lane], kh[j], sc);
                        dP = GM_MFMA(Gr[(qt * 8 + j) * 64 + lane], vh[j], dP);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int qi = qt * 32 + gm_nidx(r, h);
                        const float pr = __expf(sc[r] - Ms[qi]) * Ls[qi];
                        const float kp = gm_keep<TRAIN>(dr, 8 * i + 4, ((gray * 4 + (unsigned)hd) * S + (unsigned)qi) * S + (unsigned)c.s);
                        dP[r] = pr * (dP[r] * kp - Ds[qi]);                     // d S
                        sc[r] = pr * kp;                                        // what multiplied v_k in the forward
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool on = (lane & 31) < 16;
                        const int at = (qt * 16 + r) * GM_VT_STRIDE + (lane & 31) + 32 * h;
                        dka = GM_MFMA(on ? Qt[at] : 0.f, dP[r], dka);
                        dva = GM_MFMA(on ? Gt[at] : 0.f, sc[r], dva);
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    dq.t[tt][r0 + j] = dqa[j] * 0.25f;
                    dk.t[tt][r0 + j] = dka[j];
                    dv.t[tt][r0 + j] = dva[j];
                }
                __syncthreads();
            }
        }
        {
            V64 dx = gm_lin64_sj(w, lane, dq, nullptr, nullptr);
            dx = gm_lin64_sj(w, lane, dk, &dx, nullptr);
            dx = gm_lin64_sj(w, lane, dv, &dx, (i & 1) ? Bst + BS_VFF * 64 : nullptr);      // odd layers: no positional MLP records
            const V64 dl = gm_ln_bwd(c, dx, ln + 256, ls + SL_RXH1, ls + SL_RRSTD1);
#pragma unroll
            for (int t = 0; t < 2; ++t) dcur.t[t] += dl.t[t];
        }
        // ================= positional MLP backward (even layers): q = W2 relu(W0 [q | pe] + b0) + b2, no residual =================
        if ((i & 1) == 0) {
            V64 dg = gm_lin64_sj(w, lane, dcur, nullptr, nullptr);
            const V64 gsv = gm_load_smp(c, ls + SL_G);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dg.t[t][r] = gsv.t[t][r] > 0.f ? dg.t[t][r] : 0.f;
            dcur = gm_lin64_sj(w, lane, dg, nullptr, nullptr);
        }
        // ================= view transformer backward =================
        {
            const V64 dy = gm_ff_bwd(c, w, lane, gm_drop64<TRAIN>(dr, 8 * i + 3, smp_idx * 64, h, dcur), ls + SL_F, ff_scale);
            const V64 dl = gm_ln_bwd(c, dy, ln + 128, ls + SL_XH2, ls + SL_RSTD2);
#pragma unroll
            for (int t = 0; t < 2; ++t) dcur.t[t] += dl.t[t];
        }
        {
            const V64 du = gm_lin64_sj(w, lane, gm_drop64<TRAIN>(dr, 8 * i + 1, smp_idx * 64, h, dcur), nullptr, nullptr);
            // the view softmax (training mode: u = sum_v keep_v p_v (v + pos)_v -- the saved u is that sum, and keep_v multiplies (v + pos)_v below): p_v = exp(logit_v - mx) rs with the forward's running maximum and reciprocal sum; rs is folded
            // into du (dur = rs du), and sum_v p_v (v + pos)_v du = u du with the forward's u = sum_v p_v (v + pos)_v
            V64 dur, sp, dqs;
            // the view's saved activations are fetched one view ahead: issued in front of the LAST GEMM of the previous view (a
            // 64-MFMA window, and the point of the body with the fewest live registers).  View 0's are requested HERE, in front of
            // the statistics loads below: every path into the loop then has the 32 d X_v atomics / the statistics loads behind
            // them in the memory counter, and the wait at the loop top is vmcnt(32), not vmcnt(0) -- it does not sit out the atomics
            V64 pr = gm_load_row(c, lr + RWL_PROB, 0), vp = gm_load_row(c, lr + RWL_VP, 0);
            float mk = mask[(size_t)c.s * V];
            GM_PIN();
            const V64 mx = gm_load_smp(c, ls + SL_MX);
            {
                const V64 u = gm_load_smp(c, ls + SL_U), rs = gm_load_smp(c, ls + SL_RS);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        dur.t[t][r] = rs.t[t][r] * du.t[t][r];
                        sp.t[t][r] = u.t[t][r] * dur.t[t][r];
                    }
            }
            dqs.t[0] = dqs.t[1] = gm_zero();
            for (int v = 0; v < V; ++v) {
                const int oz = gm_opaque_zero();
                const float* Bs = Bst + oz;
                const float* Bsmall = tab + oz;
                V64 dlg, dvv;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float e = __expf(pr.t[t][r] - mx.t[t][r]);          // (pr holds the logits)
                        const float kp = gm_keep<TRAIN>(dr, 8 * i + 0, (smp_idx * V + v) * 64 + (unsigned)(32 * t + gm_nidx(r, h)));
                        dlg.t[t][r] = mk == 0.f ? 0.f : e * (kp * vp.t[t][r] * dur.t[t][r] - sp.t[t][r]);
                        dvv.t[t][r] = e * kp * dur.t[t][r];
                    }
                float hsv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) hsv[r] = gm_row_at(c, lr + RWL_H + r, v)[c.row_lane];
                GM_PIN();
                g16 dh = gm_tile_s(w, lane, dlg, gm_zero());                       // attn_fc.2^T: 8 <- 64
#pragma unroll
                for (int r = 0; r < 4; ++r) dh[r] = hsv[r] > 0.f ? dh[r] : 0.f;
                V64 da;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    da.t[t] = gm_gemm<4>(Bsmall + 4 * t * 64, lane, dh, gm_zero());                   // attn_fc.0^T: 64 <- 8
                    dqs.t[t] += da.t[t];
                }
                const V64 dK = gm_lin64_sj(w, lane, dvv, &da, nullptr);            // d K = d a + Wv^T d Vv
                {
                    const int vn = v + 1 < V ? v + 1 : v;                          // (the last view re-reads itself)
                    pr = gm_load_row(c, lr + RWL_PROB, vn);
                    vp = gm_load_row(c, lr + RWL_VP, vn);
                    mk = mask[(size_t)c.s * V + vn];
                    GM_PIN();
                }
                const V64 dxv = gm_lin64_sj(w, lane, dK, nullptr, v + 1 < V ? Bs + BS_VIEW * 64 : nullptr);      // Wk^T d K
                // d X_v accumulates over the layers in the workspace: a plain store in the layer the backward starts with, then
                // device-scope float atomics -- nothing to wait for, no accumulator to fetch (one writer per address: the order of
                // the additions, and so the result, is fixed)
                if (i == depth - 1) {
                    gm_store_row(c, RW_DX, v, dxv);
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) atomicAdd(gm_row_at(c, RW_DX + 32 * t + gm_nidx(r, 0), v) + c.row_lane, dxv.t[t][r]);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dqs.t[t][r] = -dqs.t[t][r];          // d Q = - sum_v d a_v
            const V64 dx = gm_lin64_sj(w, lane, dqs, nullptr, after);
            const V64 dl = gm_ln_bwd(c, dx, ln, ls + SL_XH1, ls + SL_RSTD1);
#pragma unroll
            for (int t = 0; t < 2; ++t) dcur.t[t] += dl.t[t];
        }
    }
    // ---- q0 = max over the views routes its gradient to the arg-max view of every channel; then the stem
    {
        const V64 am = gm_load_smp(c, SW_AMAX);
        const float* St = wb + gm_bwd_stem_base(depth);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // d X_v was accumulated with L2 atomics: the loads below must not be served by the vector L1
        for (int v = 0; v < V; ++v) {
            const float* Sv = St + gm_opaque_zero();
            V64 dxv = gm_load_row(c, RW_DX, v);
            const unsigned r1 = __float_as_uint(gm_row_at(c, RW_R1, v)[c.row_lane]);      // sign bits of the stem's hidden layer
            GM_PIN();
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((int)am.t[t][r] == v) dxv.t[t][r] += dcur.t[t][r];
            V64 dr = gm_lin64_sj(w, lane, dxv, nullptr, nullptr);                  // rgbfeat_fc.2^T
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dr.t[t][r] = (r1 >> (16 * t + r)) & 1u ? dr.t[t][r] : 0.f;
            const V64 df = gm_lin64_sj(w, lane, dr, nullptr, v + 1 < V ? Sv : nullptr);      // rgbfeat_fc.0^T: 35 <- 64
            float* o = d_rgb_feat + ((ray * S + c.s) * V + v) * 35;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[gm_nidx(r, h)] = df.t[0][r];
            if (h == 0) {
                o[32] = df.t[1][0];
                o[33] = df.t[1][1];
                o[34] = df.t[1][2];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
template <int NW, bool TRAIN>
static int gm_launch(const float* mblob, const float* rgb_feat, const float* ray_diff, const float* mask, const float* pts,
                     const float* ray_d, int64_t n_rays, int V, int depth, int save, float* rgb, float* alpha, float* workspace,
                     GntDrop dr, const unsigned* seed_dev, hipStream_t st) {
    constexpr int S = 32 * NW;
    static bool configured_on[NF_MAX_DEVICES] = {};
    bool& configured = configured_on[nf_current_device()];
    const size_t smem = GmLds<NW>::FLOATS * sizeof(float);
    if (!configured && smem > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)k_gnt_fwd_mfma<NW, true, TRAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_gnt_fwd_mfma<NW, false, TRAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
            nf_set_error("nf_gnt_fwd_mfma: cannot reserve %zu bytes of LDS", smem);
            return 1;
        }
        configured = true;
    }
    const int64_t rf = gnt_row_floats(depth, save), sf = gnt_smp_floats(depth, save);
    const int64_t per_ray = (int64_t)S * V * rf + (int64_t)S * sf;
    const int64_t step = save ? n_rays : GNT_RAYS_PER_LAUNCH;
    for (int64_t r0 = 0; r0 < n_rays; r0 += step) {
        const int64_t nr = n_rays - r0 < step ? n_rays - r0 : step;
        if (save)
            hipLaunchKernelGGL((k_gnt_fwd_mfma<NW, true, TRAIN>), dim3((unsigned)nr), dim3(64 * NW), smem, st, mblob, rgb_feat + r0 * S * V * 35,
                               ray_diff + r0 * S * V * 4, mask + r0 * S * V, pts + r0 * S * 3, ray_d + r0 * 3, V, depth, rgb + r0 * 3,
                               workspace + r0 * per_ray, rf, sf, alpha ? alpha + r0 * S : nullptr, dr, seed_dev, r0);
        else
            hipLaunchKernelGGL((k_gnt_fwd_mfma<NW, false, TRAIN>), dim3((unsigned)nr), dim3(64 * NW), smem, st, mblob, rgb_feat + r0 * S * V * 35,
                               ray_diff + r0 * S * V * 4, mask + r0 * S * V, pts + r0 * S * 3, ray_d + r0 * 3, V, depth, rgb + r0 * 3,
                               workspace, rf, sf, alpha ? alpha + r0 * S : nullptr, dr, seed_dev, r0);
        NF_LAUNCH_CHECK("nf_gnt_fwd_mfma");
    }
    return 0;
}

template <bool TRAIN>
static int gm_fwd_entry(const char* name, const float* mfma_blob, const float* rgb_feat, const float* ray_diff, const float* mask,
                        const float* pts, const float* ray_d, int64_t n_rays, int n_samples, int n_views, int depth, int save,
                        float* rgb, float* alpha, float* workspace, GntDrop dr, const unsigned* seed_dev, nf_stream_t stream) {
    NF_REQUIRE(nf_gnt_mfma_supported(n_samples, n_views) && depth >= 1 && depth <= 16 && n_rays >= 0,
               "%s: S must be 32, 64, 96 or 128 and 1 <= V <= 64 (got S %d V %d depth %d)", name, n_samples, n_views, depth);
    if (n_rays == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    switch (n_samples / 32) {
        case 1: return gm_launch<1, TRAIN>(mfma_blob, rgb_feat, ray_diff, mask, pts, ray_d, n_rays, n_views, depth, save, rgb, alpha, workspace, dr, seed_dev, st);
        case 2: return gm_launch<2, TRAIN>(mfma_blob, rgb_feat, ray_diff, mask, pts, ray_d, n_rays, n_views, depth, save, rgb, alpha, workspace, dr, seed_dev, st);
        case 3: return gm_launch<3, TRAIN>(mfma_blob, rgb_feat, ray_diff, mask, pts, ray_d, n_rays, n_views, depth, save, rgb, alpha, workspace, dr, seed_dev, st);
        default: return gm_launch<4, TRAIN>(mfma_blob, rgb_feat, ray_diff, mask, pts, ray_d, n_rays, n_views, depth, save, rgb, alpha, workspace, dr, seed_dev, st);
    }
}

/* Same arguments as nf_gnt_fwd (workspace of nf_gnt_workspace_floats, consumed by nf_gnt_bwd_mfma when save != 0); weights in
 * the layout of nf_gnt_pack_mfma.  S must satisfy nf_gnt_mfma_supported. */
extern "C" int nf_gnt_fwd_mfma(const float* mfma_blob, const float* rgb_feat, const float* ray_diff, const float* mask,
                               const float* pts, const float* ray_d, int64_t n_rays, int n_samples, int n_views, int depth, int save,
                               float* rgb, float* alpha, float* workspace, nf_stream_t stream) {
    return gm_fwd_entry<false>("nf_gnt_fwd_mfma", mfma_blob, rgb_feat, ray_diff, mask, pts, ray_d, n_rays, n_samples, n_views, depth, save, rgb,
                               alpha, workspace, gnt_drop_make(0, 0, 0.0), nullptr, stream);
}

/* nf_gnt_fwd_mfma in TRAINING mode (round 6): the eight Dropout(p) sites of every layer live, the masks of nf_gnt_fwd_train (nf_gnt.h:
 * gnt_keep).  seed_dev (device pointer, nullable): the seed is read from that word instead of `seed` -- a captured PGD step reads the
 * seed of each replay from a buffer its host refreshes.  nf_gnt_bwd_train_mfma must be given the same seed / word and p. */
extern "C" int nf_gnt_fwd_train_mfma(const float* mfma_blob, const float* rgb_feat, const float* ray_diff, const float* mask,
                                     const float* pts, const float* ray_d, int64_t n_rays, int n_samples, int n_views, int depth, int save,
                                     float* rgb, float* alpha, float* workspace, uint32_t seed, double p, const uint32_t* seed_dev,
                                     nf_stream_t stream) {
    NF_REQUIRE(p >= 0.0 && p < 1.0, "nf_gnt_fwd_train_mfma: dropout rate %g outside [0, 1)", p);
    return gm_fwd_entry<true>("nf_gnt_fwd_train_mfma", mfma_blob, rgb_feat, ray_diff, mask, pts, ray_d, n_rays, n_samples, n_views, depth, save,
                              rgb, alpha, workspace, gnt_drop_make(1, seed, p), seed_dev, stream);
}

template <int NW, bool TRAIN>
static int gm_launch_bwd(const float* mblob, const float* mask, const float* d_rgb, int64_t n_rays, int V, int depth,
                         float* d_rgb_feat, float* workspace, GntDrop dr, const unsigned* seed_dev, hipStream_t st) {
    static bool configured_on[NF_MAX_DEVICES] = {};
    bool& configured = configured_on[nf_current_device()];
    const size_t smem = GmBwdLds<NW>::FLOATS * sizeof(float);
    if (!configured && smem > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)k_gnt_bwd_mfma<NW, TRAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
            nf_set_error("nf_gnt_bwd_mfma: cannot reserve %zu bytes of LDS", smem);
            return 1;
        }
        configured = true;
    }
    hipLaunchKernelGGL((k_gnt_bwd_mfma<NW, TRAIN>), dim3((unsigned)n_rays), dim3(64 * NW), smem, st, mblob, mask, d_rgb, V, depth, d_rgb_feat,
                       workspace, gnt_row_floats(depth, 1), gnt_smp_floats(depth, 1), dr, seed_dev);
    NF_LAUNCH_CHECK("nf_gnt_bwd_mfma");
    return 0;
}

template <bool TRAIN>
static int gm_bwd_entry(const char* name, const float* mfma_blob, const float* mask, const float* d_rgb, int64_t n_rays, int n_samples,
                        int n_views, int depth, float* d_rgb_feat, float* workspace, GntDrop dr, const unsigned* seed_dev, nf_stream_t stream) {
    NF_REQUIRE(nf_gnt_mfma_supported(n_samples, n_views) && depth >= 1 && depth <= 16 && n_rays >= 0,
               "%s: S must be 32, 64, 96 or 128 and 1 <= V <= 64 (got S %d V %d depth %d)", name, n_samples, n_views, depth);
    if (n_rays == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    switch (n_samples / 32) {
        case 1: return gm_launch_bwd<1, TRAIN>(mfma_blob, mask, d_rgb, n_rays, n_views, depth, d_rgb_feat, workspace, dr, seed_dev, st);
        case 2: return gm_launch_bwd<2, TRAIN>(mfma_blob, mask, d_rgb, n_rays, n_views, depth, d_rgb_feat, workspace, dr, seed_dev, st);
        case 3: return gm_launch_bwd<3, TRAIN>(mfma_blob, mask, d_rgb, n_rays, n_views, depth, d_rgb_feat, workspace, dr, seed_dev, st);
        default: return gm_launch_bwd<4, TRAIN>(mfma_blob, mask, d_rgb, n_rays, n_views, depth, d_rgb_feat, workspace, dr, seed_dev, st);
    }
}

/* Same arguments as nf_gnt_bwd; `workspace` is the buffer nf_gnt_fwd_mfma with save != 0 filled (not nf_gnt_fwd's). */
extern "C" int nf_gnt_bwd_mfma(const float* mfma_blob, const float* mask, const float* d_rgb, int64_t n_rays, int n_samples,
                               int n_views, int depth, float* d_rgb_feat, float* workspace, nf_stream_t stream) {
    return gm_bwd_entry<false>("nf_gnt_bwd_mfma", mfma_blob, mask, d_rgb, n_rays, n_samples, n_views, depth, d_rgb_feat, workspace,
                               gnt_drop_make(0, 0, 0.0), nullptr, stream);
}

/* backward of nf_gnt_fwd_train_mfma(save = 1, seed / seed_dev, p): regenerates the forward's masks */
extern "C" int nf_gnt_bwd_train_mfma(const float* mfma_blob, const float* mask, const float* d_rgb, int64_t n_rays, int n_samples,
                                     int n_views, int depth, float* d_rgb_feat, float* workspace, uint32_t seed, double p,
                                     const uint32_t* seed_dev, nf_stream_t stream) {
    NF_REQUIRE(p >= 0.0 && p < 1.0, "nf_gnt_bwd_train_mfma: dropout rate %g outside [0, 1)", p);
    return gm_bwd_entry<true>("nf_gnt_bwd_train_mfma", mfma_blob, mask, d_rgb, n_rays, n_samples, n_views, depth, d_rgb_feat, workspace,
                              gnt_drop_make(1, seed, p), seed_dev, stream);
}
