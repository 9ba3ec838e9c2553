// IBRNet on the matrix cores (a4/a5), any V <= 32: a sample's views occupy VP = the next power of two of V adjacent lanes (the
// reference's default num_source_views = 10 -> 16 lanes, 2 samples per 32-row tile); the padding lanes carry mask 0 and are neutral
// in every cross-view reduction (RowIn::pad).
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
#include <string.h>

#include "nf_ibrnet.h"
#include "nf_geometry.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// waves per workgroup of the row kernels (all waves of a workgroup share one LDS image of the weights)
#ifndef NF_ROWS_FWD_WAVES
#define NF_ROWS_FWD_WAVES 8      // 8 waves x 2 workgroups per CU = 4 waves per SIMD at <= 128 VGPRs (measured +10 % over 4 x 2)
#endif
#ifndef NF_ROWS_FWD_OCC
#define NF_ROWS_FWD_OCC 4
#endif
#ifndef NF_ROWS_BWD_WAVES
#define NF_ROWS_BWD_WAVES 4
#endif
#ifndef NF_ROWS_BWD_OCC
#define NF_ROWS_BWD_OCC 1
#endif

// makes a value opaque to the optimiser (the two-phase backward RECOMPUTES the early activations from the laundered inputs: without
// it the compiler proves them equal to the first evaluation and keeps those registers alive instead)
#if defined(__HIP_DEVICE_COMPILE__)
#define NF_LAUNDER_F(x) asm volatile("" : "+v"(x))
#define NF_PIN_FRAG(x) asm volatile("" : "+v"(x))        // a 16-register fragment: "computed by here" (no instruction)
#else
#define NF_LAUNDER_F(x) asm volatile("" : "+x"(x))
#define NF_PIN_FRAG(x) asm volatile("" ::: "memory")
#endif

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
    MS_RGB0V = MS_RGB2 + 12,         // [h][8] column 32 (the vis2 input) of rgb_fc.0 -- backward only
    MS_END = MS_RGB0V + 16,
    NF_MFMA_FWD_FLOATS = MS_END,     // the forward kernel stages [0, NF_MFMA_FWD_FLOATS)
    // ---- transposed records for the backward-data GEMMs dX^T[K x rows] = W^T[K x N] . dY^T[N x rows]: lane l holds
    //      W[n(step, l>>5)][k = 32*kt + (l&31)]
    MT_BASE = MS_END,
    MT_RGB0 = 0,                     // 8   (units: records, relative to MT_BASE)
    MT_VISB0 = MT_RGB0 + 8,          // 16
    MT_VIS1 = MT_VISB0 + 16,         // 16
    MT_VIS0 = MT_VIS1 + 16,          // 16
    MT_BASE1 = MT_VIS0 + 16,         // 2 k-tiles x 16
    MT_BASE0 = MT_BASE1 + 32,        // 4 output tiles (mean feat, var feat, f feat, the 9 colour inputs) x 2 n-blocks x 16
    MT_RECORDS = MT_BASE0 + 128,     // 216
    NF_ROWS_BLOB_FLOATS = MT_BASE + MT_RECORDS * 64,     // everything the row kernels stage
    // ---- per-ray kernels (geometry_fc, q/k/v, fc, out_geometry_fc on the matrix cores; sample = MFMA row)
    RY_BASE = NF_ROWS_BLOB_FLOATS,
    RYR_GEO0 = 0,                    // 2 tiles x (16 mean2 + 16 var2 + 1 wmean)        (units: records, relative to RY_BASE)
    RYR_GEO1 = RYR_GEO0 + 66,        // 2 k-blocks x 16 (16 outputs)
    RYR_QKV = RYR_GEO1 + 32,         // tile [q | k] 8 steps, tile [v | -] 8 steps
    RYR_FC = RYR_QKV + 16,           // 8
    RYR_OG0 = RYR_FC + 8,            // 8
    RYF_RECORDS = RYR_OG0 + 8,       // 130
    RYS_BIAS = RYF_RECORDS * 64,     // floats, relative to RY_BASE: 4 bias tiles [h][16]: GEO0 t0, GEO0 t1, GEO1, OG0
    RYS_LNW = RYS_BIAS + 4 * 32,     // [h][8] LayerNorm weight / bias in fragment order
    RYS_LNB = RYS_LNW + 16,
    RYS_OG1 = RYS_LNB + 16,          // [h][8] out_geometry_fc.2 weight, then its bias
    RYF_FLOATS = RYS_OG1 + 16 + 8,   // forward part
    RYT_BASE = RYF_FLOATS,           // transposed records (relative to RY_BASE, floats): OG0 8, FC 8, QKV 24, GEO1 16, GEO0 64
    RYT_OG0 = 0, RYT_FC = 8, RYT_QKV = 16, RYT_GEO1 = 40, RYT_GEO0 = 56, RYT_RECORDS = 120,
    RYS_GEO0W = RYT_BASE + RYT_RECORDS * 64,     // [2 tiles][h][16] column 64 (the wmean input) of geometry_fc.0
    RY_FLOATS = RYS_GEO0W + 64,
    // ---- "bf16x3" image of the forward row network (sample-on-the-lane kernels): every fp32 record value as three bf16 parts
    //      (hi + mid + lo, 8 significant bits each: the six cross products of order <= 2^-16 reproduce the fp32 product to fp32
    //      rounding level on the bf16 matrix pipe, which runs at 16x the fp32 matrix rate and beside the vector pipe).
    //      [part 3][group 33][lane 64][8 bf16], then a copy of the VALU tables [MS_BASE, MS_END)
    X3_BASE = RY_BASE + RY_FLOATS,
    X3_PART = 33 * 256,              // floats per part: BF_FWD_GROUPS groups of 256
    X3_TAB = 3 * X3_PART,            // the tables, relative to X3_BASE
    X3_FLOATS = X3_TAB + (MS_END - MS_BASE),
    NF_MFMA_BLOB_FLOATS = X3_BASE + X3_FLOATS
};
enum { BT_DIR0, BT_DIR1, BT_BASE0A, BT_BASE0B, BT_BASE1, BT_VIS0, BT_VIS1, BT_VISB0, BT_RGB0 };

#define NF_SMP_STRIDE 72   // per-sample record written by kernel A: mean2 32 | var2 32 | wmean | rgb 3 | nvalid | vsum | pad

// ---- bf16-operand variant of the row kernels (BASELINE config 5: "bf16 MFMA path") -----------------------------------
// v_mfma_f32_32x32x16_bf16: bf16 operands, fp32 accumulation, 16x the fp32 matrix rate.  Eight consecutive fp32 k-steps
// (one lane value each) become ONE bf16 k-step: lane (i, h) element j of a "group" = the fp32 record of step 8g + j, and
// registers 8g .. 8g+7 of an activation fragment, rounded pairwise to bf16, are its B operand (the accumulator of one layer
// is still a legal operand of the next: cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand").
// The group image is derived mechanically from the fp32 record image by nf_ibrnet_pack_mfma_bf16, so the two paths cannot
// drift apart.  Everything on the VALU (pooling, ELU, softmax, the small heads) and the per-ray kernels stay fp32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define NF_MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

struct BfSeg { int start, n; };       // a run of fp32 records consumed by one gemm call: ceil(n / 8) bf16 groups
__host__ __device__ constexpr BfSeg bf_fwd_seg(int i) {
    constexpr int b0 = MR_BASE0, b1 = MR_BASE0 + 54;
    constexpr BfSeg t[] = {{MR_DIR0, 2}, {MR_DIR1, 8},
                           {b0, 16}, {b0 + 16, 2}, {b0 + 18, 16}, {b0 + 34, 2}, {b0 + 36, 16}, {b0 + 52, 2},
                           {b1, 16}, {b1 + 16, 2}, {b1 + 18, 16}, {b1 + 34, 2}, {b1 + 36, 16}, {b1 + 52, 2},
                           {MR_BASE1, 16}, {MR_BASE1 + 16, 16}, {MR_VIS0, 16}, {MR_VIS1, 16}, {MR_VISB0, 16},
                           {MR_RGB0, 16}, {MR_RGB0 + 16, 3}};
    return t[i];
}
#define BF_FWD_SEGS 21
__host__ __device__ constexpr BfSeg bf_bwd_seg(int i) {       // record indices relative to MT_BASE
    constexpr BfSeg t[] = {{MT_RGB0, 8}, {MT_VISB0, 16}, {MT_VIS1, 16}, {MT_VIS0, 16}, {MT_BASE1, 16}, {MT_BASE1 + 16, 16},
                           {MT_BASE0, 16}, {MT_BASE0 + 16, 16}, {MT_BASE0 + 32, 16}, {MT_BASE0 + 48, 16}, {MT_BASE0 + 64, 16},
                           {MT_BASE0 + 80, 16}, {MT_BASE0 + 96, 16}, {MT_BASE0 + 112, 16}};
    return t[i];
}
#define BF_BWD_SEGS 14
__host__ __device__ constexpr int bf_grp_fwd(int rec) {       // fp32 record index -> bf16 group index
    int g = 0;
    for (int i = 0; i < BF_FWD_SEGS; ++i) {
        BfSeg s = bf_fwd_seg(i);
        if (rec >= s.start && rec < s.start + s.n) return g + (rec - s.start) / 8;
        g += (s.n + 7) / 8;
    }
    return -1;
}
__host__ __device__ constexpr int bf_grp_bwd(int rec) {
    int g = 0;
    for (int i = 0; i < BF_BWD_SEGS; ++i) {
        BfSeg s = bf_bwd_seg(i);
        if (rec >= s.start && rec < s.start + s.n) return g + (rec - s.start) / 8;
        g += (s.n + 7) / 8;
    }
    return -1;
}
enum {
    BF_FWD_GROUPS = 33, BF_BWD_GROUPS = 27,
    BF_GROUP_FLOATS = 256,                                   // 64 lanes x 8 bf16 = 1 KB
    // LDS / blob image of the bf16 row kernels, in floats: forward groups at 0; the fp32 VALU tables at their usual offsets
    // [MS_BASE, MS_END); the transposed groups fill the gap in between and continue behind the tables
    BF_BWD_IN_GAP = (MS_BASE - BF_FWD_GROUPS * BF_GROUP_FLOATS) / BF_GROUP_FLOATS,          // 21
    NF_BF_FWD_FLOATS = MS_END,                               // what the forward kernel stages (58 KB, as the fp32 one)
    NF_BF_BLOB_FLOATS = MS_END + (BF_BWD_GROUPS - BF_BWD_IN_GAP) * BF_GROUP_FLOATS          // 64 KB: forward + backward
};
static_assert(bf_grp_fwd(MR_RGB0 + 16) == BF_FWD_GROUPS - 1, "forward group count");
static_assert(X3_PART == BF_FWD_GROUPS * BF_GROUP_FLOATS, "bf16x3 image: one part = the forward groups");
static_assert(bf_grp_bwd(MT_BASE0 + 120) == BF_BWD_GROUPS - 1, "backward group count");
static_assert(BF_BWD_IN_GAP >= 1 && BF_BWD_IN_GAP < BF_BWD_GROUPS && MS_END % 4 == 0, "bf16 image layout");
static_assert(MS_BIAS % 4 == 0 && MS_VIS1L % 4 == 0 && MS_VISB1 % 4 == 0 && RYF_FLOATS % 4 == 0 && RY_FLOATS % 4 == 0,
              "16-byte LDS reads of the VALU tables and of the per-ray K / V rows");
__host__ __device__ constexpr int bf_fwd_off(int g) { return g * BF_GROUP_FLOATS; }
__host__ __device__ constexpr int bf_bwd_off(int g) {
    return g < BF_BWD_IN_GAP ? (BF_FWD_GROUPS + g) * BF_GROUP_FLOATS : MS_END + (g - BF_BWD_IN_GAP) * BF_GROUP_FLOATS;
}

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

// transposed record block: lane (i, h), step r -> W[n = nbase + n(r,h)][k = kbase + i]   (W is [N][K])
static void emit_frag_block_T(float*& rec, const float* W, int N, int K, int kbase, int nbase, int nsteps) {
    for (int r = 0; r < nsteps; ++r, rec += 64)
        for (int lane = 0; lane < 64; ++lane) {
            int i = lane & 31, h = lane >> 5;
            int n = nbase + nf_nidx(r, h), k = kbase + i;
            rec[lane] = (n < N && k < K) ? W[(size_t)n * K + k] : 0.f;
        }
}

static void emit_bias_tile(float* dst, const float* b, int N, int base) {
    for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 16; ++r) {
            int n = base + nf_nidx(r, h);
            dst[h * 16 + r] = (n >= 0 && n - base < 32 && n < N) ? b[n] : 0.f;
        }
}

static uint16_t bf16_rne(float f) {       // round to nearest even, NaN kept a NaN (the blob holds finite weights)
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
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
    const float* W0 = nat + nf_lin_w(NF_L_RGB0);             // [16][37]
    for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 8; ++r) out[MS_RGB0V + h * 8 + r] = W0[nf_nidx(r, h) * 37 + 32];
    // ---- transposed records (backward)
    rec = out + MT_BASE;
    emit_frag_block_T(rec, nat + nf_lin_w(NF_L_RGB0), 16, 37, 0, 0, 8);
    emit_frag_block_T(rec, nat + nf_lin_w(NF_L_VISB0), 32, 32, 0, 0, 16);
    emit_frag_block_T(rec, nat + nf_lin_w(NF_L_VIS1), 32, 32, 0, 0, 16);     // rows 0..31 of the [33][32] matrix
    emit_frag_block_T(rec, nat + nf_lin_w(NF_L_VIS0), 32, 32, 0, 0, 16);
    for (int kt = 0; kt < 2; ++kt) emit_frag_block_T(rec, nat + nf_lin_w(NF_L_BASE1), 32, 64, kt * 32, 0, 16);
    W = nat + nf_lin_w(NF_L_BASE0);                          // [64][105]
    for (int part = 0; part < 3; ++part)
        for (int nb = 0; nb < 2; ++nb) emit_frag_block_T(rec, W, 64, 105, part * 35 + 3, nb * 32, 16);
    {
        static const int kcol[9] = {0, 1, 2, 35, 36, 37, 70, 71, 72};
        for (int nb = 0; nb < 2; ++nb)
            for (int r = 0; r < 16; ++r, rec += 64)
                for (int lane = 0; lane < 64; ++lane) {
                    int i = lane & 31, h = lane >> 5, n = nb * 32 + nf_nidx(r, h);
                    rec[lane] = i < 9 ? W[(size_t)n * 105 + kcol[i]] : 0.f;
                }
    }
    if (rec - out != NF_ROWS_BLOB_FLOATS) return 2;
    // ================= per-ray kernels =================
    float* ry = out + RY_BASE;
    rec = ry;
    W = nat + nf_lin_w(NF_L_GEO0);                           // [64][65]
    for (int nt = 0; nt < 2; ++nt) {
        emit_frag_block(rec, W, 64, 65, 0, nt, 0, 16);
        emit_frag_block(rec, W, 64, 65, 0, nt, 32, 16);
        emit_record(rec, W, 64, 65, 0, nt, 64, -1); rec += 64;
    }
    W = nat + nf_lin_w(NF_L_GEO1);                           // [16][64]
    emit_frag_block(rec, W, 16, 64, 0, 0, 0, 16);
    emit_frag_block(rec, W, 16, 64, 0, 0, 32, 16);
    {
        float qk[32 * 16];                                   // rows 0..15 = w_qs, rows 16..31 = w_ks
        for (int i = 0; i < 256; ++i) { qk[i] = nat[nf_att_w(0) + i]; qk[256 + i] = nat[nf_att_w(1) + i]; }
        emit_frag_block(rec, qk, 32, 16, 0, 0, 0, 8);
        emit_frag_block(rec, nat + nf_att_w(2), 16, 16, 0, 0, 0, 8);
        emit_frag_block(rec, nat + nf_att_w(3), 16, 16, 0, 0, 0, 8);             // fc
        emit_frag_block(rec, nat + nf_lin_w(NF_L_OG0), 16, 16, 0, 0, 0, 8);
        if (rec - ry != RYF_RECORDS * 64) return 3;
        emit_bias_tile(ry + RYS_BIAS + 0 * 32, nat + nf_lin_b(NF_L_GEO0), 32, 0);
        emit_bias_tile(ry + RYS_BIAS + 1 * 32, nat + nf_lin_b(NF_L_GEO0) + 32, 32, 0);
        emit_bias_tile(ry + RYS_BIAS + 2 * 32, nat + nf_lin_b(NF_L_GEO1), 16, 0);
        emit_bias_tile(ry + RYS_BIAS + 3 * 32, nat + nf_lin_b(NF_L_OG0), 16, 0);
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 8; ++r) {
                ry[RYS_LNW + h * 8 + r] = nat[NF_LN_W + nf_nidx(r, h)];
                ry[RYS_LNB + h * 8 + r] = nat[NF_LN_B + nf_nidx(r, h)];
                ry[RYS_OG1 + h * 8 + r] = nat[nf_lin_w(NF_L_OG1) + nf_nidx(r, h)];
            }
        ry[RYS_OG1 + 16] = nat[nf_lin_b(NF_L_OG1)];
        // transposed records
        rec = ry + RYT_BASE;
        emit_frag_block_T(rec, nat + nf_lin_w(NF_L_OG0), 16, 16, 0, 0, 8);
        emit_frag_block_T(rec, nat + nf_att_w(3), 16, 16, 0, 0, 8);
        emit_frag_block_T(rec, qk, 32, 16, 0, 0, 16);
        emit_frag_block_T(rec, nat + nf_att_w(2), 16, 16, 0, 0, 8);
        emit_frag_block_T(rec, nat + nf_lin_w(NF_L_GEO1), 16, 64, 0, 0, 8);
        emit_frag_block_T(rec, nat + nf_lin_w(NF_L_GEO1), 16, 64, 32, 0, 8);
        W = nat + nf_lin_w(NF_L_GEO0);
        for (int kt = 0; kt < 2; ++kt)
            for (int nb = 0; nb < 2; ++nb) emit_frag_block_T(rec, W, 64, 65, kt * 32, nb * 32, 16);
        if (rec - (ry + RYT_BASE) != RYT_RECORDS * 64) return 4;
        for (int t = 0; t < 2; ++t)
            for (int h = 0; h < 2; ++h)
                for (int r = 0; r < 16; ++r) ry[RYS_GEO0W + t * 32 + h * 16 + r] = W[(size_t)(t * 32 + nf_nidx(r, h)) * 65 + 64];
    }
    // ================= bf16x3 image of the forward records =================
    {
        float* x3 = out + X3_BASE;
        int g0 = 0;
        for (int i = 0; i < BF_FWD_SEGS; ++i) {
            const BfSeg sg = bf_fwd_seg(i);
            for (int g = 0; g < (sg.n + 7) / 8; ++g)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int st = 8 * g + j;
                        float rem = st < sg.n ? out[(size_t)(sg.start + st) * 64 + lane] : 0.f;
                        for (int p = 0; p < 3; ++p) {
                            const uint16_t b = bf16_rne(rem);
                            reinterpret_cast<uint16_t*>(x3 + (size_t)p * X3_PART + (size_t)(g0 + g) * 256)[lane * 8 + j] = b;
                            const uint32_t u = (uint32_t)b << 16;
                            float up;
                            memcpy(&up, &u, 4);
                            rem -= up;            // exact
                        }
                    }
            g0 += (sg.n + 7) / 8;
        }
        if (g0 != BF_FWD_GROUPS) return 5;
        for (int i = MS_BASE; i < MS_END; ++i) x3[X3_TAB + (i - MS_BASE)] = out[i];
    }
    return 0;
}

extern "C" int64_t nf_ibrnet_mfma_bf16_blob_floats(void) { return NF_BF_BLOB_FLOATS; }

/* mfma_blob: output of nf_ibrnet_pack_mfma (host memory); out: nf_ibrnet_mfma_bf16_blob_floats() floats (host memory) */
extern "C" int nf_ibrnet_pack_mfma_bf16(const float* mfma_blob, float* out) {
    for (int i = 0; i < NF_BF_BLOB_FLOATS; ++i) out[i] = 0.f;
    for (int i = MS_BASE; i < MS_END; ++i) out[i] = mfma_blob[i];
    auto pack = [&](const float* recs, BfSeg sg, int g0, bool bwd) {
        for (int g = 0; g < (sg.n + 7) / 8; ++g) {
            uint16_t* dst = (uint16_t*)(out + (bwd ? bf_bwd_off(g0 + g) : bf_fwd_off(g0 + g)));
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    int st = 8 * g + j;
                    dst[lane * 8 + j] = st < sg.n ? bf16_rne(recs[(size_t)(sg.start + st) * 64 + lane]) : (uint16_t)0;
                }
        }
    };
    int g = 0;
    for (int i = 0; i < BF_FWD_SEGS; ++i) { pack(mfma_blob, bf_fwd_seg(i), g, false); g += (bf_fwd_seg(i).n + 7) / 8; }
    if (g != BF_FWD_GROUPS) return 1;
    g = 0;
    for (int i = 0; i < BF_BWD_SEGS; ++i) { pack(mfma_blob + MT_BASE, bf_bwd_seg(i), g, true); g += (bf_bwd_seg(i).n + 7) / 8; }
    return g == BF_BWD_GROUPS ? 0 : 2;
}

// ---------------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------------
// v_exp_f32-based exponentials (2 instructions, ~2 ulp) instead of the ~20-30 instruction libm forms: on the row kernels
// the ~140 ELUs per lane per tile would otherwise cost more VALU cycles than the tile's 217 MFMAs take
__device__ __forceinline__ float mf_exp(float x) { return __expf(x); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 nf_exp2x2(f32x2 x) {      // 2^x per element
#if defined(__HIP_DEVICE_COMPILE__)
    return f32x2{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
#else
    return f32x2{exp2f(x.x), exp2f(x.y)};
#endif
}
__device__ __forceinline__ float nf_med3(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(a, b, c);
#else
    return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
#endif
}
// ELU as the median of (x, e^x - 1, 0): x above zero (e^x - 1 >= x > 0), e^x - 1 below (x <= e^x - 1 <= 0) -- one v_med3_f32 in
// place of a compare and a select.  (For 0 < x < ~3e-4 the rounded e^x - 1 can fall below x and is returned instead: within 1.2e-7.)
// The ELUs are 40 % of the row kernels' vector instructions.
__device__ __forceinline__ float mf_elu(float x) { return nf_med3(x, mf_exp(x) - 1.f, 0.f); }
__device__ __forceinline__ float mf_sigmoid(float x) { return 1.f / (1.f + mf_exp(-x)); }

// cross-view reductions of a sample: butterflies over its V adjacent lanes, as DPP modifiers of the combining instruction
// (nf_common.h) -- ~170 of them per 32-row tile, formerly as many LDS round trips
template <int V>
__device__ __forceinline__ float grp_sum(float x) { return nf_grp_reduce<V>(x, NfAdd()); }
template <int V>
__device__ __forceinline__ float grp_min(float x) { return nf_grp_reduce<V>(x, NfMin()); }
template <int V>
__device__ __forceinline__ float grp_max(float x) { return nf_grp_reduce<V>(x, NfMax()); }
__device__ __forceinline__ float half_sum(float x) { return nf_half_sum(x); }

// 16 consecutive floats of a VALU-side table as four 16-byte LDS reads (every table row starts on a multiple of 4 floats): as
// scalars they become ds_read2_b32 pairs with an address add each
__device__ __forceinline__ f32x16 lds_row16(const float* b) {
    f32x16 a;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(b + 4 * q);
        a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
    return a;
}
__device__ __forceinline__ f32x16 bias_tile(const float* lds, int tile, int h) { return lds_row16(lds + MS_BIAS + tile * 32 + h * 16); }

// A 16-register fragment <-> 32 consecutive floats in memory (feature n(r, h) at base[n]): registers 4 q .. 4 q + 3 of lane half h are the
// four CONSECUTIVE features 8 q + 4 h + (0 .. 3), so a fragment is four 16-byte accesses instead of sixteen 4-byte ones (base 16-byte
// aligned: the per-sample records are 288 bytes apart).  A lane's record is 288 bytes from its neighbour's: every access instruction touches
// 32 cache lines either way, so the instruction count is what the texture path pays for.
__device__ __forceinline__ f32x16 frag_load32(const float* base, int h) {
    f32x16 a;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(base + 8 * q + 4 * h);
        a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
    return a;
}
__device__ __forceinline__ void frag_store32(float* base, int h, const f32x16& a) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(base + 8 * q + 4 * h) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
}

__device__ __forceinline__ bf16x8 pack_bf8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
    bf16x8 b;
    b[0] = (__bf16)v0; b[1] = (__bf16)v1; b[2] = (__bf16)v2; b[3] = (__bf16)v3;
    b[4] = (__bf16)v4; b[5] = (__bf16)v5; b[6] = (__bf16)v6; b[7] = (__bf16)v7;
    return b;
}

// acc += W(records rec .. rec+NSTEPS-1) . x  (k-step r consumes register r of the fragment x); BF: one bf16 MFMA per 8 steps.
// `rec` must be a compile-time constant in the BF form (it is after unrolling: the group lookup folds away).
template <bool BF, int NSTEPS>
__device__ __forceinline__ f32x16 gemm_frag(const float* lds, int rec, int lane, const f32x16& x, f32x16 acc) {
    if constexpr (BF) {
        static_assert(NSTEPS % 8 == 0, "whole groups");
        const int g0 = bf_grp_fwd(rec);
#pragma unroll
        for (int g = 0; g < NSTEPS / 8; ++g) {
            const bf16x8 w = *(const bf16x8*)(lds + bf_fwd_off(g0 + g) + lane * 4);
            acc = NF_MFMA_BF(w, pack_bf8(x[8 * g], x[8 * g + 1], x[8 * g + 2], x[8 * g + 3], x[8 * g + 4], x[8 * g + 5],
                                         x[8 * g + 6], x[8 * g + 7]), acc);
        }
    } else {
#pragma unroll
        for (int r = 0; r < NSTEPS; ++r) acc = NF_MFMA(lds[(rec + r) * 64 + lane], x[r], acc);
    }
    return acc;
}

// the same for N <= 8 loose per-lane values (inputs that are not a fragment: ray_diff, colour statistics)
template <bool BF, int N>
__device__ __forceinline__ f32x16 gemm_small(const float* lds, int rec, int lane, const float (&v)[N], f32x16 acc) {
    if constexpr (BF) {
        float p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = j < N ? v[j] : 0.f;
        const bf16x8 w = *(const bf16x8*)(lds + bf_fwd_off(bf_grp_fwd(rec)) + lane * 4);
        acc = NF_MFMA_BF(w, pack_bf8(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]), acc);
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) acc = NF_MFMA(lds[(rec + j) * 64 + lane], v[j], acc);
    }
    return acc;
}

// ---- operand forms of the sample-on-the-lane kernels: OP 0 = fp32 records on v_mfma_f32_32x32x2_f32, OP 1 = plain bf16 groups,
//      OP 3 = "bf16x3": weights AND activations as three bf16 parts, the six cross products of order <= 2^-16 on
//      v_mfma_f32_32x32x16_bf16 -- the dropped terms are <= 2^-24 of the product (fp32 rounding level), the matrix pipe is busy
//      6 / 16 of the fp32 form's time, and its instructions run BESIDE the vector instructions (the fp32 matrix instructions'
//      time adds to theirs: profiles/r03_probe_mfma_valu_overlap.txt, tools/experimental/probe_bf3.hip).
//      For OP 3 the kernel's `lds` pointer is the TABLE base (lds[MS_...] as in the other images); the parts sit in front of it.
#define X3_LDS_SHIFT (X3_TAB - MS_BASE)        // lds (table base) - shared-memory base, in floats
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// two fp32 values -> their bf16 roundings (one packed register) and, in place, the exact remainders
__device__ __forceinline__ unsigned x3_split_pair(float& x0, float& x1) { return nf_split_pair_bf16(x0, x1); }
// acc += W(group g) . (8 values), all six cross terms
__device__ __forceinline__ f32x16 x3_group(const float* lds, int g, int lane, float (&v)[8], f32x16 acc) {
    u32x4 b[3];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int p = 0; p < 3; ++p) b[p][q] = x3_split_pair(v[2 * q], v[2 * q + 1]);
    const float* rec = lds - X3_LDS_SHIFT + g * BF_GROUP_FLOATS + lane * 4;
    const u32x4 a0 = *reinterpret_cast<const u32x4*>(rec), a1 = *reinterpret_cast<const u32x4*>(rec + X3_PART),
                a2 = *reinterpret_cast<const u32x4*>(rec + 2 * X3_PART);
#define X3_PROD(a, bb) acc = NF_MFMA_BF(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bb), acc)
    X3_PROD(a0, b[0]);
    X3_PROD(a0, b[1]);
    X3_PROD(a1, b[0]);
    X3_PROD(a0, b[2]);
    X3_PROD(a2, b[0]);
    X3_PROD(a1, b[1]);
#undef X3_PROD
    return acc;
}
template <int OP, int NSTEPS>
__device__ __forceinline__ f32x16 sgemm_frag(const float* lds, int rec, int lane, const f32x16& x, f32x16 acc) {
    if constexpr (OP == 3) {
        static_assert(NSTEPS % 8 == 0, "whole groups");
        const int g0 = bf_grp_fwd(rec);
#pragma unroll
        for (int g = 0; g < NSTEPS / 8; ++g) {
            float v[8] = {x[8 * g], x[8 * g + 1], x[8 * g + 2], x[8 * g + 3], x[8 * g + 4], x[8 * g + 5], x[8 * g + 6], x[8 * g + 7]};
            acc = x3_group(lds, g0 + g, lane, v, acc);
        }
        return acc;
    } else {
        return gemm_frag<OP == 1, NSTEPS>(lds, rec, lane, x, acc);
    }
}
template <int OP, int N>
__device__ __forceinline__ f32x16 sgemm_small(const float* lds, int rec, int lane, const float (&v)[N], f32x16 acc) {
    if constexpr (OP == 3) {
        float p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = j < N ? v[j] : 0.f;
        return x3_group(lds, bf_grp_fwd(rec), lane, p, acc);
    } else {
        return gemm_small<OP == 1, N>(lds, rec, lane, v, acc);
    }
}

// ... and over a fragment the scaling by log2(e) and the "- 1" are packed (v_pk_mul_f32 / v_pk_add_f32: two elements per
// instruction), the same arithmetic per element as mf_elu
__device__ __forceinline__ f32x16 elu16(f32x16 a) {
    const f32x2 log2e = {1.44269502f, 1.44269502f};          // 0x3fb8aa3b: the constant __expf multiplies by
    const f32x2 one = {1.f, 1.f};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 x = {a[r], a[r + 1]};
        const f32x2 em = nf_exp2x2(x * log2e) - one;
        a[r] = nf_med3(x.x, em.x, 0.f);
        a[r + 1] = nf_med3(x.y, em.y, 0.f);
    }
    return a;
}

// dot of a fragment with an [h][16] VALU vector, summed over both lane halves (all 32 features of the row)
__device__ __forceinline__ float dot_frag16(const float* vec_h, const f32x16& x) {
    const f32x16 w = lds_row16(vec_h);
    float d = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) d = fmaf(w[r], x[r], d);
    return half_sum(d);
}

// ---------------------------------------------------------------------------------------------------------------
// kernel A: the per-(sample, view) row network.  rows_forward keeps every activation the backward needs in registers
// (the forward kernel only consumes the outputs, the rest is dead code there).
// ---------------------------------------------------------------------------------------------------------------
struct RowIn {
    f32x16 feat;                 // rgb_feat channels 3 + n(r,h)
    float c[3], rd[4], mk;       // clean colour taps, ray_diff, validity
    bool pad;                    // a padding lane (view index >= the real view count): mk = 0, all inputs 0
    float inv_nv;                // 1 / real view count
};

// lane -> (sample, view) of a 32-row tile when a sample's nv views sit on V = 2^k >= nv adjacent lanes
template <int V>
struct RowMap {
    int64_t sample, row;         // row = sample * nv + view: the index into rgb_feat / ray_diff / mask / d rgb_feat
    int v;
    bool live, pad;
    __device__ __forceinline__ RowMap(int64_t tile, int m, int nv, int64_t n_samples) {
        sample = tile * (32 / V) + m / V;
        v = m % V;
        pad = v >= nv;
        live = sample < n_samples && !pad;
        if (sample >= n_samples) sample = n_samples - 1;       // whole groups beyond the end recompute the last sample (nothing written)
        row = sample * nv + (pad ? 0 : v);
    }
};

__device__ __forceinline__ void pad_row(RowIn& in) {
#pragma unroll
    for (int r = 0; r < 16; ++r) in.feat[r] = 0.f;
    in.c[0] = in.c[1] = in.c[2] = 0.f;
    in.rd[0] = in.rd[1] = in.rd[2] = in.rd[3] = 0.f;
    in.mk = 0.f;
}

struct RowActs {
    f32x16 F, MEAN, VAR, H1a, H1b, H, V1, XV, X2, U, MEAN2, VAR2;
    float fc[3], mc[3], vc[3], r1[8], r2[8];
    float w, logit, sig1, vis1, sig2, vis2, vsum, w2, wmean, nval, beta, rgb[3];
};

template <int V>
__device__ __forceinline__ void load_row(const float* __restrict__ rgb_feat, const float* __restrict__ ray_diff,
                                         const float* __restrict__ mask, int64_t row, int h, RowIn& in) {
    const float* rf = rgb_feat + row * 35;
    const float* rd = ray_diff + row * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) in.feat[r] = rf[3 + nf_nidx(r, h)];
    in.c[0] = rf[0]; in.c[1] = rf[1]; in.c[2] = rf[2];
    in.rd[0] = rd[0]; in.rd[1] = rd[1]; in.rd[2] = rd[2]; in.rd[3] = rd[3];
    in.mk = mask[row];
}

// Projector.compute (ibrnet/projection.py:42-132) done by the row itself: the kernel reads the feature maps and the source
// images, rgb_feat / ray_diff never exist in memory (north-star design: the per-ray source features go from the maps to the
// registers of the lanes that consume them).  Same arithmetic, tap by tap, as k_project_gather_fwd, so the values are the ones the
// stand-alone gather would have written.  Feature maps channels-last (unit channel stride, 16-byte aligned pixel records).
struct RowGather {
    const float* xyz;         // [n_samples][3]
    const float* cam_ws;      // nf_camera_setup workspace
    const float* src_rgbs;    // [V][H][W][3]
    const float* featmap;     // [V][32][Hf][Wf] through fs_v / fs_h / fs_w (channel stride 1)
    float* mask_out;          // [n_samples * V]: the validity mask (the caller's pixel mask needs it)
    int64_t fs_v, fs_h, fs_w;
    int H, W, Hf, Wf;
};

__device__ __forceinline__ void load_row_gather(const RowGather& g, int nv, int64_t row, int64_t sample, int v, int h, bool live, RowIn& in) {
    const float* cam = g.cam_ws + (int64_t)v * NF_CAM_STRIDE;
    const float* qc = g.cam_ws + (int64_t)nv * NF_CAM_STRIDE;
    const float x = g.xyz[sample * 3 + 0], y = g.xyz[sample * 3 + 1], z = g.xyz[sample * 3 + 2];
    float px, py;
    bool front;
    nf_project_point(cam, x, y, z, px, py, front);
    const NfTaps tf = nf_bilinear_taps(px, py, qc[0], qc[1], g.Hf, g.Wf);
    // Branch-free taps: a tap outside the map carries weight 0 (nf_bilinear_taps), so it is read from the nearest in-map position
    // and adds val * 0 -- the same sum as skipping it, without 4 + 4 divergent branches per row.  Addresses are a wave-uniform base
    // plus a 32-bit element offset (the maps of one level hold < 2^31 floats: checked by the host wrapper).
    {
        const int xa = min(max(tf.x0, 0), g.Wf - 1), xb = min(max(tf.x0 + 1, 0), g.Wf - 1);
        const int ya = min(max(tf.y0, 0), g.Hf - 1), yb = min(max(tf.y0 + 1, 0), g.Hf - 1);
        const unsigned lane_off = (unsigned)(v * (int)g.fs_v + 4 * h);        // this lane half's channels: 8 q + 4 h + (0 .. 3)
        const unsigned off[4] = {lane_off + (unsigned)(ya * (int)g.fs_h + xa * (int)g.fs_w), lane_off + (unsigned)(ya * (int)g.fs_h + xb * (int)g.fs_w),
                                 lane_off + (unsigned)(yb * (int)g.fs_h + xa * (int)g.fs_w), lane_off + (unsigned)(yb * (int)g.fs_h + xb * (int)g.fs_w)};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 val = *reinterpret_cast<const float4*>(g.featmap + (off[t] + 8u * q));
                acc.x = acc.x + val.x * tf.w[t];
                acc.y = acc.y + val.y * tf.w[t];
                acc.z = acc.z + val.z * tf.w[t];
                acc.w = acc.w + val.w * tf.w[t];
            }
            in.feat[4 * q + 0] = acc.x;
            in.feat[4 * q + 1] = acc.y;
            in.feat[4 * q + 2] = acc.z;
            in.feat[4 * q + 3] = acc.w;
        }
    }
    const NfTaps ti = nf_bilinear_taps(px, py, qc[0], qc[1], g.H, g.W);
    float r = 0.f, gg = 0.f, b = 0.f;
    {
        const int xa = min(max(ti.x0, 0), g.W - 1), xb = min(max(ti.x0 + 1, 0), g.W - 1);
        const int ya = min(max(ti.y0, 0), g.H - 1), yb = min(max(ti.y0 + 1, 0), g.H - 1);
        const unsigned img = (unsigned)(v * g.H * g.W);
        const unsigned off[4] = {3u * (img + (unsigned)(ya * g.W + xa)), 3u * (img + (unsigned)(ya * g.W + xb)),
                                 3u * (img + (unsigned)(yb * g.W + xa)), 3u * (img + (unsigned)(yb * g.W + xb))};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float* p = g.src_rgbs + off[t];
            r = r + p[0] * ti.w[t];
            gg = gg + p[1] * ti.w[t];
            b = b + p[2] * ti.w[t];
        }
    }
    in.c[0] = r; in.c[1] = gg; in.c[2] = b;
    nf_ray_diff(qc + 12, cam + 12, x, y, z, in.rd);
    in.mk = (nf_inbound(px, py, qc[0], qc[1]) && front) ? 1.f : 0.f;
    if (live && h == 0 && g.mask_out) g.mask_out[row] = in.mk;
}

// base_fc.0 (105 -> 64), output tile NT: the [mean | var | f] feature blocks of 32 inputs and their 3 colour inputs each
template <bool BF, int NT>
__device__ __forceinline__ f32x16 base0_tile(const float* lds, int lane, int h, const RowActs& a) {
    constexpr int rec = MR_BASE0 + NT * 54;
    f32x16 acc = bias_tile(lds, BT_BASE0A + NT, h);
    acc = gemm_frag<BF, 16>(lds, rec, lane, a.MEAN, acc);
    const float m2[2] = {h ? a.mc[1] : a.mc[0], h ? 0.f : a.mc[2]};
    acc = gemm_small<BF, 2>(lds, rec + 16, lane, m2, acc);
    acc = gemm_frag<BF, 16>(lds, rec + 18, lane, a.VAR, acc);
    const float v2[2] = {h ? a.vc[1] : a.vc[0], h ? 0.f : a.vc[2]};
    acc = gemm_small<BF, 2>(lds, rec + 34, lane, v2, acc);
    acc = gemm_frag<BF, 16>(lds, rec + 36, lane, a.F, acc);
    const float f2[2] = {h ? a.fc[1] : a.fc[0], h ? 0.f : a.fc[2]};
    acc = gemm_small<BF, 2>(lds, rec + 52, lane, f2, acc);
    return acc;
}

// rows_forward = rows_forward_early (direction MLP, first pooling, base_fc.0: F, MEAN, VAR, H1a, H1b and the pooling weight w) followed by
// rows_forward_late (everything behind base_fc.0).  The two-phase backward re-runs the early half instead of keeping its 80
// registers of activations alive across the late half of the backward.
template <int V, bool BF>
__device__ __forceinline__ void rows_forward_early(const float* lds, int lane, int h, int aa, const RowIn& in, RowActs& a) {
    const float s_abs = lds[MS_RGB2 + 9];
    // ---- direction MLP 4 -> 16 -> 35, f = rgb_feat + dir_feat   (mlp_network.py:231-233)
    f32x16 d1 = bias_tile(lds, BT_DIR0, h);
    {
        const float v2[2] = {h ? in.rd[1] : in.rd[0], h ? in.rd[3] : in.rd[2]};
        d1 = gemm_small<BF, 2>(lds, MR_DIR0, lane, v2, d1);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) d1[r] = mf_elu(d1[r]);
    {
        const f32x16 df = elu16(gemm_frag<BF, 8>(lds, MR_DIR1, lane, d1, bias_tile(lds, BT_DIR1, h)));
#pragma unroll
        for (int r = 0; r < 16; ++r) a.F[r] = in.feat[r] + df[r];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* wv = lds + MS_DIR1C + c * 16 + h * 8;
        const float4 wa = *reinterpret_cast<const float4*>(wv), wb = *reinterpret_cast<const float4*>(wv + 4);
        const float wr[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
        float d = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) d = fmaf(wr[r], d1[r], d);
        d = half_sum(d) + lds[MS_DIR1C + 48 + c];
        a.fc[c] = in.c[c] + mf_elu(d);
    }
    // ---- first pooling weight (:234-241)
    float w;
    if (aa) {
        float e = in.pad ? 3.0e38f : mf_exp(s_abs * (in.rd[3] - 1.f));      // padding lanes never are the minimum
        w = (e - grp_min<V>(e)) * in.mk;
    } else {
        w = in.mk;
    }
    w = w / (grp_sum<V>(w) + 1e-8f);
    a.w = w;
    // ---- weighted mean / variance over the V views of the sample (:144-149)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float mu = grp_sum<V>(a.F[r] * w);
        float d = a.F[r] - mu;
        a.MEAN[r] = mu;
        a.VAR[r] = grp_sum<V>(w * (d * d));
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a.mc[c] = grp_sum<V>(a.fc[c] * w);
        float d = a.fc[c] - a.mc[c];
        a.vc[c] = grp_sum<V>(w * (d * d));
    }
    // ---- base_fc.0 (105 -> 64) as two 32-output tiles, base_fc.2 (64 -> 32)
    a.H1a = elu16(base0_tile<BF, 0>(lds, lane, h, a));
    a.H1b = elu16(base0_tile<BF, 1>(lds, lane, h, a));
}

template <int V, bool BF>
__device__ __forceinline__ void rows_forward_late(const float* lds, int lane, int h, const RowIn& in, RowActs& a) {
    const float w = a.w;
    // ---- base_fc.2 (64 -> 32)
    {
        f32x16 acc = bias_tile(lds, BT_BASE1, h);
        acc = gemm_frag<BF, 16>(lds, MR_BASE1, lane, a.H1a, acc);
        acc = gemm_frag<BF, 16>(lds, MR_BASE1 + 16, lane, a.H1b, acc);
        a.H = elu16(acc);
    }
    // ---- vis_fc on h * w, residual; vis_fc2 on x2 * vis1   (:249-254)
    {
        f32x16 t;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = a.H[r] * w;
        a.V1 = elu16(gemm_frag<BF, 16>(lds, MR_VIS0, lane, t, bias_tile(lds, BT_VIS0, h)));
        a.XV = elu16(gemm_frag<BF, 16>(lds, MR_VIS1, lane, a.V1, bias_tile(lds, BT_VIS1, h)));
        a.logit = mf_elu(dot_frag16(lds + MS_VIS1L + h * 16, a.V1) + lds[MS_VIS1L + 32]);
        a.sig1 = mf_sigmoid(a.logit);
        a.vis1 = a.sig1 * in.mk;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            a.X2[r] = a.H[r] + a.XV[r];
            t[r] = a.X2[r] * a.vis1;
        }
        a.U = elu16(gemm_frag<BF, 16>(lds, MR_VISB0, lane, t, bias_tile(lds, BT_VISB0, h)));
        float z2 = dot_frag16(lds + MS_VISB1 + h * 16, a.U) + lds[MS_VISB1 + 32];
        a.sig2 = mf_sigmoid(z2);
        a.vis2 = a.sig2 * in.mk;
    }
    a.vsum = grp_sum<V>(a.vis2) + 1e-8f;
    a.w2 = a.vis2 / a.vsum;
    a.wmean = grp_sum<V>(a.w2) * in.inv_nv;
    a.nval = grp_sum<V>(in.mk);
    // ---- second pooling (:255-257)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float mu = grp_sum<V>(a.X2[r] * a.w2);
        float d = a.X2[r] - mu;
        a.MEAN2[r] = mu;
        a.VAR2[r] = grp_sum<V>(a.w2 * (d * d));
    }
    // ---- colour head: rgb_fc 37 -> 16 -> 8 -> 1, softmax over views, blend of the clean colours  (:268-273)
    float y;
    {
        f32x16 acc = gemm_frag<BF, 16>(lds, MR_RGB0, lane, a.X2, bias_tile(lds, BT_RGB0, h));
        const float v3[3] = {h ? in.rd[0] : a.vis2, h ? in.rd[2] : in.rd[1], h ? 0.f : in.rd[3]};
        acc = gemm_small<BF, 3>(lds, MR_RGB0 + 16, lane, v3, acc);
#pragma unroll
        for (int r = 0; r < 8; ++r) a.r1[r] = mf_elu(acc[r]);
        y = lds[MS_RGB2 + 8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* wv = lds + MS_RGB1 + j * 16 + h * 8;
            const float4 wa = *reinterpret_cast<const float4*>(wv), wb = *reinterpret_cast<const float4*>(wv + 4);
            const float wr[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t = fmaf(wr[r], a.r1[r], t);
            a.r2[j] = mf_elu(half_sum(t) + lds[MS_RGB1 + 128 + j]);
            y = fmaf(lds[MS_RGB2 + j], a.r2[j], y);
        }
    }
    if (in.mk == 0.f) y = in.pad ? -3.0e38f : -1e9f;      // masked_fill(-1e9) for real views; a padding lane gets weight 0 even
                                                          // when every real view is masked (softmax of equal logits = 1 / nv)
    float p = mf_exp(y - grp_max<V>(y));
    a.beta = p / grp_sum<V>(p);
#pragma unroll
    for (int c = 0; c < 3; ++c) a.rgb[c] = grp_sum<V>(a.beta * in.c[c]);
}

template <int V, bool BF>
__device__ __forceinline__ void rows_forward(const float* lds, int lane, int h, int aa, const RowIn& in, RowActs& a) {
    rows_forward_early<V, BF>(lds, lane, h, aa, in, a);
    rows_forward_late<V, BF>(lds, lane, h, in, a);
}

template <int V, bool BF, bool GATH>
__global__ void __launch_bounds__(64 * NF_ROWS_FWD_WAVES, NF_ROWS_FWD_OCC) k_ibr_rows_fwd(const float* __restrict__ wblob, const float* __restrict__ rgb_feat,
                                                         const float* __restrict__ ray_diff, const float* __restrict__ mask,
                                                         int64_t n_samples, int nv, int aa, float* __restrict__ smp, RowGather gather) {
    HIP_DYNAMIC_SHARED(float, lds)
    static_assert(NF_BF_FWD_FLOATS == NF_MFMA_FWD_FLOATS, "both forward images are MS_END floats");
    for (int i = threadIdx.x; i < NF_MFMA_FWD_FLOATS; i += blockDim.x) lds[i] = wblob[i];      // BF: wblob is the bf16 image
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (n_samples + 32 / V - 1) / (32 / V);
    for (int64_t tile = (int64_t)blockIdx.x * NF_ROWS_FWD_WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * NF_ROWS_FWD_WAVES) {
        // compiler barrier: keeps the (tile-invariant) weight reads from being hoisted out of the loop into 200+ VGPRs
        asm volatile("" ::: "memory");
        const RowMap<V> rm(tile, m, nv, n_samples);
        const int64_t row = rm.row, sample = rm.sample;
        const int v = rm.v;
        const bool live = rm.live;
        RowIn in;
        in.pad = rm.pad;
        in.inv_nv = 1.f / (float)nv;
        if (rm.pad) pad_row(in);
        else if (GATH) load_row_gather(gather, nv, row, sample, v, h, live, in);
        else load_row<V>(rgb_feat, ray_diff, mask, row, h, in);
        RowActs a;
        rows_forward<V, BF>(lds, lane, h, aa, in, a);
        // per-sample record; the lane holding view 0 writes (both lane halves, 16 features each)
        float* out = smp + sample * NF_SMP_STRIDE;
        if (rm.live && v == 0) {
            frag_store32(out, h, a.MEAN2);
            frag_store32(out + 32, h, a.VAR2);
            if (h == 0) {
                *reinterpret_cast<float4*>(out + 64) = make_float4(a.wmean, a.rgb[0], a.rgb[1], a.rgb[2]);
                out[68] = a.nval;
                out[69] = a.vsum;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// kernel A, SAMPLE-ON-THE-LANE form ("sol"): a wave owns 32 SAMPLES -- the MFMA column is the sample, lane (m, h) holds half of
// sample m's 32 features -- and walks the V views of its samples IN TIME.  Against the row form above (column = (sample, view) row):
//   * the view-invariant part of base_fc.0 -- W[:, mean | var] . [mean | var], 70 of its 105 inputs (mlp_network.py:245-247: the
//     global feature is expanded over the views) -- is multiplied ONCE per sample instead of once per view: 72 of the row form's
//     217 matrix instructions per 32 rows become 72 per 32 SAMPLES (V = 4: 868 -> 652 per 128 rows);
//   * every cross-view reduction (pooling mean / variance, normalisations, the blending softmax) is in-lane accumulation over the
//     view loop: the ~170 DPP butterflies of a row tile are gone;
//   * any view count runs without padding lanes (the row form pads V to a power of two: 10 views on 16 lanes).
// What it costs: the views' features F_v stay in registers from the pooling to their base_fc.0 product (16 V registers), so the
// kernel runs at two waves per SIMD (V <= 4) or one (V <= 10) instead of four.  Same weight records as the row form.
// The variance of the SECOND pooling needs its mean first, and keeping x2_v of every view would cost another 16 V registers: it is
// accumulated around a pivot (x2 of view 0) instead,  var = sum w (d - mu)^2 = sum w d^2 - 2 mu sum w d + mu^2 sum w  with
// d = x2_v - pivot, mu = mean - pivot: all three terms are of the size of the view-to-view spread squared (no cancellation
// against the features' magnitude); the backward uses the exact deviations, it knows the mean when it re-walks the views.
// ---------------------------------------------------------------------------------------------------------------
// OCC = waves per SIMD the kernel is built for: 2 (eight-wave workgroups, <= 256 registers: V <= 4) or 1 (four-wave workgroups, <= 512)
__host__ __device__ constexpr int sol_fwd_waves(int occ) { return 4 * occ; }
#define NF_SOL_MAX_V 10

// direction MLP 4 -> 16 -> 35 and f = rgb_feat + dir_feat of one (sample, view)   (mlp_network.py:231-233)
template <int OP>
__device__ __forceinline__ void sol_direction(const float* lds, int lane, int h, const RowIn& in, f32x16& F, float (&fc)[3]) {
    f32x16 d1 = bias_tile(lds, BT_DIR0, h);
    {
        const float v2[2] = {h ? in.rd[1] : in.rd[0], h ? in.rd[3] : in.rd[2]};
        d1 = sgemm_small<OP, 2>(lds, MR_DIR0, lane, v2, d1);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) d1[r] = mf_elu(d1[r]);
    {
        const f32x16 df = elu16(sgemm_frag<OP, 8>(lds, MR_DIR1, lane, d1, bias_tile(lds, BT_DIR1, h)));
#pragma unroll
        for (int r = 0; r < 16; ++r) F[r] = in.feat[r] + df[r];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* wv = lds + MS_DIR1C + c * 16 + h * 8;
        const float4 wa = *reinterpret_cast<const float4*>(wv), wb = *reinterpret_cast<const float4*>(wv + 4);
        const float wr[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
        float d = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) d = fmaf(wr[r], d1[r], d);
        d = half_sum(d) + lds[MS_DIR1C + 48 + c];
        fc[c] = in.c[c] + mf_elu(d);
    }
}

// the view-invariant part of base_fc.0, output tile NT: bias + W[:, mean] . mean + W[:, var] . var   (once per sample)
template <int OP, int NT>
__device__ __forceinline__ f32x16 sol_base0_global(const float* lds, int lane, int h, const f32x16& MEAN, const float (&mc)[3],
                                                   const f32x16& VAR, const float (&vc)[3]) {
    constexpr int rec = MR_BASE0 + NT * 54;
    f32x16 acc = bias_tile(lds, BT_BASE0A + NT, h);
    acc = sgemm_frag<OP, 16>(lds, rec, lane, MEAN, acc);
    const float m2[2] = {h ? mc[1] : mc[0], h ? 0.f : mc[2]};
    acc = sgemm_small<OP, 2>(lds, rec + 16, lane, m2, acc);
    acc = sgemm_frag<OP, 16>(lds, rec + 18, lane, VAR, acc);
    const float v2[2] = {h ? vc[1] : vc[0], h ? 0.f : vc[2]};
    acc = sgemm_small<OP, 2>(lds, rec + 34, lane, v2, acc);
    return acc;
}
// ... and the per-view part on top of it: G + W[:, f] . f_v
template <int OP, int NT>
__device__ __forceinline__ f32x16 sol_base0_view(const float* lds, int lane, int h, const f32x16& G, const f32x16& F, const float (&fc)[3]) {
    constexpr int rec = MR_BASE0 + NT * 54;
    f32x16 acc = sgemm_frag<OP, 16>(lds, rec + 36, lane, F, G);
    const float f2[2] = {h ? fc[1] : fc[0], h ? 0.f : fc[2]};
    return sgemm_small<OP, 2>(lds, rec + 52, lane, f2, acc);
}

// everything of one (sample, view) behind base_fc.0 that does not look at the other views: base_fc.2, vis_fc, vis_fc2 and the
// colour head's logit (mlp_network.py:248-254, 268-270).  The backward re-walks the views through the same function.
struct SolViewActs {
    f32x16 H, V1, XV, X2, U;
    float r1[8], r2[8];
    float logit, sig1, vis1, sig2, vis2, y;
};
template <int OP>
__device__ __forceinline__ void sol_view_chain(const float* lds, int lane, int h, const f32x16& H1a, const f32x16& H1b, float w, float mk,
                                               const float (&rd)[4], SolViewActs& a) {
    {
        f32x16 acc = bias_tile(lds, BT_BASE1, h);
        acc = sgemm_frag<OP, 16>(lds, MR_BASE1, lane, H1a, acc);
        acc = sgemm_frag<OP, 16>(lds, MR_BASE1 + 16, lane, H1b, acc);
        a.H = elu16(acc);
    }
    {
        f32x16 t;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = a.H[r] * w;
        a.V1 = elu16(sgemm_frag<OP, 16>(lds, MR_VIS0, lane, t, bias_tile(lds, BT_VIS0, h)));
        a.XV = elu16(sgemm_frag<OP, 16>(lds, MR_VIS1, lane, a.V1, bias_tile(lds, BT_VIS1, h)));
        a.logit = mf_elu(dot_frag16(lds + MS_VIS1L + h * 16, a.V1) + lds[MS_VIS1L + 32]);
        a.sig1 = mf_sigmoid(a.logit);
        a.vis1 = a.sig1 * mk;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            a.X2[r] = a.H[r] + a.XV[r];
            t[r] = a.X2[r] * a.vis1;
        }
        a.U = elu16(sgemm_frag<OP, 16>(lds, MR_VISB0, lane, t, bias_tile(lds, BT_VISB0, h)));
        const float z2 = dot_frag16(lds + MS_VISB1 + h * 16, a.U) + lds[MS_VISB1 + 32];
        a.sig2 = mf_sigmoid(z2);
        a.vis2 = a.sig2 * mk;
    }
    // colour head: rgb_fc 37 -> 16 -> 8 -> 1
    {
        f32x16 acc = sgemm_frag<OP, 16>(lds, MR_RGB0, lane, a.X2, bias_tile(lds, BT_RGB0, h));
        const float v3[3] = {h ? rd[0] : a.vis2, h ? rd[2] : rd[1], h ? 0.f : rd[3]};
        acc = sgemm_small<OP, 3>(lds, MR_RGB0 + 16, lane, v3, acc);
#pragma unroll
        for (int r = 0; r < 8; ++r) a.r1[r] = mf_elu(acc[r]);
        float y = lds[MS_RGB2 + 8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* wv = lds + MS_RGB1 + j * 16 + h * 8;
            const float4 wa = *reinterpret_cast<const float4*>(wv), wb = *reinterpret_cast<const float4*>(wv + 4);
            const float wr[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t = fmaf(wr[r], a.r1[r], t);
            a.r2[j] = mf_elu(half_sum(t) + lds[MS_RGB1 + 128 + j]);
            y = fmaf(lds[MS_RGB2 + j], a.r2[j], y);
        }
        a.y = mk == 0.f ? -1e9f : y;          // masked_fill(mask == 0, -1e9)  (:271)
    }
}

// the per-view state that stays in registers across a tile.  KEEP (the gather-fused form, whose colour taps and ray differences are
// computed, not loaded): also the clean colour and the ray difference; otherwise those are read again where they are used (L1 / L2
// hits of lines the first read brought in) instead of holding 7 V registers across the view loops.
template <bool KEEP>
struct SolView {
    f32x16 F;
    float fc[3], mk, w, rd3;
    float c[KEEP ? 3 : 1], rd[KEEP ? 3 : 1];
};

// first pooling of a tile: weights (:234-241), weighted mean / variance over the views (:144-149) -- all in-lane
template <int V, bool KEEP>
__device__ __forceinline__ void sol_pool_first(const float* lds, int aa, SolView<KEEP> (&vw)[V], f32x16& MEAN, f32x16& VAR, float (&mc)[3],
                                               float (&vc)[3]) {
    if (aa) {
        const float s_abs = lds[MS_RGB2 + 9];
        float e[V];
        float mn = 3.0e38f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            e[v] = mf_exp(s_abs * (vw[v].rd3 - 1.f));
            mn = fminf(mn, e[v]);
        }
#pragma unroll
        for (int v = 0; v < V; ++v) vw[v].w = (e[v] - mn) * vw[v].mk;
    } else {
#pragma unroll
        for (int v = 0; v < V; ++v) vw[v].w = vw[v].mk;
    }
    float ws = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v) ws += vw[v].w;
    ws += 1e-8f;
#pragma unroll
    for (int v = 0; v < V; ++v) vw[v].w = vw[v].w / ws;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float mu = vw[0].F[r] * vw[0].w;
#pragma unroll
        for (int v = 1; v < V; ++v) mu = fmaf(vw[v].F[r], vw[v].w, mu);
        float var = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float d = vw[v].F[r] - mu;
            var = fmaf(vw[v].w * d, d, var);
        }
        MEAN[r] = mu;
        VAR[r] = var;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float mu = vw[0].fc[c] * vw[0].w;
#pragma unroll
        for (int v = 1; v < V; ++v) mu = fmaf(vw[v].fc[c], vw[v].w, mu);
        float var = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float d = vw[v].fc[c] - mu;
            var = fmaf(vw[v].w * d, d, var);
        }
        mc[c] = mu;
        vc[c] = var;
    }
}

template <int V, int OP, bool GATH, int OCC>
__global__ void __launch_bounds__(64 * sol_fwd_waves(OCC), OCC) k_ibr_sol_fwd(const float* __restrict__ wblob, const float* __restrict__ rgb_feat,
                                                         const float* __restrict__ ray_diff, const float* __restrict__ mask,
                                                         int64_t n_samples, int aa, float* __restrict__ smp, RowGather gather) {
    HIP_DYNAMIC_SHARED(float, smem_base)
    // OP 0 / 1: the fp32 / bf16 forward image (wblob points at it); OP 3: the bf16x3 section of the matrix-core blob, `lds` = its tables
    constexpr int IMG = OP == 3 ? (int)X3_FLOATS : (int)NF_MFMA_FWD_FLOATS;
    {
        const float* src = wblob + (OP == 3 ? (int)X3_BASE : 0);
        for (int i = threadIdx.x; i < IMG; i += blockDim.x) smem_base[i] = src[i];
    }
    const float* lds = smem_base + (OP == 3 ? (int)X3_LDS_SHIFT : 0);
    __syncthreads();
    constexpr int NWV = sol_fwd_waves(OCC);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (n_samples + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * NWV + wave; tile < n_tiles; tile += (int64_t)gridDim.x * NWV) {
        asm volatile("" ::: "memory");       // keeps the (tile-invariant) weight reads inside the loop
        int64_t sample = tile * 32 + m;
        const bool live = sample < n_samples;
        if (!live) sample = n_samples - 1;   // lanes beyond the end recompute the last sample (nothing written)
        // ---- per view: inputs, direction MLP, f_v.  (The memory clobber keeps the loads of view v + 1 behind the arithmetic of view
        //      v: hoisted to the top of the tile, the V views' taps would hold 76 V registers at once.)
        SolView<GATH> vw[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
#ifndef NF_SOL_NO_CLOB1
            asm volatile("" ::: "memory");
#endif
            RowIn in;
            in.pad = false;
            const int64_t row = sample * V + v;
            if (GATH) load_row_gather(gather, V, row, sample, v, h, live, in);
            else load_row<1>(rgb_feat, ray_diff, mask, row, h, in);
            sol_direction<OP>(lds, lane, h, in, vw[v].F, vw[v].fc);
            if constexpr (GATH) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    vw[v].c[c] = in.c[c];
                    vw[v].rd[c] = in.rd[c];
                }
            }
            vw[v].rd3 = in.rd[3];
            vw[v].mk = in.mk;
            // (pins the view's taps and direction MLP inside its iteration: f_v is only consumed by the pooling behind the loop, and
            //  the compiler otherwise keeps the raw taps of several views -- 64 registers each -- to do their arithmetic later)
            NF_PIN_FRAG(vw[v].F);
        }
        // ---- first pooling, the view-invariant part of base_fc.0
        f32x16 G0, G1;
        {
            f32x16 MEAN, VAR;
            float mc[3], vc[3];
            sol_pool_first<V, GATH>(lds, aa, vw, MEAN, VAR, mc, vc);
            G0 = sol_base0_global<OP, 0>(lds, lane, h, MEAN, mc, VAR, vc);
            G1 = sol_base0_global<OP, 1>(lds, lane, h, MEAN, mc, VAR, vc);
        }
        // ---- per view: the rest of the row network; second-pooling sums around the pivot x2 of view 0
        f32x16 piv, S1, S2;
        float vis2[V], y[V];
        float vs = 0.f, nval = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
#ifndef NF_SOL_NO_CLOB2
            // (without this clobber the compiler reads every weight record ONCE per tile and keeps it in a register for all V views:
            //  145 registers of A operands)
            asm volatile("" ::: "memory");
#endif
            SolViewActs a;
            {
                const f32x16 H1a = elu16(sol_base0_view<OP, 0>(lds, lane, h, G0, vw[v].F, vw[v].fc));
                const f32x16 H1b = elu16(sol_base0_view<OP, 1>(lds, lane, h, G1, vw[v].F, vw[v].fc));
                float rd[4];
                if constexpr (GATH) {
                    rd[0] = vw[v].rd[0]; rd[1] = vw[v].rd[1]; rd[2] = vw[v].rd[2];
                } else {
                    const float* p = ray_diff + (sample * V + v) * 4;
                    rd[0] = p[0]; rd[1] = p[1]; rd[2] = p[2];
                }
                rd[3] = vw[v].rd3;
                sol_view_chain<OP>(lds, lane, h, H1a, H1b, vw[v].w, vw[v].mk, rd, a);
            }
            vis2[v] = a.vis2;
            y[v] = a.y;
            // (pins the colour head of view v inside its iteration: its only consumer is the softmax behind the loop, and the compiler
            //  otherwise parks the 19 weight records and x2 of every view in scratch and runs the V heads at the end)
            NF_LAUNDER_F(y[v]);
            vs += a.vis2;
            nval += vw[v].mk;
            if (v == 0) {
                piv = a.X2;
#pragma unroll
                for (int r = 0; r < 16; ++r) S1[r] = S2[r] = 0.f;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = a.X2[r] - piv[r];
                    const float wd = a.vis2 * d;
                    S1[r] += wd;
                    S2[r] = fmaf(wd, d, S2[r]);
                }
            }
        }
        // ---- second pooling (:255-257): w2_v = vis2_v / (sum + 1e-8)
        const float vsum = vs + 1e-8f;
        const float inv_vsum = 1.f / vsum;
        float W = 0.f;                        // sum of the normalised weights (1, or 0 when no view is valid)
#pragma unroll
        for (int v = 0; v < V; ++v) W += vis2[v] * inv_vsum;
        const float wmean = W * (1.f / (float)V);
        f32x16 MEAN2, VAR2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m1 = S1[r] * inv_vsum;                 // sum w2 d
            const float mu = fmaf(piv[r], W - 1.f, m1);        // mean - pivot
            MEAN2[r] = mu + piv[r];
            VAR2[r] = fmaf(mu * mu, W, fmaf(-2.f * mu, m1, S2[r] * inv_vsum));
        }
        // ---- blending softmax over the views, blend of the clean colours (:271-273)
        float rgb[3] = {0.f, 0.f, 0.f};
        {
            float mx = y[0];
#pragma unroll
            for (int v = 1; v < V; ++v) mx = fmaxf(mx, y[v]);
            float p[V], ps = 0.f;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                p[v] = mf_exp(y[v] - mx);
                ps += p[v];
            }
            const float inv = 1.f / ps;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float beta = p[v] * inv;
                float col[3];
                if constexpr (GATH) {
                    col[0] = vw[v].c[0]; col[1] = vw[v].c[1]; col[2] = vw[v].c[2];
                } else {
                    const float* q = rgb_feat + (sample * V + v) * 35;
                    col[0] = q[0]; col[1] = q[1]; col[2] = q[2];
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) rgb[c] = fmaf(beta, col[c], rgb[c]);
            }
        }
        if (live) {
            float* out = smp + sample * NF_SMP_STRIDE;
            frag_store32(out, h, MEAN2);
            frag_store32(out + 32, h, VAR2);
            if (h == 0) {
                *reinterpret_cast<float4*>(out + 64) = make_float4(wmean, rgb[0], rgb[1], rgb[2]);
                out[68] = nval;
                out[69] = vsum;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// kernel A backward: d(per-sample record) -> d rgb_feat.  Follows oracle/ibrnet_manual_bwd.py; the backward-data GEMMs
// dX^T = W^T . dY^T use the transposed records and chain through registers exactly like the forward.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float mf_elu_grad(float y) { return y > 0.f ? 1.f : y + 1.f; }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

template <bool BF, int NSTEPS>
__device__ __forceinline__ f32x16 gemm_frag_T(const float* lds, int rec, int lane, const f32x16& dy, f32x16 acc) {
    if constexpr (BF) {
        static_assert(NSTEPS % 8 == 0, "whole groups");
        const int g0 = bf_grp_bwd(rec);
#pragma unroll
        for (int g = 0; g < NSTEPS / 8; ++g) {
            const bf16x8 w = *(const bf16x8*)(lds + bf_bwd_off(g0 + g) + lane * 4);
            acc = NF_MFMA_BF(w, pack_bf8(dy[8 * g], dy[8 * g + 1], dy[8 * g + 2], dy[8 * g + 3], dy[8 * g + 4], dy[8 * g + 5],
                                         dy[8 * g + 6], dy[8 * g + 7]), acc);
        }
    } else {
        const float* base = lds + MT_BASE;
#pragma unroll
        for (int r = 0; r < NSTEPS; ++r) acc = NF_MFMA(base[(rec + r) * 64 + lane], dy[r], acc);
    }
    return acc;
}

// d_feat[r] = d rgb_feat[row][3 + n(r,h)], d_col[c] = d rgb_feat[row][c]
// rows_backward = rows_backward_late (colour head, second pooling, vis_fc2, vis_fc: -> d h, the gradient w.r.t. base_fc's output BEFORE
// its ELU) followed by rows_backward_early (base_fc, first pooling: -> d rgb_feat); the late half reads only the late activations,
// the early half only F, MEAN, H1a, H1b, fc, mc, w, beta.
template <int V, bool BF>
__device__ __forceinline__ void rows_backward_late(const float* lds, int lane, int h, const RowIn& in, const RowActs& a,
                                                   const f32x16& d_mean2, const f32x16& d_var2, float d_wmean,
                                                   const float (&d_rgb)[3], f32x16& d_h) {
    // ---- colour head: blend softmax over views, rgb_fc 1 <- 8 <- 16 <- 37
    f32x16 d_x2;
    float d_vis2;
    {
        float d_beta = in.c[0] * d_rgb[0] + in.c[1] * d_rgb[1] + in.c[2] * d_rgb[2];
        float sbd = grp_sum<V>(a.beta * d_beta);
        float d_y = in.mk == 0.f ? 0.f : a.beta * (d_beta - sbd);      // masked_fill blocks the gradient
        float d_r2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) d_r2[j] = lds[MS_RGB2 + j] * d_y * mf_elu_grad(a.r2[j]);
        f32x16 d_r1 = zero16();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t = fmaf(lds[MS_RGB1 + j * 16 + h * 8 + r], d_r2[j], t);
            d_r1[r] = t * mf_elu_grad(a.r1[r]);
        }
        d_x2 = gemm_frag_T<BF, 8>(lds, MT_RGB0, lane, d_r1, zero16());
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t = fmaf(lds[MS_RGB0V + h * 8 + r], d_r1[r], t);
        d_vis2 = half_sum(t);
    }
    // ---- second pooling (weights depend on the features through vis2)
    {
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float dev = a.X2[r] - a.MEAN2[r];
            float s2 = grp_sum<V>(a.w2 * dev);
            float dm = d_mean2[r] + d_var2[r] * (-2.f * s2);
            d_x2[r] += a.w2 * (dm + 2.f * dev * d_var2[r]);
            part = fmaf(a.X2[r], dm, part);
            part = fmaf(dev * dev, d_var2[r], part);
        }
        float d_w2 = half_sum(part) + d_wmean * in.inv_nv;
        float sdw = grp_sum<V>(d_w2 * a.w2);
        d_vis2 += (d_w2 - sdw) / a.vsum;
    }
    // ---- vis_fc2 (1 <- 32 <- 32) on x2 * vis1
    float d_vis1;
    {
        float d_z2 = d_vis2 * in.mk * a.sig2 * (1.f - a.sig2);
        f32x16 d_u;
        const f32x16 wvb = lds_row16(lds + MS_VISB1 + h * 16);
#pragma unroll
        for (int r = 0; r < 16; ++r) d_u[r] = wvb[r] * d_z2 * mf_elu_grad(a.U[r]);
        f32x16 d_xvis = gemm_frag_T<BF, 16>(lds, MT_VISB0, lane, d_u, zero16());
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            d_x2[r] = fmaf(d_xvis[r], a.vis1, d_x2[r]);
            t = fmaf(d_xvis[r], a.X2[r], t);
        }
        d_vis1 = half_sum(t);
    }
    // ---- vis_fc (33 <- 32 <- 32) on h * w ; x2 = h + xv[:32]
    {
        float d_logit = d_vis1 * in.mk * a.sig1 * (1.f - a.sig1) * mf_elu_grad(a.logit);
        f32x16 d_xv;
#pragma unroll
        for (int r = 0; r < 16; ++r) d_xv[r] = d_x2[r] * mf_elu_grad(a.XV[r]);
        f32x16 d_v1 = gemm_frag_T<BF, 16>(lds, MT_VIS1, lane, d_xv, zero16());
        const f32x16 wv1 = lds_row16(lds + MS_VIS1L + h * 16);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            d_v1[r] = fmaf(wv1[r], d_logit, d_v1[r]) * mf_elu_grad(a.V1[r]);
        f32x16 d_t = gemm_frag_T<BF, 16>(lds, MT_VIS0, lane, d_v1, zero16());
#pragma unroll
        for (int r = 0; r < 16; ++r) d_h[r] = (d_x2[r] + d_t[r] * a.w) * mf_elu_grad(a.H[r]);
    }
}

template <int V, bool BF>
__device__ __forceinline__ void rows_backward_early(const float* lds, int lane, int h, const RowIn& in, const RowActs& a, const f32x16& d_h,
                                                    const float (&d_rgb)[3], f32x16& d_feat, float (&d_col)[3]) {
    // ---- base_fc (32 <- 64 <- 105)
    f32x16 g_mean, g_var, g_f, g_col;
    {
        f32x16 d_h1a = gemm_frag_T<BF, 16>(lds, MT_BASE1, lane, d_h, zero16());
        f32x16 d_h1b = gemm_frag_T<BF, 16>(lds, MT_BASE1 + 16, lane, d_h, zero16());
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            d_h1a[r] *= mf_elu_grad(a.H1a[r]);
            d_h1b[r] *= mf_elu_grad(a.H1b[r]);
        }
        g_mean = gemm_frag_T<BF, 16>(lds, MT_BASE0 + 16, lane, d_h1b, gemm_frag_T<BF, 16>(lds, MT_BASE0, lane, d_h1a, zero16()));
        g_var = gemm_frag_T<BF, 16>(lds, MT_BASE0 + 48, lane, d_h1b, gemm_frag_T<BF, 16>(lds, MT_BASE0 + 32, lane, d_h1a, zero16()));
        g_f = gemm_frag_T<BF, 16>(lds, MT_BASE0 + 80, lane, d_h1b, gemm_frag_T<BF, 16>(lds, MT_BASE0 + 64, lane, d_h1a, zero16()));
        g_col = gemm_frag_T<BF, 16>(lds, MT_BASE0 + 112, lane, d_h1b, gemm_frag_T<BF, 16>(lds, MT_BASE0 + 96, lane, d_h1a, zero16()));
    }
    // ---- first pooling (the weights w are constants): feature channels
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float dmean = grp_sum<V>(g_mean[r]);
        float dvar = grp_sum<V>(g_var[r]);
        float dev = a.F[r] - a.MEAN[r];
        float s1 = grp_sum<V>(a.w * dev);
        float dm = dmean + dvar * (-2.f * s1);
        d_feat[r] = g_f[r] + a.w * (dm + 2.f * dev * dvar);
    }
    // ---- ... and the 3 colour channels.  The colour tile holds rows 0..8 = d/d(mean c0..2, var c0..2, f c0..2):
    //      rows 0-3 and 8 sit in the low lane half (registers 0-3, 4), rows 4-7 in the high half (registers 0-3)
    {
        float lo0 = g_col[0], lo1 = g_col[1], lo2 = g_col[2], lo3 = g_col[3], lo4 = g_col[4];
        float o0 = nf_half_other(lo0), o1 = nf_half_other(lo1), o2 = nf_half_other(lo2), o3 = nf_half_other(lo3), o4 = nf_half_other(lo4);
        // value of tile row i for this (row, either half): low half owns rows {0,1,2,3,8}, high half rows {4,5,6,7}
        float row0 = h ? o0 : lo0, row1 = h ? o1 : lo1, row2 = h ? o2 : lo2, row3 = h ? o3 : lo3, row8 = h ? o4 : lo4;
        float row4 = h ? lo0 : o0, row5 = h ? lo1 : o1, row6 = h ? lo2 : o2, row7 = h ? lo3 : o3;
        float gm[3] = {row0, row1, row2}, gv[3] = {row3, row4, row5}, gf[3] = {row6, row7, row8};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float dmean = grp_sum<V>(gm[c]);
            float dvar = grp_sum<V>(gv[c]);
            float dev = a.fc[c] - a.mc[c];
            float s1 = grp_sum<V>(a.w * dev);
            float dm = dmean + dvar * (-2.f * s1);
            d_col[c] = gf[c] + a.w * (dm + 2.f * dev * dvar) + a.beta * d_rgb[c];
        }
    }
}

template <int V, bool BF>
__device__ __forceinline__ void rows_backward(const float* lds, int lane, int h, const RowIn& in, const RowActs& a,
                                              const f32x16& d_mean2, const f32x16& d_var2, float d_wmean,
                                              const float (&d_rgb)[3], f32x16& d_feat, float (&d_col)[3]) {
    f32x16 d_h;
    rows_backward_late<V, BF>(lds, lane, h, in, a, d_mean2, d_var2, d_wmean, d_rgb, d_h);
    rows_backward_early<V, BF>(lds, lane, h, in, a, d_h, d_rgb, d_feat, d_col);
}

// The scatter of d rgb_feat into the feature-map gradient (autograd of F.grid_sample, ibrnet/projection.py:120-121), fused into
// the row kernel's output stage: d rgb_feat never goes to memory (north-star design: no rgb_feat round trip on the way back) and
// the float atomics -- fire and forget -- drain while the matrix work of the next tile runs, where the stand-alone scatter kernel
// is bound by the chip-wide atomic rate with nothing else to do.
// wave-level ordering point for data handed from lane to lane through LDS (on the CPU stand-in of tests/host_harness, where lanes
// are threads, the collective is a real rendezvous)
__device__ __forceinline__ void rs_wave_sync() {
    asm volatile("" ::: "memory");
    (void)__shfl(0, 0, NF_WAVE);
    asm volatile("" ::: "memory");
}

struct RowScatter {
    const float* xyz;         // [n_samples][3] sample points
    const float* cam_ws;      // nf_camera_setup workspace: V source cameras + the query entry
    float* d_featmap;         // [V][C = 32][Hf][Wf] through the element strides below; nullptr: write d rgb_feat instead
    int64_t fs_v, fs_c, fs_h, fs_w;
    int Hf, Wf;
};
#ifndef NF_SCATTER_MERGE
#define NF_SCATTER_MERGE 1     // contributions of neighbouring samples to one bilinear cell summed before the atomics
#endif
#define RS_ROW 33             // floats per row of the wave's staging tile (32 channels, padded: conflict-free column reads)
#define RS_TAP 8              // per row: 4 tap weights, x0, y0 (as int bits), view, pad
#define RS_FLOATS (32 * RS_ROW + 32 * RS_TAP)

// Two waves per SIMD in every form of the backward row kernel.  fp32 rows (113 KB weight image, one workgroup per CU): 8-wave
// workgroups and the TWO-PHASE backward below, which fits 256 registers; bf16 rows (64 KB image): 4-wave workgroups x 2 per CU, or -- with
// the fused scatter, whose staging tiles leave no room for a second workgroup -- 8 waves x 1.
#ifndef NF_ROWS_BWD_TWO_PHASE
#define NF_ROWS_BWD_TWO_PHASE 1
#endif
__host__ __device__ constexpr bool rows_bwd_two_phase(bool bf) { return NF_ROWS_BWD_TWO_PHASE && !bf; }
__host__ __device__ constexpr int rows_bwd_waves(bool bf, bool scat) { return (rows_bwd_two_phase(bf) || (bf && scat)) ? 8 : NF_ROWS_BWD_WAVES; }
__host__ __device__ constexpr int rows_bwd_occ(bool bf, bool scat) { return (rows_bwd_two_phase(bf) || bf) ? 2 : NF_ROWS_BWD_OCC; }
__host__ __device__ constexpr int rows_bwd_wgs_per_cu(bool bf, bool scat) { return (bf && !scat) ? 2 : 1; }


template <int V, bool BF, bool SCAT, bool GATH>
__global__ void __launch_bounds__(64 * rows_bwd_waves(BF, SCAT), rows_bwd_occ(BF, SCAT)) k_ibr_rows_bwd(const float* __restrict__ wblob, const float* __restrict__ rgb_feat,
                                                         const float* __restrict__ ray_diff, const float* __restrict__ mask,
                                                         const float* __restrict__ d_smp, int64_t n_samples, int nv, int aa,
                                                         float* __restrict__ d_rgb_feat, RowScatter sc, RowGather gather) {
    HIP_DYNAMIC_SHARED(float, lds)
    for (int i = threadIdx.x; i < (BF ? (int)NF_BF_BLOB_FLOATS : (int)NF_ROWS_BLOB_FLOATS); i += blockDim.x) lds[i] = wblob[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (n_samples + 32 / V - 1) / (32 / V);
    constexpr int NWV = rows_bwd_waves(BF, SCAT);
    for (int64_t tile = (int64_t)blockIdx.x * NWV + wave; tile < n_tiles; tile += (int64_t)gridDim.x * NWV) {
        asm volatile("" ::: "memory");
        const RowMap<V> rm(tile, m, nv, n_samples);
        const int64_t row = rm.row, sample = rm.sample;
        const bool live = rm.live;
        RowIn in;
        in.pad = rm.pad;
        in.inv_nv = 1.f / (float)nv;
        if (rm.pad) pad_row(in);
        else if (GATH) load_row_gather(gather, nv, row, sample, rm.v, h, live, in);      // (a recompute that gathers again: experiments only)
        else load_row<V>(rgb_feat, ray_diff, mask, row, h, in);
        RowActs a;
        rows_forward<V, BF>(lds, lane, h, aa, in, a);
        if constexpr (rows_bwd_two_phase(BF)) asm volatile("" ::: "memory");       // the 35 gradient values are not fetched above the forward
        const float* g = d_smp + sample * NF_SMP_STRIDE;
        f32x16 d_mean2, d_var2;
        d_mean2 = frag_load32(g, h);
        d_var2 = frag_load32(g + 32, h);
        float d_rgb[3] = {g[65], g[66], g[67]};
        f32x16 d_feat;
        float d_col[3];
        if constexpr (rows_bwd_two_phase(BF)) {
            // late half of the backward on the late activations, then the early half on a RE-EVALUATION of the early activations
            // (direction MLP, first pooling, base_fc.0: 118 of the forward's 217 matrix instructions): F, MEAN, VAR, H1a, H1b need not
            // survive the late half, which is what takes the kernel from 391 registers to two waves per SIMD
            f32x16 d_h;
            rows_backward_late<V, BF>(lds, lane, h, in, a, d_mean2, d_var2, g[64], d_rgb, d_h);
            const float beta = a.beta;
            // the row's inputs are READ AGAIN (L2 hits) rather than kept in 24 registers across the late half; the memory clobber
            // keeps the compiler from re-using the first loads
            asm volatile("" ::: "memory");
            if (rm.pad) pad_row(in);
            else if (GATH) load_row_gather(gather, nv, row, sample, rm.v, h, live, in);
            else load_row<V>(rgb_feat, ray_diff, mask, row, h, in);
            RowActs e;
            rows_forward_early<V, BF>(lds, lane, h, aa, in, e);
            e.beta = beta;
            rows_backward_early<V, BF>(lds, lane, h, in, e, d_h, d_rgb, d_feat, d_col);
        } else {
            rows_backward<V, BF>(lds, lane, h, in, a, d_mean2, d_var2, g[64], d_rgb, d_feat, d_col);
        }
        if (SCAT) {
            // the wave's 32 rows x 32 channels pass through a private LDS tile so that lane = channel afterwards: one atomic
            // instruction then adds 2 rows x 128 contiguous bytes (as the stand-alone kernel does), not 64 scattered floats
            float* gt = lds + (BF ? (int)NF_BF_BLOB_FLOATS : (int)NF_ROWS_BLOB_FLOATS) + wave * RS_FLOATS;
            float* tp = gt + 32 * RS_ROW;
#pragma unroll
            for (int r = 0; r < 16; ++r) gt[m * RS_ROW + nf_nidx(r, h)] = d_feat[r];
            if (h == 0) {
                const int v = rm.pad ? 0 : rm.v;
                const float* cam = sc.cam_ws + (int64_t)v * NF_CAM_STRIDE;
                const float* qc = sc.cam_ws + (int64_t)nv * NF_CAM_STRIDE;
                float px, py;
                bool front;
                nf_project_point(cam, sc.xyz[sample * 3 + 0], sc.xyz[sample * 3 + 1], sc.xyz[sample * 3 + 2], px, py, front);
                const NfTaps tf = nf_bilinear_taps(px, py, qc[0], qc[1], sc.Hf, sc.Wf);
#pragma unroll
                for (int t = 0; t < 4; ++t) tp[m * RS_TAP + t] = (live && tf.in[t]) ? tf.w[t] : 0.f;       // 0 = nothing to add
                tp[m * RS_TAP + 4] = __builtin_bit_cast(float, (int)(tf.x0));
                tp[m * RS_TAP + 5] = __builtin_bit_cast(float, (int)(tf.y0));
                tp[m * RS_TAP + 6] = __builtin_bit_cast(float, (int)(v));
            }
            // the tile is private to the wave, whose lanes run in lockstep and whose LDS operations execute in order: rs_wave_sync
            // only keeps the compiler from moving the reads below above the writes of other lanes
            rs_wave_sync();
            // lane & 31 = channel; half-wave h walks the 32 / V consecutive samples of view 2 i + h: neighbours along a ray mostly fall
            // into the same bilinear cell of a source view, and their four tap contributions are summed before they go out as atomics
            // (fine samples of config 5: ~2.5 of the 4 samples of a view per cell)
#if NF_SCATTER_MERGE
#pragma unroll 1
            for (int i = 0; i < (V + 1) / 2; ++i) {
                const int vv = 2 * i + h;
                float accv[4] = {0.f, 0.f, 0.f, 0.f};
                int cx = 0, cy = 0, cv = 0, have = 0;         // `have`: taps of the open cell that received a contribution
                auto flush = [&]() {
                    float* fb = sc.d_featmap + (int64_t)cv * sc.fs_v + (int64_t)m * sc.fs_c;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (have & (1 << t)) atomicAdd(fb + (int64_t)(cy + (t >> 1)) * sc.fs_h + (int64_t)(cx + (t & 1)) * sc.fs_w, accv[t]);
                };
                if (vv < V) {
#pragma unroll 1
                    for (int sl = 0; sl < 32 / V; ++sl) {
                        const int j = sl * V + vv;                  // row of the tile
                        const float* tj = tp + j * RS_TAP;
                        int hit = 0;
#pragma unroll
                        for (int t = 0; t < 4; ++t) hit |= (tj[t] != 0.f) << t;
                        if (!hit) continue;                         // (uniform over the half-wave: a row's taps are the same for its 32 channels)
                        const int x0 = __builtin_bit_cast(int, (float)(tj[4])), y0 = __builtin_bit_cast(int, (float)(tj[5]));
                        if (have && (x0 != cx || y0 != cy)) {
                            flush();
                            have = 0;
                        }
                        if (!have) {
                            cx = x0, cy = y0, cv = __builtin_bit_cast(int, (float)(tj[6]));
#pragma unroll
                            for (int t = 0; t < 4; ++t) accv[t] = 0.f;
                        }
                        const float gv = gt[j * RS_ROW + m];
#pragma unroll
                        for (int t = 0; t < 4; ++t) accv[t] += gv * tj[t];
                        have |= hit;
                    }
                    if (have) flush();
                }
            }
#else
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int j = 2 * i + h;                    // row of the tile; lane & 31 = channel
                const float gv = gt[j * RS_ROW + m];
                const float* tj = tp + j * RS_TAP;
                const int x0 = __builtin_bit_cast(int, (float)(tj[4])), y0 = __builtin_bit_cast(int, (float)(tj[5])), v = __builtin_bit_cast(int, (float)(tj[6]));
                float* fb = sc.d_featmap + (int64_t)v * sc.fs_v + (int64_t)m * sc.fs_c;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float wt = tj[t];
                    if (wt != 0.f) atomicAdd(fb + (int64_t)(y0 + (t >> 1)) * sc.fs_h + (int64_t)(x0 + (t & 1)) * sc.fs_w, gv * wt);
                }
            }
#endif
            rs_wave_sync();         // ... and the next tile's writes below these reads
        } else if (live) {
            float* o = d_rgb_feat + row * 35;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[3 + nf_nidx(r, h)] = d_feat[r];
            if (h == 0) {
                o[0] = d_col[0];
                o[1] = d_col[1];
                o[2] = d_col[2];
            }
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
            float p = mf_exp(sc - mx);
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
// kernel B backward: d raw -> d(per-sample record).  Recomputes the per-sample forward (cheap), then walks back through
// the density head, LayerNorm, the ray attention (dQ by query thread, dK/dV by key thread) and geometry_fc.
// LDS per ray: Q, K, V, dO [S][16]; row max / sum / D [S][4]; n_valid [S].
// ---------------------------------------------------------------------------------------------------------------
template <int MAXT>
__global__ void __launch_bounds__(MAXT) k_ibr_ray_bwd(const float* __restrict__ B, const float* __restrict__ pos_enc,
                                                      const float* __restrict__ smp, const float* __restrict__ d_raw, int S,
                                                      float* __restrict__ d_smp) {
    HIP_DYNAMIC_SHARED(float, sm)
    float* Qs = sm;
    float* Ks = Qs + (size_t)S * 16;
    float* Vs = Ks + (size_t)S * 16;
    float* Gs = Vs + (size_t)S * 16;     // d_o
    float* Ms = Gs + (size_t)S * 16;
    float* Ls = Ms + (size_t)S * 4;
    float* Ds = Ls + (size_t)S * 4;
    float* Ns = Ds + (size_t)S * 4;
    const int64_t ray = blockIdx.x;
    const int s = threadIdx.x;
    const bool active = s < S;
    const float* rec = smp + (ray * S + (active ? s : 0)) * NF_SMP_STRIDE;
    const float nval = rec[68];
    float g1[64], g[16], gpe[16], q[16];
    if (active) {
#pragma unroll
        for (int n = 0; n < 64; ++n) g1[n] = B[nf_lin_b(NF_L_GEO0) + n];
        for (int k = 0; k < 65; ++k) {
            float xk = rec[k];
            const float* wr = B + nf_lin_wt(NF_L_GEO0) + k * 64;
#pragma unroll
            for (int n = 0; n < 64; ++n) g1[n] = fmaf(wr[n], xk, g1[n]);
        }
#pragma unroll
        for (int n = 0; n < 16; ++n) g[n] = B[nf_lin_b(NF_L_GEO1) + n];
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            g1[k] = mf_elu(g1[k]);
#pragma unroll
            for (int n = 0; n < 16; ++n) g[n] = fmaf(B[nf_lin_wt(NF_L_GEO1) + k * 16 + n], g1[k], g[n]);
        }
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            g[n] = mf_elu(g[n]);
            gpe[n] = g[n] + pos_enc[(size_t)s * 16 + n];
        }
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
            Qs[s * 16 + n] = q[n];
            Ks[s * 16 + n] = kk[n];
            Vs[s * 16 + n] = vv[n];
        }
        Ns[s] = nval;
    }
    __syncthreads();
    const bool row_on = nval > 1.f;
    float d_pre[16];
    if (active) {
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
            Ms[s * 4 + hd] = mx;
            Ls[s * 4 + hd] = 1.f / l;
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
        float xhat[16], og1[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) og1[n] = B[nf_lin_b(NF_L_OG0) + n];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            xhat[k] = (pre[k] - mu) * rstd;
            float gat = xhat[k] * B[NF_LN_W + k] + B[NF_LN_B + k];
#pragma unroll
            for (int n = 0; n < 16; ++n) og1[n] = fmaf(B[nf_lin_wt(NF_L_OG0) + k * 16 + n], gat, og1[n]);
        }
        float sp = B[nf_lin_b(NF_L_OG1)];
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            og1[n] = mf_elu(og1[n]);
            sp = fmaf(B[nf_lin_wt(NF_L_OG1) + n], og1[n], sp);
        }
        // ---- density head + LayerNorm backward
        float d_sp = (nval >= 1.f && sp > 0.f) ? d_raw[(ray * S + s) * 4 + 3] : 0.f;
        float d_xh[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) d_xh[k] = 0.f;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            float d_og = B[nf_lin_w(NF_L_OG1) + n] * d_sp * mf_elu_grad(og1[n]);
#pragma unroll
            for (int k = 0; k < 16; ++k) d_xh[k] = fmaf(B[nf_lin_w(NF_L_OG0) + n * 16 + k], d_og, d_xh[k]);
        }
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            d_xh[k] *= B[NF_LN_W + k];
            m1 += d_xh[k];
            m2 += d_xh[k] * xhat[k];
        }
        m1 = m1 / 16.f;
        m2 = m2 / 16.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) d_pre[k] = rstd * (d_xh[k] - m1 - xhat[k] * m2);
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
                Gs[s * 16 + hd * 4 + d] = d_o[hd * 4 + d];
                D = fmaf(d_o[hd * 4 + d], o[hd * 4 + d], D);
            }
            Ds[s * 4 + hd] = D;
        }
    }
    __syncthreads();
    if (!active) return;
    // ---- attention backward: dQ (this thread as query), dK / dV (this thread as key)
    float d_gpe[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) d_gpe[j] = d_pre[j];      // residual
    float dq[16], dk[16], dv[16];
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (row_on) {
            float q0 = q[hd * 4] / 2.f, q1 = q[hd * 4 + 1] / 2.f, q2 = q[hd * 4 + 2] / 2.f, q3 = q[hd * 4 + 3] / 2.f;
            const float* gp = Gs + s * 16 + hd * 4;
            float g0 = gp[0], g1v = gp[1], g2 = gp[2], g3 = gp[3];
            float mx = Ms[s * 4 + hd], rl = Ls[s * 4 + hd], D = Ds[s * 4 + hd];     // Ls holds 1 / row sum
            for (int k = 0; k < S; ++k) {
                const float* kp = Ks + k * 16 + hd * 4;
                const float* vp = Vs + k * 16 + hd * 4;
                float sc = fmaf(q3, kp[3], fmaf(q2, kp[2], fmaf(q1, kp[1], q0 * kp[0])));
                float p = mf_exp(sc - mx) * rl;
                float dA = fmaf(g3, vp[3], fmaf(g2, vp[2], fmaf(g1v, vp[1], g0 * vp[0])));
                float dS = p * (dA - D);
                a0 = fmaf(dS, kp[0], a0); a1 = fmaf(dS, kp[1], a1); a2 = fmaf(dS, kp[2], a2); a3 = fmaf(dS, kp[3], a3);
            }
        }
        dq[hd * 4] = a0 / 2.f; dq[hd * 4 + 1] = a1 / 2.f; dq[hd * 4 + 2] = a2 / 2.f; dq[hd * 4 + 3] = a3 / 2.f;
        const float* kp = Ks + s * 16 + hd * 4;
        const float* vp = Vs + s * 16 + hd * 4;
        float k0 = kp[0], k1 = kp[1], k2 = kp[2], k3 = kp[3], v0 = vp[0], v1 = vp[1], v2 = vp[2], v3 = vp[3];
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
        for (int qi = 0; qi < S; ++qi) {
            const float* qp = Qs + qi * 16 + hd * 4;
            const float* gp = Gs + qi * 16 + hd * 4;
            bool on = Ns[qi] > 1.f;
            float q0 = qp[0] / 2.f, q1 = qp[1] / 2.f, q2 = qp[2] / 2.f, q3 = qp[3] / 2.f;
            float sc = on ? fmaf(q3, k3, fmaf(q2, k2, fmaf(q1, k1, q0 * k0))) : -1e9f;
            float p = mf_exp(sc - Ms[qi * 4 + hd]) * Ls[qi * 4 + hd];     // Ls holds 1 / row sum here
            c0 = fmaf(p, gp[0], c0); c1 = fmaf(p, gp[1], c1); c2 = fmaf(p, gp[2], c2); c3 = fmaf(p, gp[3], c3);
            if (on) {
                float dA = fmaf(gp[3], v3, fmaf(gp[2], v2, fmaf(gp[1], v1, gp[0] * v0)));
                float dS = p * (dA - Ds[qi * 4 + hd]);
                b0 = fmaf(dS, q0, b0); b1 = fmaf(dS, q1, b1); b2 = fmaf(dS, q2, b2); b3 = fmaf(dS, q3, b3);
            }
        }
        dk[hd * 4] = b0; dk[hd * 4 + 1] = b1; dk[hd * 4 + 2] = b2; dk[hd * 4 + 3] = b3;
        dv[hd * 4] = c0; dv[hd * 4 + 1] = c1; dv[hd * 4 + 2] = c2; dv[hd * 4 + 3] = c3;
    }
#pragma unroll
    for (int n = 0; n < 16; ++n) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            d_gpe[j] = fmaf(B[nf_att_w(0) + n * 16 + j], dq[n], d_gpe[j]);
            d_gpe[j] = fmaf(B[nf_att_w(1) + n * 16 + j], dk[n], d_gpe[j]);
            d_gpe[j] = fmaf(B[nf_att_w(2) + n * 16 + j], dv[n], d_gpe[j]);
        }
    }
    // ---- geometry_fc backward (16 <- 64 <- 65)
    float d_g1[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) d_g1[k] = 0.f;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        float dgn = d_gpe[n] * mf_elu_grad(g[n]);
#pragma unroll
        for (int k = 0; k < 64; ++k) d_g1[k] = fmaf(B[nf_lin_w(NF_L_GEO1) + n * 64 + k], dgn, d_g1[k]);
    }
    float* out = d_smp + (ray * S + s) * NF_SMP_STRIDE;
    for (int k0 = 0; k0 < 65; k0 += 13) {       // 5 chunks of 13 outputs keep the accumulators in registers
        float acc[13];
#pragma unroll
        for (int j = 0; j < 13; ++j) acc[j] = 0.f;
#pragma unroll
        for (int n = 0; n < 64; ++n) {
            float dn = d_g1[n] * mf_elu_grad(g1[n]);
            const float* wr = B + nf_lin_w(NF_L_GEO0) + n * 65 + k0;
#pragma unroll
            for (int j = 0; j < 13; ++j) acc[j] = fmaf(wr[j], dn, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 13; ++j) out[k0 + j] = acc[j];
    }
    const float* gr = d_raw + (ray * S + s) * 4;
    out[65] = gr[0];
    out[66] = gr[1];
    out[67] = gr[2];
}

// ---------------------------------------------------------------------------------------------------------------
// kernel B on the matrix cores (S in {32, 64, 128}): a wave owns 32 consecutive samples of one ray (sample = MFMA row),
// geometry_fc / q,k,v / fc / out_geometry_fc are register-chained MFMA GEMMs like the row kernels, the ray attention runs
// on the VALU with K/V of the ray in LDS -- each lane half holds two complete heads of its sample (fragment registers
// 0-3 = head h, 4-7 = head 2 + h), so the halves split the four heads.  Workgroups are persistent (weights staged once).
// ---------------------------------------------------------------------------------------------------------------
#define RYL(off) (lds[RY_OFF + (off)])      // RY_OFF: where the ray section sits inside this kernel's LDS image

struct RayActs {
    f32x16 MEAN2, VAR2, G1a, G1b;
    float g[8], gpe[8], q[8], k[8], v[8], o[8], xhat[8], og1[8];
    float wmean, nval, rstd, sig_pre, mx[2], l[2];
};

__device__ __forceinline__ f32x16 ray_bias(const float* rs, int tile, int h) {
    f32x16 a;
    const float* b = rs + RYS_BIAS + tile * 32 + h * 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = b[r];
    return a;
}

template <int NSTEPS>
__device__ __forceinline__ f32x16 ray_gemm(const float* rs, int rec, int lane, const f32x16& x, f32x16 acc) {
#pragma unroll
    for (int r = 0; r < NSTEPS; ++r) acc = NF_MFMA(rs[(rec + r) * 64 + lane], x[r], acc);
    return acc;
}

template <int NSTEPS>
__device__ __forceinline__ f32x16 ray_gemm8(const float* rs, int rec, int lane, const float (&x)[8], f32x16 acc) {
#pragma unroll
    for (int r = 0; r < NSTEPS; ++r) acc = NF_MFMA(rs[(rec + r) * 64 + lane], x[r], acc);
    return acc;
}

// per-sample forward up to q/k/v (rs = LDS pointer to the ray section)
__device__ __forceinline__ void ray_forward_a(const float* rs, int lane, int h, const float* __restrict__ rec,
                                              const float* __restrict__ pe, RayActs& a) {
    a.MEAN2 = frag_load32(rec, h);
    a.VAR2 = frag_load32(rec + 32, h);
    a.wmean = rec[64];          // (the colours at 65 .. 67 go straight to the output, see the kernel)
    a.nval = rec[68];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        int rc = RYR_GEO0 + nt * 33;
        f32x16 acc = ray_bias(rs, nt, h);
        acc = ray_gemm<16>(rs, rc, lane, a.MEAN2, acc);
        acc = ray_gemm<16>(rs, rc + 16, lane, a.VAR2, acc);
        acc = NF_MFMA(rs[(rc + 32) * 64 + lane], h ? 0.f : a.wmean, acc);
        if (nt == 0) a.G1a = elu16(acc); else a.G1b = elu16(acc);
    }
    {
        f32x16 acc = ray_bias(rs, 2, h);
        acc = ray_gemm<16>(rs, RYR_GEO1, lane, a.G1a, acc);
        acc = ray_gemm<16>(rs, RYR_GEO1 + 16, lane, a.G1b, acc);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            a.g[r] = mf_elu(acc[r]);
            a.gpe[r] = a.g[r] + pe[nf_nidx(r, h)];
        }
    }
    {
        f32x16 z = zero16();
        f32x16 qk = ray_gemm8<8>(rs, RYR_QKV, lane, a.gpe, z);
        f32x16 vv = ray_gemm8<8>(rs, RYR_QKV + 8, lane, a.gpe, z);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            a.q[r] = qk[r];
            a.k[r] = qk[8 + r];
            a.v[r] = vv[r];
        }
    }
}

// attention of this lane's two heads over the S keys in LDS (Ks / Vs: [S][16]); masked query rows (n_valid <= 1) use
// q = 0: all scores equal => uniform attention, exactly what masked_fill(-1e9) + softmax gives (mlp_network.py:36).
// ONE pass over the keys: scores of a block of 8 keys, running maximum, the sums rescaled when the maximum moves (a block, not a
// key, at a time: the rescale is one exponential per block and mostly exp(0)) -- every K row is read and multiplied once instead
// of twice.  mx / l end as the row maximum and the softmax denominator relative to it, which is what the backward re-uses.
__device__ __forceinline__ void ray_attention(const float* Ks, const float* Vs, int S, int h, RayActs& a) {
    const bool row_on = a.nval > 1.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int head = h + 2 * j;
        float q0 = row_on ? a.q[4 * j + 0] * 0.5f : 0.f, q1 = row_on ? a.q[4 * j + 1] * 0.5f : 0.f,
              q2 = row_on ? a.q[4 * j + 2] * 0.5f : 0.f, q3 = row_on ? a.q[4 * j + 3] * 0.5f : 0.f;
        float mx = -3.0e38f, l = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int k0 = 0; k0 < S; k0 += 8) {          // S is a multiple of 32
            float sc[8], bm = -3.0e38f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // one 16-byte LDS read per key and head (rows are 64 bytes, a head's slice 16: aligned) -- as four scalars the
                // compiler emits ds_read2_b32 pairs, whose 1 KB offset range costs an address add per read
                const float4 kv = *reinterpret_cast<const float4*>(Ks + (k0 + i) * 16 + head * 4);
                sc[i] = fmaf(q3, kv.w, fmaf(q2, kv.z, fmaf(q1, kv.y, q0 * kv.x)));
                bm = fmaxf(bm, sc[i]);
            }
            if (bm > mx) {                            // the maximum moves: bring the sums to the new reference
                const float r = mf_exp(mx - bm);      // first block: exp(-3e38) = 0 on zero sums
                l *= r; a0 *= r; a1 *= r; a2 *= r; a3 *= r;
                mx = bm;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 vv = *reinterpret_cast<const float4*>(Vs + (k0 + i) * 16 + head * 4);
                const float p = mf_exp(sc[i] - mx);
                l += p;
                a0 = fmaf(p, vv.x, a0); a1 = fmaf(p, vv.y, a1); a2 = fmaf(p, vv.z, a2); a3 = fmaf(p, vv.w, a3);
            }
        }
        float rl = 1.f / l;
        a.o[4 * j + 0] = a0 * rl; a.o[4 * j + 1] = a1 * rl; a.o[4 * j + 2] = a2 * rl; a.o[4 * j + 3] = a3 * rl;
        a.mx[j] = mx;
        a.l[j] = l;
    }
}

// The same attention with the lane's TWO heads (h and h + 2) carried side by side in packed registers: every multiply-add of a key is one
// v_pk_fma_f32 for both heads, the exponent arguments one v_pk_add / v_pk_mul -- 15 vector instructions per key where the scalar form
// above issues ~28 (vector instructions are paid in matrix-pipe time).  Kp / Vp: [S][16] with the two heads of a lane half INTERLEAVED,
// position 8 h + 2 d + j for head h + 2 j, dimension d (ray_kv_slot), so that one 16-byte read delivers (d0: both heads, d1: both heads).
// Element for element the arithmetic of ray_attention (same fma chains, same block-wise running maximum, v_exp_f32 of the same
// argument), except that a head whose maximum did not move is rescaled by exp(0) = 1 instead of being skipped.
__host__ __device__ constexpr int ray_kv_slot(int r, int h) { return 8 * h + 2 * (r & 3) + (r >> 2); }       // fragment register r of lane half h
__device__ __forceinline__ void ray_attention_pk(const float* Kp, const float* Vp, int S, int h, RayActs& a) {
    const bool row_on = a.nval > 1.f;
    const f32x2 log2e = {1.44269502f, 1.44269502f};          // 0x3fb8aa3b: the constant __expf multiplies by
    f32x2 q[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) q[d] = row_on ? f32x2{a.q[d] * 0.5f, a.q[4 + d] * 0.5f} : f32x2{0.f, 0.f};
    f32x2 mx = {-3.0e38f, -3.0e38f}, l = {0.f, 0.f}, acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    for (int k0 = 0; k0 < S; k0 += 8) {          // S is a multiple of 32
        f32x2 sc[8], bm = mx;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 ka = *reinterpret_cast<const float4*>(Kp + (k0 + i) * 16 + 8 * h);
            const float4 kb = *reinterpret_cast<const float4*>(Kp + (k0 + i) * 16 + 8 * h + 4);
            f32x2 t = q[0] * f32x2{ka.x, ka.y};
            t = __builtin_elementwise_fma(q[1], f32x2{ka.z, ka.w}, t);
            t = __builtin_elementwise_fma(q[2], f32x2{kb.x, kb.y}, t);
            t = __builtin_elementwise_fma(q[3], f32x2{kb.z, kb.w}, t);
            sc[i] = t;
            bm = __builtin_elementwise_max(bm, t);
        }
        // bring the sums to the block's reference (exp(0) = 1 for a head whose maximum stays; first block: exp(-3e38) = 0 on zero sums)
        const f32x2 rsc = nf_exp2x2((mx - bm) * log2e);
        l = l * rsc;
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[d] = acc[d] * rsc;
        mx = bm;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 va = *reinterpret_cast<const float4*>(Vp + (k0 + i) * 16 + 8 * h);
            const float4 vb = *reinterpret_cast<const float4*>(Vp + (k0 + i) * 16 + 8 * h + 4);
            const f32x2 p = nf_exp2x2((sc[i] - mx) * log2e);
            l = l + p;
            acc[0] = __builtin_elementwise_fma(p, f32x2{va.x, va.y}, acc[0]);
            acc[1] = __builtin_elementwise_fma(p, f32x2{va.z, va.w}, acc[1]);
            acc[2] = __builtin_elementwise_fma(p, f32x2{vb.x, vb.y}, acc[2]);
            acc[3] = __builtin_elementwise_fma(p, f32x2{vb.z, vb.w}, acc[3]);
        }
    }
    const f32x2 rl = {1.f / l.x, 1.f / l.y};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const f32x2 o = acc[d] * rl;
        a.o[d] = o.x;
        a.o[4 + d] = o.y;
    }
    a.mx[0] = mx.x; a.mx[1] = mx.y;
    a.l[0] = l.x; a.l[1] = l.y;
}

// fc + residual, LayerNorm (eps 1e-6), out_geometry_fc -> sigma
__device__ __forceinline__ float ray_forward_b(const float* rs, int lane, int h, RayActs& a) {
    f32x16 pre = ray_gemm8<8>(rs, RYR_FC, lane, a.o, zero16());
    float s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        pre[r] += a.gpe[r];
        s1 += pre[r];
    }
    float mu = half_sum(s1) / 16.f;
    float s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s2 += (pre[r] - mu) * (pre[r] - mu);
    a.rstd = 1.f / sqrtf(half_sum(s2) / 16.f + 1e-6f);
    float gat[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        a.xhat[r] = (pre[r] - mu) * a.rstd;
        gat[r] = a.xhat[r] * rs[RYS_LNW + h * 8 + r] + rs[RYS_LNB + h * 8 + r];
    }
    f32x16 og = ray_gemm8<8>(rs, RYR_OG0, lane, gat, ray_bias(rs, 3, h));
    float sp = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        a.og1[r] = mf_elu(og[r]);
        sp = fmaf(rs[RYS_OG1 + h * 8 + r], a.og1[r], sp);
    }
    a.sig_pre = half_sum(sp) + rs[RYS_OG1 + 16];
    float sigma = fmaxf(a.sig_pre, 0.f);
    return a.nval < 1.f ? 0.f : sigma;
}

template <int WPR>      // waves (32-sample tiles) per ray: S = 32 * WPR; workgroup = max(4, WPR) waves
__global__ void __launch_bounds__(WPR > 4 ? 64 * WPR : 256, 2) k_ibr_ray_fwd_mfma(const float* __restrict__ wblob, const float* __restrict__ pos_enc,
                                                             const float* __restrict__ smp, int64_t n_rays,
                                                             float* __restrict__ raw) {
    HIP_DYNAMIC_SHARED(float, lds)
    constexpr int S = 32 * WPR, NW = WPR > 4 ? WPR : 4, RPI = NW / WPR;      // rays per workgroup iteration
    float* rs = lds;                                      // [0, RYF_FLOATS): forward part of the ray section
    float* kv = lds + RYF_FLOATS;                         // NW waves x 32 samples x (16 K + 16 V)
    for (int i = threadIdx.x; i < RYF_FLOATS; i += blockDim.x) rs[i] = wblob[RY_BASE + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int64_t n_iter = (n_rays + RPI - 1) / RPI;
    for (int64_t it = blockIdx.x; it < n_iter; it += gridDim.x) {
        asm volatile("" ::: "memory");
        int64_t ray = it * RPI + wave / WPR;
        const bool live = ray < n_rays;
        if (!live) ray = n_rays - 1;
        const int s = (wave % WPR) * 32 + m;
        const float* rec = smp + (ray * S + s) * NF_SMP_STRIDE;
        RayActs a;
        ray_forward_a(rs, lane, h, rec, pos_enc + (size_t)s * 16, a);
        float* Ks = kv + (wave / WPR) * (S * 32);
        float* Vs = Ks + S * 16;
#pragma unroll
        for (int r = 0; r < 8; ++r) {              // the two heads of a lane half interleaved: ray_attention_pk
            Ks[s * 16 + ray_kv_slot(r, h)] = a.k[r];
            Vs[s * 16 + ray_kv_slot(r, h)] = a.v[r];
        }
        __syncthreads();
        ray_attention_pk(Ks, Vs, S, h, a);
        float sigma = ray_forward_b(rs, lane, h, a);
        if (live && h == 0) {
            float* ro = raw + (ray * S + s) * 4;
            const float4 tail = *reinterpret_cast<const float4*>(rec + 64);          // wmean | rgb
            *reinterpret_cast<float4*>(ro) = make_float4(tail.y, tail.z, tail.w, sigma);
        }
        __syncthreads();      // K/V of this iteration are dead before the next one overwrites them
    }
}

// ---- backward ---------------------------------------------------------------------------------------------------
template <int NSTEPS>
__device__ __forceinline__ f32x16 ray_gemm_T(const float* rs, int rec, int lane, const f32x16& dy, f32x16 acc) {
#pragma unroll
    for (int r = 0; r < NSTEPS; ++r) acc = NF_MFMA(rs[RYT_BASE + (rec + r) * 64 + lane], dy[r], acc);
    return acc;
}
template <int NSTEPS>
__device__ __forceinline__ f32x16 ray_gemm8_T(const float* rs, int rec, int lane, const float (&dy)[8], f32x16 acc) {
#pragma unroll
    for (int r = 0; r < NSTEPS; ++r) acc = NF_MFMA(rs[RYT_BASE + (rec + r) * 64 + lane], dy[r], acc);
    return acc;
}

// LDS per ray in the backward: K, V, Q, dO [S][16]; {M, 1 / L, D, -} [S][4 heads] (one 16-byte read per query and head)
#define RAY_BWD_LDS_PER_SAMPLE (4 * 16 + 4 * 4)

// NWV: waves per workgroup (4 or 8; at least WPR).  The weight image takes 65 KB of LDS, so one workgroup per CU: with 8 waves
// (two per SIMD, 247 registers each) the VALU attention of one wave runs under the LDS / matrix work of the other -- chosen by the
// host whenever the rays still fill the chip at 8 / WPR rays per workgroup.
template <int WPR, int NWV>
__global__ void __launch_bounds__(64 * (WPR > NWV ? WPR : NWV), 1) k_ibr_ray_bwd_mfma(const float* __restrict__ wblob, const float* __restrict__ pos_enc,
                                                             const float* __restrict__ smp, const float* __restrict__ d_raw,
                                                             int64_t n_rays, float* __restrict__ d_smp) {
    HIP_DYNAMIC_SHARED(float, lds)
    constexpr int S = 32 * WPR, NW = WPR > NWV ? WPR : NWV, RPI = NW / WPR;
    float* rs = lds;
    float* ray_lds = lds + RY_FLOATS;
    for (int i = threadIdx.x; i < RY_FLOATS; i += blockDim.x) rs[i] = wblob[RY_BASE + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int64_t n_iter = (n_rays + RPI - 1) / RPI;
    for (int64_t it = blockIdx.x; it < n_iter; it += gridDim.x) {
        asm volatile("" ::: "memory");
        int64_t ray = it * RPI + wave / WPR;
        const bool live = ray < n_rays;
        if (!live) ray = n_rays - 1;
        const int s = (wave % WPR) * 32 + m;
        const float* rec = smp + (ray * S + s) * NF_SMP_STRIDE;
        RayActs a;
        ray_forward_a(rs, lane, h, rec, pos_enc + (size_t)s * 16, a);
        float* Ks = ray_lds + (wave / WPR) * (S * RAY_BWD_LDS_PER_SAMPLE);
        float* Vs = Ks + S * 16;
        float* Qs = Vs + S * 16;
        float* Gs = Qs + S * 16;          // d_o
        float* MLD = Gs + S * 16;         // {row max, 1 / row sum, D, -} per (sample, head)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            Ks[s * 16 + nf_nidx(r, h)] = a.k[r];
            Vs[s * 16 + nf_nidx(r, h)] = a.v[r];
            Qs[s * 16 + nf_nidx(r, h)] = a.nval > 1.f ? a.q[r] * 0.5f : 0.f;      // scaled; masked query rows score 0
        }
        __syncthreads();
        ray_attention(Ks, Vs, S, h, a);
        (void)ray_forward_b(rs, lane, h, a);
        // ---- density head + LayerNorm backward (oracle/ibrnet_manual_bwd.py)
        const float* graw = d_raw + (ray * S + s) * 4;
        float d_sp = (a.nval >= 1.f && a.sig_pre > 0.f) ? graw[3] : 0.f;
        float d_og[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) d_og[r] = rs[RYS_OG1 + h * 8 + r] * d_sp * mf_elu_grad(a.og1[r]);
        f32x16 d_gat = ray_gemm8_T<8>(rs, RYT_OG0, lane, d_og, zero16());
        float d_xh[8], m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            d_xh[r] = d_gat[r] * rs[RYS_LNW + h * 8 + r];
            m1 += d_xh[r];
            m2 += d_xh[r] * a.xhat[r];
        }
        m1 = half_sum(m1) / 16.f;
        m2 = half_sum(m2) / 16.f;
        float d_pre[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) d_pre[r] = a.rstd * (d_xh[r] - m1 - a.xhat[r] * m2);
        f32x16 d_o = ray_gemm8_T<8>(rs, RYT_FC, lane, d_pre, zero16());
        const bool row_on = a.nval > 1.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int head = h + 2 * j;
            float D = 0.f;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                Gs[s * 16 + head * 4 + d] = d_o[4 * j + d];
                D = fmaf(d_o[4 * j + d], a.o[4 * j + d], D);
            }
            *reinterpret_cast<float4*>(MLD + (s * 4 + head) * 4) = make_float4(row_on ? a.mx[j] : 0.f, 1.f / a.l[j], D, 0.f);
        }
        __syncthreads();
        // ---- attention backward: dQ (this sample as query), dK / dV (this sample as key); masked query rows have zero
        //      scores (q = 0), contribute to dV with uniform weights and nothing to dQ / dK
        float dq[8], dk[8], dv[8];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int head = h + 2 * j;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            if (row_on) {
                float q0 = a.q[4 * j] * 0.5f, q1 = a.q[4 * j + 1] * 0.5f, q2 = a.q[4 * j + 2] * 0.5f, q3 = a.q[4 * j + 3] * 0.5f;
                float g0 = d_o[4 * j], g1 = d_o[4 * j + 1], g2 = d_o[4 * j + 2], g3 = d_o[4 * j + 3];
                const float4 own = *reinterpret_cast<const float4*>(MLD + (s * 4 + head) * 4);
                float mx = a.mx[j], rl = own.y, D = own.z;
                for (int k = 0; k < S; ++k) {           // 16-byte LDS reads (see ray_attention)
                    const float4 kv = *reinterpret_cast<const float4*>(Ks + k * 16 + head * 4);
                    const float4 vv = *reinterpret_cast<const float4*>(Vs + k * 16 + head * 4);
                    float p = mf_exp(fmaf(q3, kv.w, fmaf(q2, kv.z, fmaf(q1, kv.y, q0 * kv.x))) - mx) * rl;
                    float dA = fmaf(g3, vv.w, fmaf(g2, vv.z, fmaf(g1, vv.y, g0 * vv.x)));
                    float dS = p * (dA - D);
                    a0 = fmaf(dS, kv.x, a0); a1 = fmaf(dS, kv.y, a1); a2 = fmaf(dS, kv.z, a2); a3 = fmaf(dS, kv.w, a3);
                }
            }
            dq[4 * j] = a0 * 0.5f; dq[4 * j + 1] = a1 * 0.5f; dq[4 * j + 2] = a2 * 0.5f; dq[4 * j + 3] = a3 * 0.5f;
            float k0 = a.k[4 * j], k1 = a.k[4 * j + 1], k2 = a.k[4 * j + 2], k3 = a.k[4 * j + 3];
            float v0 = a.v[4 * j], v1 = a.v[4 * j + 1], v2 = a.v[4 * j + 2], v3 = a.v[4 * j + 3];
            float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
            for (int qi = 0; qi < S; ++qi) {
                const float4 qv = *reinterpret_cast<const float4*>(Qs + qi * 16 + head * 4);
                const float4 gv = *reinterpret_cast<const float4*>(Gs + qi * 16 + head * 4);
                float q0 = qv.x, q1 = qv.y, q2 = qv.z, q3 = qv.w;              // pre-scaled; zero for masked query rows
                const float4 mld = *reinterpret_cast<const float4*>(MLD + (qi * 4 + head) * 4);
                float p = mf_exp(fmaf(q3, k3, fmaf(q2, k2, fmaf(q1, k1, q0 * k0))) - mld.x) * mld.y;
                c0 = fmaf(p, gv.x, c0); c1 = fmaf(p, gv.y, c1); c2 = fmaf(p, gv.z, c2); c3 = fmaf(p, gv.w, c3);
                float dA = fmaf(gv.w, v3, fmaf(gv.z, v2, fmaf(gv.y, v1, gv.x * v0)));
                float dS = p * (dA - mld.z);
                b0 = fmaf(dS, q0, b0); b1 = fmaf(dS, q1, b1); b2 = fmaf(dS, q2, b2); b3 = fmaf(dS, q3, b3);
            }
            dk[4 * j] = b0; dk[4 * j + 1] = b1; dk[4 * j + 2] = b2; dk[4 * j + 3] = b3;
            dv[4 * j] = c0; dv[4 * j + 1] = c1; dv[4 * j + 2] = c2; dv[4 * j + 3] = c3;
        }
        // ---- d gpe = d pre + Wq^T dq + Wk^T dk + Wv^T dv ; geometry_fc backward
        f32x16 dqk;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            dqk[r] = dq[r];
            dqk[8 + r] = dk[r];
        }
        f32x16 d_gpe = ray_gemm_T<16>(rs, RYT_QKV, lane, dqk, zero16());
        d_gpe = ray_gemm8_T<8>(rs, RYT_QKV + 16, lane, dv, d_gpe);
        float d_g[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) d_g[r] = (d_gpe[r] + d_pre[r]) * mf_elu_grad(a.g[r]);
        f32x16 d_g1a = ray_gemm8_T<8>(rs, RYT_GEO1, lane, d_g, zero16());
        f32x16 d_g1b = ray_gemm8_T<8>(rs, RYT_GEO1 + 8, lane, d_g, zero16());
        float dw = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            d_g1a[r] *= mf_elu_grad(a.G1a[r]);
            d_g1b[r] *= mf_elu_grad(a.G1b[r]);
            dw = fmaf(rs[RYS_GEO0W + 0 * 32 + h * 16 + r], d_g1a[r], dw);
            dw = fmaf(rs[RYS_GEO0W + 1 * 32 + h * 16 + r], d_g1b[r], dw);
        }
        dw = half_sum(dw);
        f32x16 d_mean2 = ray_gemm_T<16>(rs, RYT_GEO0 + 16, lane, d_g1b, ray_gemm_T<16>(rs, RYT_GEO0, lane, d_g1a, zero16()));
        f32x16 d_var2 = ray_gemm_T<16>(rs, RYT_GEO0 + 48, lane, d_g1b, ray_gemm_T<16>(rs, RYT_GEO0 + 32, lane, d_g1a, zero16()));
        if (live) {
            float* out = d_smp + (ray * S + s) * NF_SMP_STRIDE;
            frag_store32(out, h, d_mean2);
            frag_store32(out + 32, h, d_var2);
            if (h == 0) {
                out[64] = dw;
                out[65] = graw[0];
                out[66] = graw[1];
                out[67] = graw[2];
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
extern "C" int nf_ibrnet_mfma_supported(int n_samples, int n_views) {
    return (n_views >= 1 && n_views <= 32 && n_samples >= 1 && n_samples <= NF_IBR_MAX_S) ? 1 : 0;
}

// lanes per sample: the next power of two of the view count
static int rows_lanes_per_sample(int n_views) {
    int vp = 1;
    while (vp < n_views) vp <<= 1;
    return vp;
}

extern "C" int64_t nf_ibrnet_mfma_workspace_floats(int64_t n_rays, int n_samples) {
    return n_rays * n_samples * NF_SMP_STRIDE;
}

template <int V>
static void launch_rows_fwd(const float* wblob, const float* bf_blob, const float* rgb_feat, const float* ray_diff, const float* mask,
                            int64_t n_samples, int nv, int aa, float* smp, const RowGather& g, hipStream_t st) {
    int64_t tiles = (n_samples + 32 / V - 1) / (32 / V);
    int64_t blocks = (tiles + NF_ROWS_FWD_WAVES - 1) / NF_ROWS_FWD_WAVES;
    if (blocks > 1024) blocks = 1024;     // persistent-ish: 2 workgroups per CU hold the 58 KB weight image each
#define NF_ROWS_FWD_GO(BFV, GV, blobp, floats)                                                                                          \
    hipLaunchKernelGGL((k_ibr_rows_fwd<V, BFV, GV>), dim3((unsigned)blocks), dim3(64 * NF_ROWS_FWD_WAVES), (floats) * sizeof(float), st, \
                       blobp, rgb_feat, ray_diff, mask, n_samples, nv, aa, smp, g)
    if (g.featmap) {
        if (bf_blob) NF_ROWS_FWD_GO(true, true, bf_blob, NF_BF_FWD_FLOATS);
        else NF_ROWS_FWD_GO(false, true, wblob, NF_MFMA_FWD_FLOATS);
    } else {
        if (bf_blob) NF_ROWS_FWD_GO(true, false, bf_blob, NF_BF_FWD_FLOATS);
        else NF_ROWS_FWD_GO(false, false, wblob, NF_MFMA_FWD_FLOATS);
    }
#undef NF_ROWS_FWD_GO
}

// ---- which form of kernel A runs (forward): the sample-on-the-lane form where it is built (fp32-grade operands, 2 <= V <=
// NF_SOL_MAX_V), the row form otherwise.  nf_ibrnet_rows_form: 0 = default (sample-on-the-lane, bf16x3 operands), 1 = the row form
// always, 2 = sample-on-the-lane with fp32 matrix-core operands (V <= 4; above that the row form) -- TEST / DIAGNOSTIC hook: the
// tests run the forms on the same inputs.
static int g_rows_form = 0;
extern "C" int nf_ibrnet_rows_form(int form) {
    const int old = g_rows_form;
    if (form >= 0 && form <= 2) g_rows_form = form;
    return old;
}
static bool sol_selected(int n_views, bool bf16) {
    if (bf16 || n_views < 2 || n_views > NF_SOL_MAX_V) return false;
    return g_rows_form == 0 || (g_rows_form == 2 && n_views <= 4);
}
/* does the sample-on-the-lane form run for this shape under the current setting? */
extern "C" int nf_ibrnet_sol_selected(int n_views, int bf16_operands) { return sol_selected(n_views, bf16_operands != 0) ? 1 : 0; }

template <int V, int OP, int OCC>
static int launch_sol_fwd_vo(const float* wblob, const float* rgb_feat, const float* ray_diff, const float* mask, int64_t n_samples, int aa,
                             float* smp, const RowGather& g, hipStream_t st) {
    constexpr int NWV = sol_fwd_waves(OCC);
    const int64_t tiles = (n_samples + 31) / 32;
    int64_t blocks = (tiles + NWV - 1) / NWV;
    const int64_t cap = 256;                // one resident workgroup per CU (MI355X: 256) stages the weight image once and walks its tiles
    if (blocks > cap) blocks = cap;
    const size_t smem = (size_t)(OP == 3 ? (int)X3_FLOATS : (int)NF_MFMA_FWD_FLOATS) * sizeof(float);
    static bool configured_on[NF_MAX_DEVICES][2] = {};      // > 64 KB of dynamic LDS: opt-in once per kernel and device
    bool& configured = configured_on[nf_current_device()][g.featmap ? 1 : 0];
    if (!configured && smem > 64 * 1024) {
        const void* fn = g.featmap ? (const void*)k_ibr_sol_fwd<V, OP, true, OCC> : (const void*)k_ibr_sol_fwd<V, OP, false, OCC>;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
            nf_set_error("nf_ibrnet_fwd_mfma: cannot reserve %zu bytes of LDS", smem);
            return 1;
        }
        configured = true;
    }
    if (g.featmap)
        hipLaunchKernelGGL((k_ibr_sol_fwd<V, OP, true, OCC>), dim3((unsigned)blocks), dim3(64 * NWV), smem, st, wblob, rgb_feat, ray_diff, mask,
                           n_samples, aa, smp, g);
    else
        hipLaunchKernelGGL((k_ibr_sol_fwd<V, OP, false, OCC>), dim3((unsigned)blocks), dim3(64 * NWV), smem, st, wblob, rgb_feat, ray_diff, mask,
                           n_samples, aa, smp, g);
    return 0;
}
template <int V>
static int launch_sol_fwd_v(const float* wblob, const float* rgb_feat, const float* ray_diff, const float* mask, int64_t n_samples, int aa,
                            float* smp, const RowGather& g, hipStream_t st) {
    // A tile is 32 samples x V views of serial work.  With no more tiles than the chip has SIMDs (1024) every tile gets a SIMD of its own
    // in four-wave workgroups (the attack's coarse level: 512 rays x 64 samples = 1024 tiles); beyond that two waves per SIMD hide each
    // other's LDS / memory latency where the registers allow it (V <= 4: measured 0.361 -> 0.311 ms at 4096 x 64 x 4, fp32 operands).
    const bool two = V <= 4 && (n_samples + 31) / 32 > 1024;
    if constexpr (V <= 4) {
        if (g_rows_form == 2)
            return two ? launch_sol_fwd_vo<V, 0, 2>(wblob, rgb_feat, ray_diff, mask, n_samples, aa, smp, g, st)
                       : launch_sol_fwd_vo<V, 0, 1>(wblob, rgb_feat, ray_diff, mask, n_samples, aa, smp, g, st);
        if (two) return launch_sol_fwd_vo<V, 3, 2>(wblob, rgb_feat, ray_diff, mask, n_samples, aa, smp, g, st);
    }
    return launch_sol_fwd_vo<V, 3, 1>(wblob, rgb_feat, ray_diff, mask, n_samples, aa, smp, g, st);
}
static int launch_sol_fwd(int n_views, const float* wblob, const float* rgb_feat, const float* ray_diff, const float* mask, int64_t n_samples,
                          int aa, float* smp, const RowGather& g, hipStream_t st) {
    switch (n_views) {
#define NF_SOL_CASE(VV) case VV: return launch_sol_fwd_v<VV>(wblob, rgb_feat, ray_diff, mask, n_samples, aa, smp, g, st);
        NF_SOL_CASE(2) NF_SOL_CASE(3) NF_SOL_CASE(4) NF_SOL_CASE(5) NF_SOL_CASE(6) NF_SOL_CASE(7) NF_SOL_CASE(8) NF_SOL_CASE(9) NF_SOL_CASE(10)
#undef NF_SOL_CASE
    }
    nf_set_error("nf_ibrnet_fwd_mfma: no sample-on-the-lane kernel for %d views", n_views);
    return 1;
}

static int ibrnet_fwd_impl(const char* who, const float* bf_blob, const float* mfma_blob, const float* blob, const float* pos_enc,
                           const float* rgb_feat, const float* ray_diff, const float* mask, int64_t n_rays, int n_samples, int n_views,
                           int anti_alias_pooling, float* raw, float* workspace, nf_stream_t stream, const RowGather& gather = RowGather{}) {
    NF_REQUIRE(nf_ibrnet_mfma_supported(n_samples, n_views), "%s: 1 <= V <= 32 (got %d)", who, n_views);
    if (n_rays == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int64_t ns = n_rays * n_samples;
    if (sol_selected(n_views, bf_blob != nullptr)) {
        const int rc = launch_sol_fwd(n_views, mfma_blob, rgb_feat, ray_diff, mask, ns, anti_alias_pooling, workspace, gather, st);
        if (rc) return rc;
    } else
    switch (rows_lanes_per_sample(n_views)) {
        case 1: launch_rows_fwd<1>(mfma_blob, bf_blob, rgb_feat, ray_diff, mask, ns, n_views, anti_alias_pooling, workspace, gather, st); break;
        case 2: launch_rows_fwd<2>(mfma_blob, bf_blob, rgb_feat, ray_diff, mask, ns, n_views, anti_alias_pooling, workspace, gather, st); break;
        case 4: launch_rows_fwd<4>(mfma_blob, bf_blob, rgb_feat, ray_diff, mask, ns, n_views, anti_alias_pooling, workspace, gather, st); break;
        case 8: launch_rows_fwd<8>(mfma_blob, bf_blob, rgb_feat, ray_diff, mask, ns, n_views, anti_alias_pooling, workspace, gather, st); break;
        case 16: launch_rows_fwd<16>(mfma_blob, bf_blob, rgb_feat, ray_diff, mask, ns, n_views, anti_alias_pooling, workspace, gather, st); break;
        default: launch_rows_fwd<32>(mfma_blob, bf_blob, rgb_feat, ray_diff, mask, ns, n_views, anti_alias_pooling, workspace, gather, st); break;
    }
    NF_LAUNCH_CHECK("nf_ibrnet_fwd_mfma (rows)");
    if (n_samples == 32 || n_samples == 64 || n_samples == 128 || n_samples == 256) {       // per-ray part on the matrix cores as well
        const int wpr = n_samples / 32, nw = wpr > 4 ? wpr : 4, rpi = nw / wpr;
        int64_t iters = (n_rays + rpi - 1) / rpi;
        unsigned blocks = (unsigned)(iters < 768 ? iters : 768);        // three workgroups per CU (512 and 1024 measured 8 % slower)
        size_t smem_ray = (size_t)(RYF_FLOATS + nw * 32 * 32) * sizeof(float);
        if (wpr == 1)
            hipLaunchKernelGGL(k_ibr_ray_fwd_mfma<1>, dim3(blocks), dim3(256), smem_ray, st, mfma_blob, pos_enc, workspace, n_rays, raw);
        else if (wpr == 2)
            hipLaunchKernelGGL(k_ibr_ray_fwd_mfma<2>, dim3(blocks), dim3(256), smem_ray, st, mfma_blob, pos_enc, workspace, n_rays, raw);
        else if (wpr == 4)
            hipLaunchKernelGGL(k_ibr_ray_fwd_mfma<4>, dim3(blocks), dim3(256), smem_ray, st, mfma_blob, pos_enc, workspace, n_rays, raw);
        else {
            static bool configured_on[NF_MAX_DEVICES] = {};
            bool& configured = configured_on[nf_current_device()];
            if (!configured) {
                if (hipFuncSetAttribute((const void*)k_ibr_ray_fwd_mfma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_ray) != hipSuccess) {
                    nf_set_error("%s: cannot reserve %zu bytes of LDS", who, smem_ray);
                    return 1;
                }
                configured = true;
            }
            hipLaunchKernelGGL(k_ibr_ray_fwd_mfma<8>, dim3(blocks), dim3(512), smem_ray, st, mfma_blob, pos_enc, workspace, n_rays, raw);
        }
        NF_LAUNCH_CHECK("nf_ibrnet_fwd_mfma (ray, mfma)");
        return 0;
    }
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

extern "C" int nf_ibrnet_fwd_mfma(const float* mfma_blob, const float* blob, const float* pos_enc, const float* rgb_feat,
                                  const float* ray_diff, const float* mask, int64_t n_rays, int n_samples, int n_views,
                                  int anti_alias_pooling, float* raw, float* workspace, nf_stream_t stream) {
    return ibrnet_fwd_impl("nf_ibrnet_fwd_mfma", nullptr, mfma_blob, blob, pos_enc, rgb_feat, ray_diff, mask, n_rays, n_samples, n_views,
                           anti_alias_pooling, raw, workspace, stream);
}

/* nf_ibrnet_fwd_mfma (bf16_blob == nullptr) / nf_ibrnet_fwd_mfma_bf16 with Projector.compute folded into the row kernel: the inputs
 * are the sample points, the camera workspace, the source images [V][H][W][3] and the channels-last feature maps [V][32][Hf][Wf]
 * (fs_c == 1, pixel records 16-byte aligned); rgb_feat / ray_diff are never written, the validity mask [n_rays * n_samples * V] is. */
extern "C" int nf_ibrnet_fwd_mfma_gather(const float* bf16_blob, const float* mfma_blob, const float* blob, const float* pos_enc,
                                         const float* xyz, const float* cam_ws, const float* src_rgbs, int H, int W, const float* featmap,
                                         int Hf, int Wf, int64_t fs_v, int64_t fs_c, int64_t fs_h, int64_t fs_w, int64_t n_rays,
                                         int n_samples, int n_views, int anti_alias_pooling, float* raw, float* workspace, float* mask_out,
                                         nf_stream_t stream) {
    if (n_rays == 0) return 0;
    NF_REQUIRE(xyz && cam_ws && src_rgbs && featmap && mask_out && H >= 1 && W >= 1 && Hf >= 1 && Wf >= 1,
               "nf_ibrnet_fwd_mfma_gather: bad arguments");
    NF_REQUIRE((int64_t)n_views * fs_v < ((int64_t)1 << 31) && (int64_t)n_views * H * W * 3 < ((int64_t)1 << 31),
               "nf_ibrnet_fwd_mfma_gather: feature maps / source images too large for 32-bit element offsets");
    NF_REQUIRE(fs_c == 1 && fs_v % 4 == 0 && fs_h % 4 == 0 && fs_w % 4 == 0 && ((uintptr_t)featmap) % 16 == 0,
               "nf_ibrnet_fwd_mfma_gather: the feature maps must be channels-last with 16-byte aligned pixel records");
    const RowGather g = {xyz, cam_ws, src_rgbs, featmap, mask_out, fs_v, fs_h, fs_w, H, W, Hf, Wf};
    return ibrnet_fwd_impl("nf_ibrnet_fwd_mfma_gather", bf16_blob, mfma_blob, blob, pos_enc, nullptr, nullptr, nullptr, n_rays, n_samples,
                           n_views, anti_alias_pooling, raw, workspace, stream, g);
}

/* bf16_blob: device copy of nf_ibrnet_pack_mfma_bf16's output; the per-(sample, view) row network runs on bf16 operands with
 * fp32 accumulation, everything else as nf_ibrnet_fwd_mfma */
extern "C" int nf_ibrnet_fwd_mfma_bf16(const float* bf16_blob, const float* mfma_blob, const float* blob, const float* pos_enc,
                                       const float* rgb_feat, const float* ray_diff, const float* mask, int64_t n_rays, int n_samples,
                                       int n_views, int anti_alias_pooling, float* raw, float* workspace, nf_stream_t stream) {
    NF_REQUIRE(bf16_blob != nullptr, "nf_ibrnet_fwd_mfma_bf16: bf16 blob missing");
    return ibrnet_fwd_impl("nf_ibrnet_fwd_mfma_bf16", bf16_blob, mfma_blob, blob, pos_enc, rgb_feat, ray_diff, mask, n_rays, n_samples,
                           n_views, anti_alias_pooling, raw, workspace, stream);
}

template <int V, bool BF, bool SCAT, bool GATH>
static int launch_rows_bwd(const float* wblob, const float* rgb_feat, const float* ray_diff, const float* mask,
                           const float* d_smp, int64_t n_samples, int nv, int aa, float* d_rgb_feat, const RowScatter& sc, const RowGather& ga,
                           hipStream_t st) {
    static bool configured_on[NF_MAX_DEVICES] = {};      // > 64 KB of dynamic LDS needs an explicit opt-in, once per kernel and device
    bool& configured = configured_on[nf_current_device()];
    // the weight image (+ one staging tile per wave for the fused scatter)
    constexpr int NWV = rows_bwd_waves(BF, SCAT);
    const size_t smem = ((BF ? (size_t)NF_BF_BLOB_FLOATS : (size_t)NF_ROWS_BLOB_FLOATS) + (SCAT ? NWV * RS_FLOATS : 0)) * sizeof(float);
    if (!configured) {
        if (hipFuncSetAttribute((const void*)k_ibr_rows_bwd<V, BF, SCAT, GATH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
            hipSuccess) {
            nf_set_error("nf_ibrnet_bwd_mfma: cannot reserve %zu bytes of LDS", smem);
            return 1;
        }
        configured = true;
    }
    int64_t tiles = (n_samples + 32 / V - 1) / (32 / V);
    int64_t blocks = (tiles + NWV - 1) / NWV;
    // fp32: one workgroup per CU holds the 113 KB weight image (fwd + transposed); bf16: 64 KB, two per CU (one of 8 waves with the scatter)
    const int64_t cap = 512 * rows_bwd_wgs_per_cu(BF, SCAT);          // two rounds of resident workgroups
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((k_ibr_rows_bwd<V, BF, SCAT, GATH>), dim3((unsigned)blocks), dim3(64 * NWV), smem, st, wblob, rgb_feat, ray_diff, mask,
                       d_smp, n_samples, nv, aa, d_rgb_feat, sc, ga);
    return 0;
}

template <int V>
static int launch_rows_bwd_any(const float* wblob, const float* bf_blob, const float* rgb_feat, const float* ray_diff, const float* mask,
                               const float* d_smp, int64_t n_samples, int nv, int aa, float* d_rgb_feat, const RowScatter& sc, const RowGather& ga,
                               hipStream_t st) {
    if (sc.d_featmap) {     // fused scatter
        return bf_blob ? launch_rows_bwd<V, true, true, false>(bf_blob, rgb_feat, ray_diff, mask, d_smp, n_samples, nv, aa, d_rgb_feat, sc, ga, st)
                       : launch_rows_bwd<V, false, true, false>(wblob, rgb_feat, ray_diff, mask, d_smp, n_samples, nv, aa, d_rgb_feat, sc, ga, st);
    }
    return bf_blob ? launch_rows_bwd<V, true, false, false>(bf_blob, rgb_feat, ray_diff, mask, d_smp, n_samples, nv, aa, d_rgb_feat, sc, ga, st)
                   : launch_rows_bwd<V, false, false, false>(wblob, rgb_feat, ray_diff, mask, d_smp, n_samples, nv, aa, d_rgb_feat, sc, ga, st);
}

template <int WPR, int NWV>
static int launch_ray_bwd_nw(const float* mfma_blob, const float* pos_enc, const float* smp, const float* d_raw, int64_t n_rays,
                             float* d_workspace, hipStream_t st) {
    constexpr int nw = WPR > NWV ? WPR : NWV, rpi = nw / WPR;
    int64_t iters = (n_rays + rpi - 1) / rpi;
    unsigned blocks = (unsigned)(iters < 256 ? iters : 256);
    size_t smem_ray = (size_t)(RY_FLOATS + nw * 32 * RAY_BWD_LDS_PER_SAMPLE) * sizeof(float);
    static bool configured_on[NF_MAX_DEVICES] = {};
    bool& configured = configured_on[nf_current_device()];
    if (!configured) {
        if (hipFuncSetAttribute((const void*)k_ibr_ray_bwd_mfma<WPR, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_ray) != hipSuccess) {
            nf_set_error("nf_ibrnet_bwd_mfma: cannot reserve %zu bytes of LDS", smem_ray);
            return 1;
        }
        configured = true;
    }
    hipLaunchKernelGGL((k_ibr_ray_bwd_mfma<WPR, NWV>), dim3(blocks), dim3(64 * nw), smem_ray, st, mfma_blob, pos_enc, smp, d_raw, n_rays, d_workspace);
    return 0;
}

template <int WPR>
static int launch_ray_bwd(const float* mfma_blob, const float* pos_enc, const float* smp, const float* d_raw, int64_t n_rays,
                          float* d_workspace, hipStream_t st) {
    // eight waves per workgroup once every CU still gets a workgroup that way
    if (WPR < 8 && n_rays * WPR >= 8 * 256) return launch_ray_bwd_nw<WPR, 8>(mfma_blob, pos_enc, smp, d_raw, n_rays, d_workspace, st);
    return launch_ray_bwd_nw<WPR, 4>(mfma_blob, pos_enc, smp, d_raw, n_rays, d_workspace, st);
}

static int ibrnet_bwd_impl(const char* who, const float* bf_blob, const float* mfma_blob, const float* blob, const float* pos_enc,
                           const float* rgb_feat, const float* ray_diff, const float* mask, const float* smp, const float* d_raw,
                           int64_t n_rays, int n_samples, int n_views, int anti_alias_pooling, float* d_rgb_feat,
                           float* d_workspace, nf_stream_t stream, const RowScatter& sc = RowScatter{}, const RowGather& ga = RowGather{}) {
    NF_REQUIRE(nf_ibrnet_mfma_supported(n_samples, n_views), "%s: 1 <= V <= 32 (got %d)", who, n_views);
    NF_REQUIRE(!ga.featmap || sc.d_featmap, "%s: the gathering recompute comes with the fused scatter", who);
    if (n_rays == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int threads = ((n_samples + 63) / 64) * 64;
    size_t smem = (size_t)n_samples * 77 * sizeof(float);
    const bool ray_mfma = n_samples == 32 || n_samples == 64 || n_samples == 128 || n_samples == 256;
    if (ray_mfma) {
        int rc;
        switch (n_samples / 32) {
            case 1: rc = launch_ray_bwd<1>(mfma_blob, pos_enc, smp, d_raw, n_rays, d_workspace, st); break;
            case 2: rc = launch_ray_bwd<2>(mfma_blob, pos_enc, smp, d_raw, n_rays, d_workspace, st); break;
            case 4: rc = launch_ray_bwd<4>(mfma_blob, pos_enc, smp, d_raw, n_rays, d_workspace, st); break;
            default: rc = launch_ray_bwd<8>(mfma_blob, pos_enc, smp, d_raw, n_rays, d_workspace, st); break;
        }
        if (rc) return rc;
    } else {
        NF_REQUIRE(smem <= 64 * 1024, "%s: S=%d exceeds the LDS budget of the ray kernel", who, n_samples);
        if (threads <= 256)
            hipLaunchKernelGGL(k_ibr_ray_bwd<256>, dim3((unsigned)n_rays), dim3(threads), smem, st, blob, pos_enc, smp, d_raw,
                               n_samples, d_workspace);
        else
            hipLaunchKernelGGL(k_ibr_ray_bwd<NF_IBR_MAX_S>, dim3((unsigned)n_rays), dim3(threads), smem, st, blob, pos_enc, smp,
                               d_raw, n_samples, d_workspace);
    }
    NF_LAUNCH_CHECK("nf_ibrnet_bwd_mfma (ray)");
    int64_t ns = n_rays * n_samples;
    int rc;
    switch (rows_lanes_per_sample(n_views)) {
        case 1: rc = launch_rows_bwd_any<1>(mfma_blob, bf_blob, rgb_feat, ray_diff, mask, d_workspace, ns, n_views, anti_alias_pooling, d_rgb_feat, sc, ga, st); break;
        case 2: rc = launch_rows_bwd_any<2>(mfma_blob, bf_blob, rgb_feat, ray_diff, mask, d_workspace, ns, n_views, anti_alias_pooling, d_rgb_feat, sc, ga, st); break;
        case 4: rc = launch_rows_bwd_any<4>(mfma_blob, bf_blob, rgb_feat, ray_diff, mask, d_workspace, ns, n_views, anti_alias_pooling, d_rgb_feat, sc, ga, st); break;
        case 8: rc = launch_rows_bwd_any<8>(mfma_blob, bf_blob, rgb_feat, ray_diff, mask, d_workspace, ns, n_views, anti_alias_pooling, d_rgb_feat, sc, ga, st); break;
        case 16: rc = launch_rows_bwd_any<16>(mfma_blob, bf_blob, rgb_feat, ray_diff, mask, d_workspace, ns, n_views, anti_alias_pooling, d_rgb_feat, sc, ga, st); break;
        default: rc = launch_rows_bwd_any<32>(mfma_blob, bf_blob, rgb_feat, ray_diff, mask, d_workspace, ns, n_views, anti_alias_pooling, d_rgb_feat, sc, ga, st); break;
    }
    if (rc) return rc;
    NF_LAUNCH_CHECK("nf_ibrnet_bwd_mfma (rows)");
    return 0;
}

/* smp = the per-sample records the forward left in its workspace; d_workspace: same size, receives d(records) */
extern "C" int nf_ibrnet_bwd_mfma(const float* mfma_blob, const float* blob, const float* pos_enc, const float* rgb_feat,
                                  const float* ray_diff, const float* mask, const float* smp, const float* d_raw,
                                  int64_t n_rays, int n_samples, int n_views, int anti_alias_pooling, float* d_rgb_feat,
                                  float* d_workspace, nf_stream_t stream) {
    return ibrnet_bwd_impl("nf_ibrnet_bwd_mfma", nullptr, mfma_blob, blob, pos_enc, rgb_feat, ray_diff, mask, smp, d_raw, n_rays, n_samples,
                           n_views, anti_alias_pooling, d_rgb_feat, d_workspace, stream);
}

/* nf_ibrnet_bwd_mfma with the scatter of d rgb_feat into the feature-map gradient fused into the row kernel (d rgb_feat is not
 * written): xyz [n_rays * n_samples][3] and cam_ws as given to nf_project_gather_fwd, d_featmap [V][32][Hf][Wf] through element
 * strides, ZEROED by the caller (the kernel adds). */
extern "C" int nf_ibrnet_bwd_mfma_scatter(const float* bf16_blob, const float* mfma_blob, const float* blob, const float* pos_enc, const float* rgb_feat,
                                          const float* ray_diff, const float* mask, const float* smp, const float* d_raw, int64_t n_rays,
                                          int n_samples, int n_views, int anti_alias_pooling, float* d_workspace, const float* xyz,
                                          const float* cam_ws, float* d_featmap, int64_t fs_v, int64_t fs_c, int64_t fs_h, int64_t fs_w,
                                          int Hf, int Wf, nf_stream_t stream) {
    if (n_rays == 0) return 0;       // nothing to add
    NF_REQUIRE(xyz && cam_ws && d_featmap && Hf >= 1 && Wf >= 1, "nf_ibrnet_bwd_mfma_scatter: bad arguments");
    const RowScatter sc = {xyz, cam_ws, d_featmap, fs_v, fs_c, fs_h, fs_w, Hf, Wf};
    return ibrnet_bwd_impl("nf_ibrnet_bwd_mfma_scatter", bf16_blob, mfma_blob, blob, pos_enc, rgb_feat, ray_diff, mask, smp, d_raw, n_rays,
                           n_samples, n_views, anti_alias_pooling, nullptr, d_workspace, stream, sc);
}

extern "C" int nf_ibrnet_bwd_mfma_bf16(const float* bf16_blob, const float* mfma_blob, const float* blob, const float* pos_enc,
                                       const float* rgb_feat, const float* ray_diff, const float* mask, const float* smp,
                                       const float* d_raw, int64_t n_rays, int n_samples, int n_views, int anti_alias_pooling,
                                       float* d_rgb_feat, float* d_workspace, nf_stream_t stream) {
    NF_REQUIRE(bf16_blob != nullptr, "nf_ibrnet_bwd_mfma_bf16: bf16 blob missing");
    return ibrnet_bwd_impl("nf_ibrnet_bwd_mfma_bf16", bf16_blob, mfma_blob, blob, pos_enc, rgb_feat, ray_diff, mask, smp, d_raw, n_rays,
                           n_samples, n_views, anti_alias_pooling, d_rgb_feat, d_workspace, stream);
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
