// Shared layout of the GNT kernels (nf_gnt.hip: shape-generic VALU kernels; nf_gnt_mfma.hip: matrix-core forward): the
// natural parameter blob and the per-ray workspace the forward fills for the backward.
#pragma once
#include "nf_common.h"

#define GNT_C 64
#define GNT_MAX_S 1024

// ---------------------------------------------------------------------------------------------------------------
// parameter blob: [stem | layer 0 | layer 1 | ... | final]; every Linear as Wt [in][out], W [out][in], bias
// ---------------------------------------------------------------------------------------------------------------
struct GntLin { int in, out, bias; const char* key; };

enum { GV_Q, GV_K, GV_V, GV_POS0, GV_POS2, GV_ATT0, GV_ATT2, GV_OUT, GV_FF1, GV_FF2, GQ_0, GQ_2, GR_Q, GR_K, GR_V, GR_OUT, GR_FF1,
       GR_FF2, G_NLIN };

static constexpr GntLin GNT_LAYER[G_NLIN] = {
    {64, 64, 0, "view_crosstrans.%d.attn.q_fc"},   {64, 64, 0, "view_crosstrans.%d.attn.k_fc"},
    {64, 64, 0, "view_crosstrans.%d.attn.v_fc"},   {4, 8, 1, "view_crosstrans.%d.attn.pos_fc.0"},
    {8, 64, 1, "view_crosstrans.%d.attn.pos_fc.2"}, {64, 8, 1, "view_crosstrans.%d.attn.attn_fc.0"},
    {8, 64, 1, "view_crosstrans.%d.attn.attn_fc.2"}, {64, 64, 1, "view_crosstrans.%d.attn.out_fc"},
    {64, 256, 1, "view_crosstrans.%d.ff.fc1"},     {256, 64, 1, "view_crosstrans.%d.ff.fc2"},
    {190, 64, 1, "q_fcs.%d.0"},                    {64, 64, 1, "q_fcs.%d.2"},
    {64, 64, 0, "view_selftrans.%d.attn.q_fc"},    {64, 64, 0, "view_selftrans.%d.attn.k_fc"},
    {64, 64, 0, "view_selftrans.%d.attn.v_fc"},    {64, 64, 1, "view_selftrans.%d.attn.out_fc"},
    {64, 256, 1, "view_selftrans.%d.ff.fc1"},      {256, 64, 1, "view_selftrans.%d.ff.fc2"}};
static constexpr const char* GNT_LN_KEYS[4] = {"view_crosstrans.%d.attn_norm", "view_crosstrans.%d.ff_norm",
                                               "view_selftrans.%d.attn_norm", "view_selftrans.%d.ff_norm"};
static constexpr GntLin GNT_STEM[2] = {{35, 64, 1, "rgbfeat_fc.0"}, {64, 64, 1, "rgbfeat_fc.2"}};

NF_HD constexpr int gnt_lin_size(const GntLin& l) { return 2 * l.in * l.out + (l.bias ? l.out : 0); }
NF_HD constexpr int gnt_lin_off(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += gnt_lin_size(GNT_LAYER[i]);
    return o;
}
NF_HD constexpr int gnt_wt(int l) { return gnt_lin_off(l); }
NF_HD constexpr int gnt_w(int l) { return gnt_lin_off(l) + GNT_LAYER[l].in * GNT_LAYER[l].out; }
NF_HD constexpr int gnt_b(int l) { return gnt_lin_off(l) + 2 * GNT_LAYER[l].in * GNT_LAYER[l].out; }
static constexpr int GNT_LN_OFF = gnt_lin_off(G_NLIN);          // 4 x (weight 64, bias 64)
static constexpr int GNT_LAYER_FLOATS = GNT_LN_OFF + 4 * 128;
static constexpr int GNT_STEM1 = gnt_lin_size(GNT_STEM[0]);
static constexpr int GNT_STEM_FLOATS = GNT_STEM1 + gnt_lin_size(GNT_STEM[1]);
NF_HD constexpr int gnt_ln_w(int j) { return GNT_LN_OFF + j * 128; }
NF_HD constexpr int gnt_ln_b(int j) { return GNT_LN_OFF + j * 128 + 64; }
// final block: norm weight 64, bias 64, rgb_fc Wt [64][3], W [3][64], bias 3
static constexpr int GNT_FINAL_FLOATS = 128 + 192 + 192 + 3 + 1;
NF_HD constexpr int64_t gnt_layer_base(int i) { return (int64_t)GNT_STEM_FLOATS + (int64_t)i * GNT_LAYER_FLOATS; }

// ---------------------------------------------------------------------------------------------------------------
// training-mode Dropout (round 5).  The reference's universal GNT loop runs before model.switch_to_eval()
// (eval/gnt/eval_adv.py:739-878 vs :959): the eight nn.Dropout(0.1) sites of every layer are live
// (gnt/transformer_network.py:45-48 feed-forward hidden / output, :85-88 view-attention probabilities / output, :162-166 ray-attention
// probabilities / output).  torch's generator cannot be matched on a GPU; the masks come from a counter-based generator instead --
// keep(seed, site, idx), a 32-bit integer hash (murmur3 finaliser rounds) of the call's seed, the site 8 * layer + {0 view probabilities
// [R,S,V,64], 1 view output [R,S,64], 2 view hidden [R,S,256], 3 view feed-forward output [R,S,64], 4 ray probabilities [R,4,S,S],
// 5 ray output, 6 ray hidden, 7 ray feed-forward output} and the element's flat index in the tensor the reference hands to nn.Dropout --
// so the forward and the backward regenerate the same mask, and the oracle (oracle/gnt_ref.py:keep_mask) and the reference itself
// (tests/golden/make_golden_gnt_train.py injects the function into its modules) evaluate identical masks.
// ---------------------------------------------------------------------------------------------------------------
struct GntDrop {
    unsigned seed, thr;       // keep when (hash >> 8) >= thr, thr = p * 2^24
    float scale;              // 1 / (1 - p)
    int on;
};
NF_HD unsigned gnt_mix32(unsigned x) {
    x ^= x >> 16;
    x *= 0x85ebca6bu;
    x ^= x >> 13;
    x *= 0xc2b2ae35u;
    x ^= x >> 16;
    return x;
}
NF_HD float gnt_keep(const GntDrop& d, unsigned site, unsigned long long idx) {
    if (!d.on) return 1.f;
    unsigned a = gnt_mix32(d.seed ^ (site * 0x9e3779b9u));
    a = gnt_mix32(a ^ (unsigned)(idx & 0xffffffffull));
    a = gnt_mix32(a + (unsigned)(idx >> 32) * 0x7f4a7c15u + 0x165667b1u);
    return (a >> 8) >= d.thr ? d.scale : 0.f;
}
static inline GntDrop gnt_drop_make(int train, unsigned seed, double p) {
    GntDrop d;
    d.on = train ? 1 : 0;
    d.seed = seed;
    d.thr = (unsigned)(p * 16777216.0);
    d.scale = 1.0f / (float)(1.0 - p);
    return d;
}

// ---------------------------------------------------------------------------------------------------------------
// workspace slots
// ---------------------------------------------------------------------------------------------------------------
enum { RW_R1 = 0, RW_X = 64, RW_DX = 128, RW_T = 192, RW_T2 = 256, RW_T3 = 320, RW_R1H = 384, RW_BASE = 392, RWL_VP = 0, RWL_H = 64, RWL_PROB = 72, RW_LAYER = 136 };
enum { SW_AMAX = 0, SW_CUR = 64, SW_PE = 128, SW_T = 256, SW_U = 512, SW_QV = 576, SW_XHF = 640, SW_RSTDF = 704, SW_HF = 705,
       SW_DCUR = 769, SW_DU = 833, SW_SP = 897, SW_DQS = 961, SW_DQ = 1025, SW_DK = 1089, SW_DV = 1153, SW_GO = 1217, SW_DH = 1281,
       SW_BASE = 1288,
       SL_XH1 = 0, SL_RSTD1 = 64, SL_XH2 = 65, SL_RSTD2 = 129, SL_F = 130, SL_G = 386, SL_RXH1 = 450, SL_RRSTD1 = 514, SL_QH = 515,
       SL_KH = 579, SL_VH = 643, SL_ML = 707, SL_OUTA = 715, SL_RXH2 = 779, SL_RRSTD2 = 843, SL_F2 = 844,
       // written by the matrix-core forward for the matrix-core backward only (which in turn reads RWL_PROB as the masked LOGITS and
       // SL_F / SL_F2 / RW_R1 as sign-bit words -- see nf_gnt_mfma.hip): view-attention output before out_fc, running maximum and
       // reciprocal sum of the view softmax
       SL_U = 1100, SL_MX = 1164, SL_RS = 1228,
       SW_LAYER = 1292 };

static int64_t gnt_row_floats(int depth, int save) { return RW_BASE + (int64_t)(save ? depth : 1) * RW_LAYER; }
static int64_t gnt_smp_floats(int depth, int save) { return SW_BASE + (int64_t)(save ? depth : 1) * SW_LAYER; }

// rays per launch when nothing is saved for a backward: bounds the recycled workspace (about 2 MB per ray at S = 64, V = 10)
#define GNT_RAYS_PER_LAUNCH 1024
