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
