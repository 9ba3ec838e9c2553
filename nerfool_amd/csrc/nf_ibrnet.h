// IBRNet (ibrnet/mlp_network.py:152-274) parameter blob and workspace layout shared by the kernels and the host-side
// packer.  Every Linear is stored twice -- transposed [in][out] (forward: one input feature feeds `out` contiguous
// weights) and native [out][in] (backward-data: one output gradient feeds `in` contiguous weights) -- then its bias.
#pragma once
#include "nf_common.h"

struct NfLin { int in, out; const char* key; };

enum {
    NF_L_DIR0, NF_L_DIR1, NF_L_BASE0, NF_L_BASE1, NF_L_VIS0, NF_L_VIS1, NF_L_VISB0, NF_L_VISB1, NF_L_GEO0, NF_L_GEO1,
    NF_L_OG0, NF_L_OG1, NF_L_RGB0, NF_L_RGB1, NF_L_RGB2, NF_N_LIN
};

static constexpr NfLin NF_LIN[NF_N_LIN] = {
    {4, 16, "ray_dir_fc.0"},  {16, 35, "ray_dir_fc.2"}, {105, 64, "base_fc.0"},      {64, 32, "base_fc.2"},
    {32, 32, "vis_fc.0"},     {32, 33, "vis_fc.2"},     {32, 32, "vis_fc2.0"},       {32, 1, "vis_fc2.2"},
    {65, 64, "geometry_fc.0"}, {64, 16, "geometry_fc.2"}, {16, 16, "out_geometry_fc.0"}, {16, 1, "out_geometry_fc.2"},
    {37, 16, "rgb_fc.0"},     {16, 8, "rgb_fc.2"},      {8, 1, "rgb_fc.4"}};

// offset of Linear l inside the blob (float index); [0] is the pooling scalar `s`
NF_HD constexpr int nf_lin_off(int l) {
    int o = 1;
    for (int i = 0; i < l; ++i) o += 2 * NF_LIN[i].in * NF_LIN[i].out + NF_LIN[i].out;
    return o;
}
NF_HD constexpr int nf_lin_wt(int l) { return nf_lin_off(l); }                                        // [in][out]
NF_HD constexpr int nf_lin_w(int l) { return nf_lin_off(l) + NF_LIN[l].in * NF_LIN[l].out; }          // [out][in]
NF_HD constexpr int nf_lin_b(int l) { return nf_lin_off(l) + 2 * NF_LIN[l].in * NF_LIN[l].out; }      // [out]

// ray_attention: w_qs, w_ks, w_vs, fc (16x16, no bias), each transposed then native; then LayerNorm weight, bias
static constexpr int NF_ATT_OFF = nf_lin_off(NF_N_LIN);
NF_HD constexpr int nf_att_wt(int m) { return NF_ATT_OFF + m * 512; }
NF_HD constexpr int nf_att_w(int m) { return NF_ATT_OFF + m * 512 + 256; }
static constexpr int NF_LN_W = NF_ATT_OFF + 4 * 512;
static constexpr int NF_LN_B = NF_LN_W + 16;
static constexpr int NF_BLOB_FLOATS = NF_LN_B + 16;
static constexpr int NF_BLOB_ENTRIES = 1 + NF_N_LIN * 3 + 4 * 2 + 2;

// ---- workspace slots of the generic (any V, any S <= 1024) kernels: floats per (sample, view) row ...
enum {
    NF_R_F = 0,            // 35  rgb_feat + direction feature
    NF_R_W = 35,           // 1   first pooling weight
    NF_R_H1 = 36,          // 64  base_fc.0 out (post-ELU)
    NF_R_H = 100,          // 32  base_fc.2 out
    NF_R_V1 = 132,         // 32  vis_fc.0 out
    NF_R_XV = 164,         // 33  vis_fc.2 out
    NF_R_X2 = 197,         // 32  h + residual
    NF_R_U = 229,          // 32  vis_fc2.0 out
    NF_R_SIG1 = 261, NF_R_VIS1 = 262, NF_R_SIG2 = 263, NF_R_VIS2 = 264, NF_R_W2 = 265,
    NF_R_R1 = 266,         // 16  rgb_fc.0 out
    NF_R_R2 = 282,         // 8   rgb_fc.2 out
    NF_R_BETA = 290,       // 1   blending weight (holds exp_dot_prod / the logit earlier)
    NF_ROW_FWD = 291,
    NF_R_DX2 = 291,        // 32  backward only from here
    NF_R_DVIS2 = 323, NF_R_DW2 = 324,
    NF_R_DF = 325,         // 35
    NF_R_GR = 360,         // 64  gradient scratch
    NF_ROW_BWD = 424
};
// ... and floats per sample
enum {
    NF_S_MEAN = 0, NF_S_VAR = 35, NF_S_GIN = 70 /* mean2 32, var2 32, wmean 1 */, NF_S_G1 = 135, NF_S_G = 199,
    NF_S_GPE = 215, NF_S_Q = 231, NF_S_K = 247, NF_S_V = 263, NF_S_O = 279, NF_S_XHAT = 295, NF_S_GAT = 311,
    NF_S_OG1 = 327, NF_S_RSTD = 343, NF_S_SIGPRE = 344, NF_S_NVAL = 345, NF_S_VSUM = 346, NF_S_M = 347 /*4*/,
    NF_S_L = 351 /*4*/, NF_SMP_FWD = 355,
    NF_S_DO = 355, NF_S_DH = 371 /*4*/, NF_S_DQ = 375, NF_S_DGIN = 391 /*65*/, NF_S_GS = 456 /*64*/, NF_SMP_BWD = 520
};

#define NF_IBR_MAX_S 1024
#define NF_IBR_RAYS_PER_LAUNCH 256
