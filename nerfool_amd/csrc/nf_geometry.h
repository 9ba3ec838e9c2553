// Per-point geometry of the IBRNet projector: camera table, projection, view-direction deltas, bilinear taps.
// Restates ibrnet/projection.py:24-87,112-131 (reference paths relative to the NeRFool tree); the operation order
// follows the reference's fp32 tensor ops (see DESIGN.md "numerics") so that pixel locations agree to ~1 ulp.
#pragma once
#include "nf_common.h"

#define NF_CAM_STRIDE 16

// cam34 = [H, W, K(16, row-major 4x4), c2w(16)].  out[0..11] = rows 0..2 of K * inv(c2w), out[12..14] = centre.
// The 4x4 inverse is a cofactor expansion in double (the reference uses fp32 LU, torch.inverse, :56).
NF_HD void nf_camera_entry(const float* cam34, float* out) {
    double m[16], inv[16], K[16];
    for (int i = 0; i < 16; ++i) {
        K[i] = (double)cam34[2 + i];
        m[i] = (double)cam34[18 + i];
    }
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    double rdet = 1.0 / det;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc += K[i * 4 + k] * (inv[k * 4 + j] * rdet);
            out[i * 4 + j] = (float)acc;
        }
    out[12] = cam34[18 + 3];
    out[13] = cam34[18 + 7];
    out[14] = cam34[18 + 11];
    out[15] = 0.f;
}

// ref: ibrnet/projection.py:56-61.  Returns pixel location (clamped to +-1e6) and the in-front flag.
NF_HD void nf_project_point(const float* cam, float x, float y, float z, float& px, float& py, bool& front) {
    float p0 = fmaf(cam[2], z, fmaf(cam[1], y, cam[0] * x)) + cam[3];
    float p1 = fmaf(cam[6], z, fmaf(cam[5], y, cam[4] * x)) + cam[7];
    float p2 = fmaf(cam[10], z, fmaf(cam[9], y, cam[8] * x)) + cam[11];
    float d = fmaxf(p2, 1e-8f);
    px = fminf(fmaxf(p0 / d, -1e6f), 1e6f);
    py = fminf(fmaxf(p1 / d, -1e6f), 1e6f);
    front = p2 > 0.f;
}

// ref: ibrnet/projection.py:32-35 (inbound) -- inclusive on both ends, in SOURCE-IMAGE pixel units.
NF_HD bool nf_inbound(float px, float py, float h, float w) {
    return (px <= w - 1.f) && (px >= 0.f) && (py <= h - 1.f) && (py >= 0.f);
}

// ref: ibrnet/projection.py:72-85.  cq/cv: query / source camera centres.  out[0..2] unit difference, out[3] dot.
// The three normalisations multiply by ONE reciprocal each (v_rcp_f32, 1 ulp) instead of dividing three times (an IEEE division is
// a ten-instruction sequence on the vector pipe, and vector instructions are paid in matrix-pipe time): the direction features move by
// <= 2 ulp, which feeds a continuous MLP -- unlike the projection and the tap positions above / below, whose divisions stay exact because
// they decide masks and tap indices.
NF_HD void nf_ray_diff(const float* cq, const float* cv, float x, float y, float z, float* out) {
    float tx = cq[0] - x, ty = cq[1] - y, tz = cq[2] - z;
    const float rt = nf_rcp(sqrtf(tx * tx + ty * ty + tz * tz) + 1e-6f);
    tx = tx * rt; ty = ty * rt; tz = tz * rt;
    float sx = cv[0] - x, sy = cv[1] - y, sz = cv[2] - z;
    const float rs = nf_rcp(sqrtf(sx * sx + sy * sy + sz * sz) + 1e-6f);
    sx = sx * rs; sy = sy * rs; sz = sz * rs;
    float dx = tx - sx, dy = ty - sy, dz = tz - sz;
    const float rd = nf_rcp(fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-6f));
    out[0] = dx * rd;
    out[1] = dy * rd;
    out[2] = dz * rd;
    out[3] = tx * sx + ty * sy + tz * sz;
}

// Bilinear taps of F.grid_sample(align_corners=True, padding 'zeros') for a map of size (Hm, Wm) sampled with the
// grid normalised by the SOURCE-IMAGE size (h, w) -- the same grid reads the RGB images and the feature maps
// (ibrnet/projection.py:112-123).  Tap order nw, ne, sw, se; weight 0 marks an out-of-map tap.
struct NfTaps {
    int x0, y0;        // north-west integer corner (may be outside the map)
    float w[4];        // nw, ne, sw, se (already zeroed when the tap is outside)
    bool in[4];
};

NF_HD NfTaps nf_bilinear_taps(float px, float py, float h, float w, int Hm, int Wm) {
    float nx = 2.f * px / (w - 1.f) - 1.f;
    float ny = 2.f * py / (h - 1.f) - 1.f;
    float ix = ((nx + 1.f) / 2.f) * (float)(Wm - 1);
    float iy = ((ny + 1.f) / 2.f) * (float)(Hm - 1);
    float fx = floorf(ix), fy = floorf(iy);
    NfTaps t;
    // keep the int conversion safe for the +-1e6-clamped coordinates of points behind a camera
    t.x0 = (int)fminf(fmaxf(fx, -2.0e9f), 2.0e9f);
    t.y0 = (int)fminf(fmaxf(fy, -2.0e9f), 2.0e9f);
    float ax = ix - fx, ay = iy - fy;          // distance to the west / north tap
    float bx = (fx + 1.f) - ix, by = (fy + 1.f) - iy;
    bool xin0 = t.x0 >= 0 && t.x0 <= Wm - 1, xin1 = t.x0 + 1 >= 0 && t.x0 + 1 <= Wm - 1;
    bool yin0 = t.y0 >= 0 && t.y0 <= Hm - 1, yin1 = t.y0 + 1 >= 0 && t.y0 + 1 <= Hm - 1;
    t.in[0] = xin0 && yin0; t.in[1] = xin1 && yin0; t.in[2] = xin0 && yin1; t.in[3] = xin1 && yin1;
    t.w[0] = t.in[0] ? bx * by : 0.f;
    t.w[1] = t.in[1] ? ax * by : 0.f;
    t.w[2] = t.in[2] ? bx * ay : 0.f;
    t.w[3] = t.in[3] ? ax * ay : 0.f;
    return t;
}

// ref: ibrnet/render_ray.py:87-110 (closed form of the python list-comprehension; det=False adds the jitter)
NF_HD float nf_coarse_depth(float near, float far, int s, int S, int inv_uniform) {
    if (inv_uniform) {
        float start = 1.f / near;
        float step = (1.f / far - start) / (float)(S - 1);
        return 1.f / (start + (float)s * step);
    }
    float step = (far - near) / (float)(S - 1);
    return near + (float)s * step;
}
