// The ResUNet's four stride-2 convolutions (a14) -- the 7x7 stem and the first 3x3 convolution of layer1/2/3 -- forward and
// backward-data, as direct (implicit-GEMM) convolutions on the fp32 matrix cores.
// ref: ibrnet/feature_network.py:188 (conv1: 7x7, stride 2), :192-195 via :51 (BasicBlock.conv1 with stride 2).  The reflect
// padding is already in the input (csrc/nf_cnn.hip writes pre-padded activations), so the convolution runs with padding 0.
//
//   forward   y[n,k,oy,ox]  = sum_{c,a,b} W[k][c][a][b] x[n,c,2 oy + a, 2 ox + b]
//   backward  dx[n,c,Y,X]   = sum_{k} sum_{a = Y (mod 2), b = X (mod 2)} W[k][c][a][b] dy[n,k,(Y-a)/2,(X-b)/2]
//
// GEMM formulation as everywhere in this library (v_mfma_f32_32x32x2_f32): A operand = weights, one host-packed record of
// 64 floats per k-step in consumption order; B operand = activations with the pixel on the lane (lane & 31) and the k pair on
// the lane half.  Forward: a workgroup owns 8 x 32 (or 4 x 32) output pixels x 64 output channels; the raw input window of a
// channel chunk is staged in LDS DE-INTERLEAVED by column parity, so that the stride-2 reads of consecutive lanes hit
// consecutive addresses; every B read is `lane base + compile-time immediate`.  Backward: the four parity classes (Y mod 2,
// X mod 2) of dx are four stride-1 correlations of dy with 2x2 / 2x1 / 1x2 / 1x1 (3x3) or 4x4 / 4x3 / 3x4 / 3x3 (7x7) taps;
// a lane keeps the accumulators of all four classes of its (u, v) position, so the two column classes leave as one 8-byte
// store and dx is written exactly once, zeros included.  Two to four workgroups per CU overlap one another's staging.
#include <string.h>

#include "nf_common.h"

typedef float s2_f16 __attribute__((ext_vector_type(16)));
typedef float s2_f2 __attribute__((ext_vector_type(2), aligned(4)));
typedef float s2_f4 __attribute__((ext_vector_type(4)));
#define S2_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__host__ __device__ constexpr int s2_nidx(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

#define S2_STEM_CK 8        // dy channels per chunk of the 7x7 backward-data kernel
#define S2_PW 36            // LDS stride of one column-parity plane of a window row (forward) / of a dy window row (backward)

// ---- forward geometry ------------------------------------------------------------------------------------------------
// KS = 3: k-step = (channel pair p, a, b), lane half = channel parity; 8 channels per chunk.
// KS = 7: 3 input channels -- k-step = (c, a, column pair q), lane half = column parity (b = 2 q + h; b = 7 has zero weights).
template <int KS> struct S2Fwd {
    static constexpr bool PAIR_B = KS == 7;
    static constexpr int CC = KS == 7 ? 3 : 8;                                  // channels per chunk
    static constexpr int STEPS = PAIR_B ? CC * KS * ((KS + 1) / 2) : (CC / 2) * KS * KS;
    static constexpr int WC = 62 + KS + (PAIR_B ? 1 : 0);                       // window columns used by 32 output columns
};
template <int KS, int RPW> struct S2FwdLds {       // RPW = output rows per wave (2: 8 x 32 pixels per workgroup, 1: 4 x 32)
    static constexpr int WR = 2 * (4 * RPW - 1) + KS;                           // window rows
    static constexpr int WIN = S2Fwd<KS>::CC * WR * 2 * S2_PW;                  // floats
    static constexpr int WGT = S2Fwd<KS>::STEPS * 2 * 64;
    static constexpr int FLOATS = WIN + WGT;
};

extern "C" int64_t nf_conv_s2_pack_floats(int c_out, int c_in, int ks, int backward) {
    if (ks != 3 && ks != 7) return -1;
    if (!backward) {
        const int cc = ks == 7 ? 3 : 8, steps = ks == 7 ? S2Fwd<7>::STEPS : S2Fwd<3>::STEPS;
        const int64_t groups = (c_out + 63) / 64, chunks = (c_in + cc - 1) / cc;
        return groups * chunks * steps * 2 * 64;
    }
    if (ks == 7) return (int64_t)((c_out + S2_STEM_CK - 1) / S2_STEM_CK) * (S2_STEM_CK / 4) * 16 * 64;      // stem form, see below
    const int ck = 16;
    const int64_t groups = (c_in + 31) / 32, chunks = (c_out + ck - 1) / ck;
    return groups * chunks * (ck / 2) * ks * ks * 64;
}

/* HOST: weight [c_out][c_in][ks][ks] -> records in consumption order.
 * forward:  [group of 64 outputs][chunk][step][tile t][lane (i, h)] = W[64 g + 32 t + i][c][a][b]
 * backward: [group of 32 inputs ][chunk of CK outputs][class (ya, xb)][tap (i, j)][pair p][lane (i_c, h)]
 *           = W[k = chunk CK + 2 p + h][c = 32 g + i_c][a = 2 i + ya][b = 2 j + xb]
 * backward, 7x7 (c_in <= 4): [chunk of 8 outputs][group of 4 outputs gq][tap (i, j) of 4 x 4][lane (m, q)], m = 4 c + 2 ya + xb
 *           = W[k = chunk 8 + 4 gq + q][c][2 i + ya][2 j + xb]  (v_mfma_f32_16x16x4_f32 A operand: row m = lane & 15, k = lane >> 4;
 *           zero where the tap lies outside the 7 x 7 kernel or m >= 4 c_in) */
extern "C" int nf_conv_s2_pack(const float* weight, int c_out, int c_in, int ks, int backward, float* out) {
    if (ks != 3 && ks != 7) return 1;
    auto W = [&](int k, int c, int a, int b) -> float {
        return (k < c_out && c < c_in && a < ks && b < ks) ? weight[(((size_t)k * c_in + c) * ks + a) * ks + b] : 0.f;
    };
    float* rec = out;
    if (!backward) {
        const bool pair_b = ks == 7;
        const int cc = pair_b ? 3 : 8;
        const int groups = (c_out + 63) / 64, chunks = (c_in + cc - 1) / cc;
        for (int g = 0; g < groups; ++g)
            for (int ch = 0; ch < chunks; ++ch) {
                auto emit = [&](int c0, int c1, int a, int b0, int b1) {
                    for (int t = 0; t < 2; ++t)
                        for (int lane = 0; lane < 64; ++lane, ++rec) {
                            const int i = lane & 31, h = lane >> 5;
                            *rec = W(64 * g + 32 * t + i, h ? c1 : c0, a, h ? b1 : b0);
                        }
                };
                if (pair_b) {
                    for (int c = 0; c < cc; ++c)
                        for (int a = 0; a < ks; ++a)
                            for (int q = 0; q < (ks + 1) / 2; ++q) emit(ch * cc + c, ch * cc + c, a, 2 * q, 2 * q + 1);
                } else {
                    for (int p = 0; p < cc / 2; ++p)
                        for (int a = 0; a < ks; ++a)
                            for (int b = 0; b < ks; ++b) emit(ch * cc + 2 * p, ch * cc + 2 * p + 1, a, b, b);
                }
            }
    } else if (ks == 7) {
        const int chunks = (c_out + S2_STEM_CK - 1) / S2_STEM_CK;
        for (int ch = 0; ch < chunks; ++ch)
            for (int gq = 0; gq < S2_STEM_CK / 4; ++gq)
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j)
                        for (int lane = 0; lane < 64; ++lane, ++rec) {
                            const int m = lane & 15, c = m >> 2, ya = (m >> 1) & 1, xb = m & 1;
                            *rec = W(ch * S2_STEM_CK + 4 * gq + (lane >> 4), c, 2 * i + ya, 2 * j + xb);
                        }
    } else {
        const int ck = 16;
        const int groups = (c_in + 31) / 32, chunks = (c_out + ck - 1) / ck;
        for (int g = 0; g < groups; ++g)
            for (int ch = 0; ch < chunks; ++ch)
                for (int ya = 0; ya < 2; ++ya)
                    for (int xb = 0; xb < 2; ++xb)
                        for (int i = 0; 2 * i + ya < ks; ++i)
                            for (int j = 0; 2 * j + xb < ks; ++j)
                                for (int p = 0; p < ck / 2; ++p)
                                    for (int lane = 0; lane < 64; ++lane, ++rec)
                                        *rec = W(ch * ck + 2 * p + (lane >> 5), 32 * g + (lane & 31), 2 * i + ya, 2 * j + xb);
    }
    return (rec - out) == nf_conv_s2_pack_floats(c_out, c_in, ks, backward) ? 0 : 2;
}

struct S2Tensor { int64_t ns, cs, rs; };        // element strides: image, channel, row (unit column stride)

template <int KS, int RPW>
__global__ void __launch_bounds__(256, 2) k_conv_s2_fwd(const float* __restrict__ rec, const float* __restrict__ x, S2Tensor xi, int Hi, int Wi,
                                                        float* __restrict__ y, S2Tensor yo, int Ho, int Wo, int C, int K, int groups,
                                                        int tiles_x, int tiles_y) {
    using G = S2Fwd<KS>;
    using L = S2FwdLds<KS, RPW>;
    constexpr int CC = G::CC, WR = L::WR, WC = G::WC, ROWF = 2 * S2_PW, CHF = WR * ROWF;
    HIP_DYNAMIC_SHARED(float, smem)
    float* win = smem;
    float* wgt = smem + L::WIN;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    int bid = blockIdx.x;
    const int grp = bid % groups;
    bid /= groups;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y, n = bid / tiles_y;
    const int oy0 = ty * (4 * RPW), ox0 = tx * 32;
    const int iy0 = 2 * oy0, ix0 = 2 * ox0;
    const int chunks = (C + CC - 1) / CC;
    const float* xn = x + n * xi.ns;
    const float* wsrc = rec + (size_t)grp * chunks * (G::STEPS * 128);
    // B read: lane base + immediate (see the header comment)
    const int bbase = (G::PAIR_B ? h * S2_PW : h * CHF) + (2 * RPW * w) * ROWF + j;
    // 8-byte window loads need 8-byte aligned rows: even strides and an 8-byte aligned base (true for the network's activations)
    const bool pair_ok = ((xi.ns | xi.cs | xi.rs) & 1) == 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0;

    s2_f16 acc[RPW][2];
#pragma unroll
    for (int b = 0; b < RPW; ++b)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][t][r] = 0.f;

    // Staging is software-pipelined: the global loads of chunk ch + 1 (window pairs, extra columns, weight records) are issued
    // into registers right behind the barrier that publishes chunk ch and are only waited for when they are committed to LDS
    // at the top of the next iteration, i.e. they fly under chunk ch's matrix-core work.
    constexpr int NROW = (CC * WR + 7) / 8;                       // window-row iterations per thread (8 rows per sweep)
    constexpr int XC = WC - 64, NXC = (CC * WR * XC + 255) / 256; // columns beyond 63 (1 for 3x3, 6 for 7x7): one element per item
    constexpr int NWG = (G::STEPS * 32 + 255) / 256;              // 16-byte weight pieces per thread
    s2_f2 pre_w[NROW];
    float pre_x[NXC];
    s2_f4 pre_g[NWG];
    const int l32 = lane & 31, sub = lane >> 5;
    auto fetch = [&](int ch) {
#pragma unroll
        for (int it = 0; it < NROW; ++it) {
            const int rr = it * 8 + 2 * w + sub;                   // window row (channel-major)
            s2_f2 v = {0.f, 0.f};
            if (rr < CC * WR) {
                const int c = rr / WR, r = rr - c * WR;
                const int gc = ch * CC + c, gy = iy0 + r, gx = ix0 + 2 * l32;
                if (gc < C && gy < Hi) {
                    const float* src = xn + gc * xi.cs + gy * xi.rs + gx;
                    if (pair_ok && gx + 1 < Wi) v = *reinterpret_cast<const s2_f2*>(src);
                    else {
                        if (gx < Wi) v[0] = src[0];
                        if (gx + 1 < Wi) v[1] = src[1];
                    }
                }
            }
            pre_w[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NXC; ++it) {
            const int i = it * 256 + threadIdx.x;
            float v = 0.f;
            if (i < CC * WR * XC) {
                const int rr = i / XC, col = 64 + (i - rr * XC);
                const int c = rr / WR, r = rr - c * WR;
                const int gc = ch * CC + c, gy = iy0 + r, gx = ix0 + col;
                if (gc < C && gy < Hi && gx < Wi) v = xn[gc * xi.cs + gy * xi.rs + gx];
            }
            pre_x[it] = v;
        }
        const s2_f4* src = reinterpret_cast<const s2_f4*>(wsrc + (size_t)ch * (G::STEPS * 128));
#pragma unroll
        for (int it = 0; it < NWG; ++it) {
            const int i = it * 256 + threadIdx.x;
            pre_g[it] = i < G::STEPS * 32 ? src[i] : s2_f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // window layout: columns split by parity -- element 2 l of a row goes to plane 0, 2 l + 1 to plane 1, both at index l
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < NROW; ++it) {
            const int rr = it * 8 + 2 * w + sub;
            if (rr < CC * WR) {
                const int c = rr / WR, r = rr - c * WR;
                float* dst = win + c * CHF + r * ROWF + l32;
                dst[0] = pre_w[it][0];
                dst[S2_PW] = pre_w[it][1];
            }
        }
#pragma unroll
        for (int it = 0; it < NXC; ++it) {
            const int i = it * 256 + threadIdx.x;
            if (i < CC * WR * XC) {
                const int rr = i / XC, col = 64 + (i - rr * XC);
                const int c = rr / WR, r = rr - c * WR;
                win[c * CHF + r * ROWF + (col & 1) * S2_PW + (col >> 1)] = pre_x[it];
            }
        }
        s2_f4* dst = reinterpret_cast<s2_f4*>(wgt);
#pragma unroll
        for (int it = 0; it < NWG; ++it) {
            const int i = it * 256 + threadIdx.x;
            if (i < G::STEPS * 32) dst[i] = pre_g[it];
        }
    };
    fetch(0);
    for (int ch = 0; ch < chunks; ++ch) {
        __syncthreads();        // the previous chunk's operands are consumed
        commit();
        __syncthreads();
        if (ch + 1 < chunks) fetch(ch + 1);
        // ---- k-steps, fully unrolled: every LDS offset below is an immediate
        auto step = [&](int s, int imm) {
            const float a0 = wgt[(2 * s) * 64 + lane], a1 = wgt[(2 * s + 1) * 64 + lane];
#pragma unroll
            for (int b = 0; b < RPW; ++b) {
                const float v = win[bbase + imm + 2 * b * ROWF];
                acc[b][0] = S2_MFMA(a0, v, acc[b][0]);
                acc[b][1] = S2_MFMA(a1, v, acc[b][1]);
            }
        };
        if constexpr (G::PAIR_B) {
            int s = 0;
#pragma unroll
            for (int c = 0; c < CC; ++c)
#pragma unroll
                for (int a = 0; a < KS; ++a)
#pragma unroll
                    for (int q = 0; q < (KS + 1) / 2; ++q, ++s) step(s, c * CHF + a * ROWF + q);
        } else {
            int s = 0;
#pragma unroll
            for (int p = 0; p < CC / 2; ++p)
#pragma unroll
                for (int a = 0; a < KS; ++a)
#pragma unroll
                    for (int b = 0; b < KS; ++b, ++s) step(s, 2 * p * CHF + a * ROWF + (b & 1) * S2_PW + (b >> 1));
        }
    }
    // ---- store: register r of lane (j, h) = channel 64 grp + 32 t + n(r, h), pixel (row, ox0 + j)
    float* yn = y + n * yo.ns;
    const int ox = ox0 + j;
#pragma unroll
    for (int b = 0; b < RPW; ++b) {
        const int oy = oy0 + RPW * w + b;
        if (oy < Ho && ox < Wo) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = 64 * grp + 32 * t + s2_nidx(r, h);
                    if (k < K) yn[k * yo.cs + oy * yo.rs + ox] = acc[b][t][r];
                }
        }
    }
}

template <int KS, int RPW>
static int s2_launch_fwd(const float* rec, const float* x, S2Tensor xi, int Hi, int Wi, float* y, S2Tensor yo, int Ho, int Wo, int n_img,
                         int C, int K, hipStream_t st) {
    using L = S2FwdLds<KS, RPW>;
    const int tiles_x = (Wo + 31) / 32, tiles_y = (Ho + 4 * RPW - 1) / (4 * RPW), groups = (K + 63) / 64;
    constexpr size_t smem = sizeof(float) * L::FLOATS;
    static bool once_on[NF_MAX_DEVICES] = {};
    bool& once = once_on[nf_current_device()];
    if (!once) {
        if (smem > 64 * 1024 &&
            hipFuncSetAttribute((const void*)k_conv_s2_fwd<KS, RPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
            nf_set_error("nf_conv_s2_fwd: cannot reserve %zu bytes of LDS", smem);
            return 1;
        }
        once = true;
    }
    hipLaunchKernelGGL((k_conv_s2_fwd<KS, RPW>), dim3((unsigned)(tiles_x * tiles_y * n_img * groups)), dim3(256), smem, st, rec, x, xi, Hi, Wi,
                       y, yo, Ho, Wo, C, K, groups, tiles_x, tiles_y);
    return 0;
}

/* y[n,k,oy,ox] = sum_{c,a,b} W[k][c][a][b] x[n,c,2 oy + a, 2 ox + b], Ho = (Hi - ks) / 2 + 1 (the network's stride-2 convolutions
 * on pre-padded activations).  records: nf_conv_s2_pack(..., backward = 0) on the device.  x, y: element strides (image, channel,
 * row), unit column stride. */
extern "C" int nf_conv_s2_fwd(const float* records, int ks, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h, int Hi, int Wi,
                              float* y, int64_t ys_n, int64_t ys_c, int64_t ys_h, int Ho, int Wo, int n_img, int c_in, int c_out,
                              nf_stream_t stream) {
    NF_REQUIRE((ks == 3 || ks == 7) && n_img >= 1 && c_in >= 1 && c_out >= 1 && Hi >= ks && Wi >= ks, "nf_conv_s2_fwd: bad arguments (ks %d)", ks);
    NF_REQUIRE(Ho == (Hi - ks) / 2 + 1 && Wo == (Wi - ks) / 2 + 1, "nf_conv_s2_fwd: output %d x %d does not match input %d x %d", Ho, Wo, Hi, Wi);
    NF_REQUIRE(ks == 3 || c_in <= 3, "nf_conv_s2_fwd: the 7x7 form takes at most 3 input channels (got %d)", c_in);
    hipStream_t st = (hipStream_t)stream;
    const S2Tensor xi{xs_n, xs_c, xs_h}, yo{ys_n, ys_c, ys_h};
    // small planes: 4-row tiles, so that the grid still covers the chip
    const int64_t wg8 = (int64_t)((Wo + 31) / 32) * ((Ho + 7) / 8) * n_img * ((c_out + 63) / 64);
    int rc;
    if (ks == 7) rc = s2_launch_fwd<7, 2>(records, x, xi, Hi, Wi, y, yo, Ho, Wo, n_img, c_in, c_out, st);
    else if (wg8 >= 512) rc = s2_launch_fwd<3, 2>(records, x, xi, Hi, Wi, y, yo, Ho, Wo, n_img, c_in, c_out, st);
    else rc = s2_launch_fwd<3, 1>(records, x, xi, Hi, Wi, y, yo, Ho, Wo, n_img, c_in, c_out, st);
    if (rc) return rc;
    NF_LAUNCH_CHECK("nf_conv_s2_fwd");
    return 0;
}

// ---- backward-data ---------------------------------------------------------------------------------------------------
// dx position (Y, X) = (2 u + ya, 2 v + xb); class (ya, xb) sums taps a = 2 i + ya, b = 2 j + xb of dy[u - i][v - j].
// Workgroup: 4 rows of u (one per wave) x 32 columns of v x 32 input channels c, all four classes.
template <int KS> struct S2Bwd {
    static constexpr int CK = 16;                               // dy channels per chunk
    static constexpr int T0 = (KS + 1) / 2, T1 = KS / 2;        // taps of the even / odd class along one axis
    static constexpr int WR = 4 + T0 - 1, WCOLS = 32 + T0 - 1;  // dy window rows / columns (rows u0 - (T0 - 1) .. u0 + 3)
    static constexpr int CHF = WR * S2_PW;
    static constexpr int WIN = CK * CHF;
    static constexpr int STEPS = (CK / 2) * KS * KS;
    static constexpr int FLOATS = WIN + STEPS * 64;
};

// ---- software-pipelined staging of the backward kernels: the dy window of CK channels (WR rows x WCOLS columns, zeros outside
// the tensor) and WFLOATS floats of weight records travel global -> registers (fetch, issued under the previous chunk's
// matrix-core work) -> LDS (commit, behind the barrier that retires the previous chunk)
template <int CK, int WR, int WCOLS, int T0, int WFLOATS>
struct S2BwdStage {
    static constexpr int NROW = (CK * WR + 7) / 8, XC = WCOLS - 32, NXC = (CK * WR * XC + 255) / 256, NWG = (WFLOATS / 4 + 255) / 256;
    static constexpr int CHF = WR * S2_PW;
    float pre_w[NROW], pre_x[NXC];
    s2_f4 pre_g[NWG];
    __device__ __forceinline__ void fetch(const float* __restrict__ dn, S2Tensor di, int Ho, int Wo, int K, int ch, int u0, int v0,
                                          const float* __restrict__ wsrc, int w, int lane) {
        const int l = lane & 31, sub = lane >> 5;
#pragma unroll
        for (int it = 0; it < NROW; ++it) {
            const int rr = it * 8 + 2 * w + sub;
            float v = 0.f;
            if (rr < CK * WR) {
                const int c = rr / WR, r = rr - c * WR;
                const int gk = ch * CK + c, gu = u0 - (T0 - 1) + r, gv = v0 - (T0 - 1) + l;
                if (gk < K && gu >= 0 && gu < Ho && gv >= 0 && gv < Wo) v = dn[gk * di.cs + gu * di.rs + gv];
            }
            pre_w[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NXC; ++it) {
            const int i = it * 256 + (int)threadIdx.x;
            float v = 0.f;
            if (i < CK * WR * XC) {
                const int rr = i / XC, cc = 32 + (i - rr * XC);
                const int c = rr / WR, r = rr - c * WR;
                const int gk = ch * CK + c, gu = u0 - (T0 - 1) + r, gv = v0 - (T0 - 1) + cc;
                if (gk < K && gu >= 0 && gu < Ho && gv >= 0 && gv < Wo) v = dn[gk * di.cs + gu * di.rs + gv];
            }
            pre_x[it] = v;
        }
        const s2_f4* src = reinterpret_cast<const s2_f4*>(wsrc + (size_t)ch * WFLOATS);
#pragma unroll
        for (int it = 0; it < NWG; ++it) {
            const int i = it * 256 + (int)threadIdx.x;
            pre_g[it] = i < WFLOATS / 4 ? src[i] : s2_f4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __device__ __forceinline__ void commit(float* win, float* wgt, int w, int lane) const {
        const int l = lane & 31, sub = lane >> 5;
#pragma unroll
        for (int it = 0; it < NROW; ++it) {
            const int rr = it * 8 + 2 * w + sub;
            if (rr < CK * WR) {
                const int c = rr / WR, r = rr - c * WR;
                win[c * CHF + r * S2_PW + l] = pre_w[it];
            }
        }
#pragma unroll
        for (int it = 0; it < NXC; ++it) {
            const int i = it * 256 + (int)threadIdx.x;
            if (i < CK * WR * XC) {
                const int rr = i / XC, cc = 32 + (i - rr * XC);
                const int c = rr / WR, r = rr - c * WR;
                win[c * CHF + r * S2_PW + cc] = pre_x[it];
            }
        }
        s2_f4* dst = reinterpret_cast<s2_f4*>(wgt);
#pragma unroll
        for (int it = 0; it < NWG; ++it) {
            const int i = it * 256 + (int)threadIdx.x;
            if (i < WFLOATS / 4) dst[i] = pre_g[it];
        }
    }
};

template <int KS>
__global__ void __launch_bounds__(256, 2) k_conv_s2_bwd(const float* __restrict__ rec, const float* __restrict__ dy, S2Tensor di, int Ho, int Wo,
                                                        float* __restrict__ dx, S2Tensor xo, int Hi, int Wi, int C, int K, int groups,
                                                        int tiles_x, int tiles_y) {
    using G = S2Bwd<KS>;
    constexpr int CK = G::CK, WR = G::WR, CHF = G::CHF, T0 = G::T0;
    HIP_DYNAMIC_SHARED(float, smem)
    float* win = smem;
    float* wgt = smem + G::WIN;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    int bid = blockIdx.x;
    const int grp = bid % groups;
    bid /= groups;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y, n = bid / tiles_y;
    const int u0 = ty * 4, v0 = tx * 32;
    const int chunks = (K + CK - 1) / CK;
    const float* dn = dy + n * di.ns;
    const float* wsrc = rec + (size_t)grp * chunks * (G::STEPS * 64);
    // window element (r, col) = dy[u0 - (T0 - 1) + r][v0 - (T0 - 1) + col]; this lane's (u, v) sits at (T0 - 1 + w, T0 - 1 + j)
    const int bbase = h * CHF + (T0 - 1 + w) * S2_PW + (T0 - 1) + j;

    s2_f16 acc[2][2];         // [ya][xb]
#pragma unroll
    for (int ya = 0; ya < 2; ++ya)
#pragma unroll
        for (int xb = 0; xb < 2; ++xb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ya][xb][r] = 0.f;

    S2BwdStage<CK, WR, G::WCOLS, T0, G::STEPS * 64> stage;
    stage.fetch(dn, di, Ho, Wo, K, 0, u0, v0, wsrc, w, lane);
    for (int ch = 0; ch < chunks; ++ch) {
        __syncthreads();
        stage.commit(win, wgt, w, lane);
        __syncthreads();
        if (ch + 1 < chunks) stage.fetch(dn, di, Ho, Wo, K, ch + 1, u0, v0, wsrc, w, lane);
        int s = 0;
#pragma unroll
        for (int ya = 0; ya < 2; ++ya)
#pragma unroll
            for (int xb = 0; xb < 2; ++xb)
#pragma unroll
                for (int i = 0; 2 * i + ya < KS; ++i)
#pragma unroll
                    for (int jj = 0; 2 * jj + xb < KS; ++jj)
#pragma unroll
                        for (int p = 0; p < CK / 2; ++p, ++s) {
                            const float a = wgt[s * 64 + lane];
                            const float v = win[bbase + 2 * p * CHF - i * S2_PW - jj];
                            acc[ya][xb] = S2_MFMA(a, v, acc[ya][xb]);
                        }
    }
    // ---- store: the two column classes of a lane are adjacent in memory; everything of dx is written (zeros included)
    float* xn = dx + n * xo.ns;
    const int u = u0 + w, v = v0 + j;
#pragma unroll
    for (int ya = 0; ya < 2; ++ya) {
        const int Y = 2 * u + ya, X = 2 * v;
        if (Y < Hi && X < Wi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * grp + s2_nidx(r, h);
                if (c < C) {
                    float* p = xn + c * xo.cs + Y * xo.rs + X;
                    if (X + 1 < Wi) *reinterpret_cast<s2_f2*>(p) = s2_f2{acc[ya][0][r], acc[ya][1][r]};
                    else p[0] = acc[ya][0][r];
                }
            }
        }
    }
}

// ---- backward-data of the stride-2 3x3 convolutions on the BF16 matrix cores, operands as three bf16 parts ("bf16x3", as
// csrc/nf_wino_bf.hip: the six cross products of order <= 2^-16 on v_mfma_f32_32x32x16_bf16 reproduce the fp32 product to fp32 rounding
// level at 6 / 16 of the fp32 matrix instructions' time, and beside the vector pipe).  Same workgroup geometry as k_conv_s2_bwd<3>
// (4 rows of u x 32 columns of v x 32 input channels, all four parity classes), a chunk = 16 dy channels = ONE k-block of the bf16
// instruction: lane (v, h) supplies the channels 2 j + h, j = 0..7, of its dy position.  The nine (class, tap) products of a chunk read
// only FOUR distinct dy positions (u - i, v - jj), i, jj in {0, 1}: each is gathered and split once and multiplied with the weight
// parts of every class that uses it.  Records: [group of 32 inputs][chunk][position (i, jj)][class using it][part 3][lane][8 bf16].
typedef unsigned s2_u4 __attribute__((ext_vector_type(4)));
typedef __bf16 s2_bf8 __attribute__((ext_vector_type(8)));
#define S2_X3_COMBOS 9
#define S2_X3_CHUNK_FLOATS (S2_X3_COMBOS * 3 * 256)

static inline uint16_t s2_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

#define S2_X3_FWD_TP 5                                       // tap pairs of the 3 x 3 kernel (the tenth tap has zero weights)
#define S2_X3_FWD_CHUNK_FLOATS (S2_X3_FWD_TP * 2 * 3 * 256)  // forward: [tap pair][channel tile 2][part 3][lane][8 bf16]

extern "C" int64_t nf_conv_s2_x3_pack_floats(int c_out, int c_in, int backward) {
    if (backward) return (int64_t)((c_in + 31) / 32) * ((c_out + 15) / 16) * S2_X3_CHUNK_FLOATS;
    return (int64_t)((c_out + 63) / 64) * ((c_in + 7) / 8) * S2_X3_FWD_CHUNK_FLOATS;
}

/* HOST: weight [c_out][c_in][3][3] -> bf16x3 records of the backward-data pass (see above); lane (c = 32 g + (lane & 31), h = lane >> 5),
 * element j <-> dy channel k = 16 chunk + 2 j + h; tap a = 2 i + ya, b = 2 jj + xb */
static void s2_x3_emit(uint16_t* piece, int lane, int j, float v) {
    float rem = v;
    for (int p = 0; p < 3; ++p) {
        const uint16_t q = s2_rne(rem);
        piece[(size_t)p * 512 + lane * 8 + j] = q;
        const uint32_t u = (uint32_t)q << 16;
        float up;
        memcpy(&up, &u, 4);
        rem -= up;
    }
}

/* forward records (backward = 0): [group of 64 outputs][chunk of 8 inputs][tap pair tp][tile t][part][lane (i, h)][8 bf16], element j
 * <-> input channel 8 chunk + 2 (j >> 1) + h and tap 2 tp + (j & 1) (tap = 3 a + b; tap 9: zero) of output 64 g + 32 t + i */
extern "C" int nf_conv_s2_x3_pack(const float* weight, int c_out, int c_in, int backward, float* out) {
    if (!backward) {
        const int groups = (c_out + 63) / 64, chunks = (c_in + 7) / 8;
        uint16_t* piece = reinterpret_cast<uint16_t*>(out);
        for (int g = 0; g < groups; ++g)
            for (int ch = 0; ch < chunks; ++ch)
                for (int tp = 0; tp < S2_X3_FWD_TP; ++tp)
                    for (int t = 0; t < 2; ++t, piece += 3 * 512)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int k = 64 * g + 32 * t + (lane & 31), c = 8 * ch + 2 * (j >> 1) + (lane >> 5), tap = 2 * tp + (j & 1);
                                const float v = (k < c_out && c < c_in && tap < 9) ? weight[((size_t)k * c_in + c) * 9 + tap] : 0.f;
                                s2_x3_emit(piece, lane, j, v);
                            }
        return (reinterpret_cast<float*>(piece) - out) == nf_conv_s2_x3_pack_floats(c_out, c_in, 0) ? 0 : 2;
    }
    const int groups = (c_in + 31) / 32, chunks = (c_out + 15) / 16;
    uint16_t* piece = reinterpret_cast<uint16_t*>(out);
    for (int g = 0; g < groups; ++g)
        for (int ch = 0; ch < chunks; ++ch)
            for (int i = 0; i < 2; ++i)
                for (int jj = 0; jj < 2; ++jj)
                    for (int ya = 0; 2 * i + ya < 3 && ya < 2; ++ya)
                        for (int xb = 0; 2 * jj + xb < 3 && xb < 2; ++xb, piece += 3 * 512)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int j = 0; j < 8; ++j) {
                                    const int c = 32 * g + (lane & 31), k = 16 * ch + 2 * j + (lane >> 5), a = 2 * i + ya, b = 2 * jj + xb;
                                    float rem = (k < c_out && c < c_in) ? weight[(((size_t)k * c_in + c) * 3 + a) * 3 + b] : 0.f;
                                    for (int p = 0; p < 3; ++p) {
                                        const uint16_t q = s2_rne(rem);
                                        piece[(size_t)p * 512 + lane * 8 + j] = q;
                                        const uint32_t u = (uint32_t)q << 16;
                                        float up;
                                        memcpy(&up, &u, 4);
                                        rem -= up;
                                    }
                                }
    return (reinterpret_cast<float*>(piece) - out) == nf_conv_s2_x3_pack_floats(c_out, c_in, 1) ? 0 : 2;
}

__device__ __forceinline__ unsigned s2_split_pair(float& x0, float& x1) { return nf_split_pair_bf16(x0, x1); }

__global__ void __launch_bounds__(256, 2) k_conv_s2_bwd3_x3(const float* __restrict__ rec, const float* __restrict__ dy, S2Tensor di, int Ho, int Wo,
                                                            float* __restrict__ dx, S2Tensor xo, int Hi, int Wi, int C, int K, int groups,
                                                            int tiles_x, int tiles_y) {
    using G = S2Bwd<3>;
    constexpr int CK = G::CK, WR = G::WR, CHF = G::CHF, T0 = G::T0;
    static_assert(CK == 16 && T0 == 2, "one bf16 k-block per chunk, taps i, jj in {0, 1}");
    HIP_DYNAMIC_SHARED(float, smem)
    float* win = smem;
    float* wgt = smem + G::WIN;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    int bid = blockIdx.x;
    const int grp = bid % groups;
    bid /= groups;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y, n = bid / tiles_y;
    const int u0 = ty * 4, v0 = tx * 32;
    const int chunks = (K + CK - 1) / CK;
    const float* dn = dy + n * di.ns;
    const float* wsrc = rec + (size_t)grp * chunks * S2_X3_CHUNK_FLOATS;
    const int bbase = h * CHF + (T0 - 1 + w) * S2_PW + (T0 - 1) + j;

    s2_f16 acc[2][2];         // [ya][xb]
#pragma unroll
    for (int ya = 0; ya < 2; ++ya)
#pragma unroll
        for (int xb = 0; xb < 2; ++xb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ya][xb][r] = 0.f;

    S2BwdStage<CK, WR, G::WCOLS, T0, S2_X3_CHUNK_FLOATS> stage;
    stage.fetch(dn, di, Ho, Wo, K, 0, u0, v0, wsrc, w, lane);
    for (int ch = 0; ch < chunks; ++ch) {
        __syncthreads();
        stage.commit(win, wgt, w, lane);
        __syncthreads();
        if (ch + 1 < chunks) stage.fetch(dn, di, Ho, Wo, K, ch + 1, u0, v0, wsrc, w, lane);
        int combo = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                // the lane's 8 channels (2 p + h) of dy[u - i][v - jj], split into bf16 parts
                s2_u4 bv[3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v0f = win[bbase + 2 * (2 * q) * CHF - i * S2_PW - jj], v1f = win[bbase + 2 * (2 * q + 1) * CHF - i * S2_PW - jj];
#pragma unroll
                    for (int p = 0; p < 3; ++p) bv[p][q] = s2_split_pair(v0f, v1f);
                }
#pragma unroll
                for (int ya = 0; ya < 2; ++ya)
#pragma unroll
                    for (int xb = 0; xb < 2; ++xb) {
                        if (2 * i + ya >= 3 || 2 * jj + xb >= 3) continue;
                        const float* rs = wgt + combo * (3 * 256) + 4 * lane;
                        const s2_u4 a0 = *reinterpret_cast<const s2_u4*>(rs), a1 = *reinterpret_cast<const s2_u4*>(rs + 256),
                                    a2 = *reinterpret_cast<const s2_u4*>(rs + 512);
#define S2_PROD(a, b) acc[ya][xb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s2_bf8, a), __builtin_bit_cast(s2_bf8, b), acc[ya][xb], 0, 0, 0)
                        S2_PROD(a0, bv[0]);
                        S2_PROD(a0, bv[1]);
                        S2_PROD(a1, bv[0]);
                        S2_PROD(a0, bv[2]);
                        S2_PROD(a2, bv[0]);
                        S2_PROD(a1, bv[1]);
#undef S2_PROD
                        ++combo;
                    }
            }
    }
    // ---- store: as k_conv_s2_bwd
    float* xn = dx + n * xo.ns;
    const int u = u0 + w, v = v0 + j;
#pragma unroll
    for (int ya = 0; ya < 2; ++ya) {
        const int Y = 2 * u + ya, X = 2 * v;
        if (Y < Hi && X < Wi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * grp + s2_nidx(r, h);
                if (c < C) {
                    float* p = xn + c * xo.cs + Y * xo.rs + X;
                    if (X + 1 < Wi) *reinterpret_cast<s2_f2*>(p) = s2_f2{acc[ya][0][r], acc[ya][1][r]};
                    else p[0] = acc[ya][0][r];
                }
            }
        }
    }
}

// ---- forward of the stride-2 3x3 convolutions on bf16x3 operands.  Geometry, window staging (de-interleaved by column parity) and
// output stage of k_conv_s2_fwd<3, RPW>; a chunk is 8 input channels, a k-block of the bf16 instruction = one PAIR of taps x the 4
// channels of the lane half's parity (five blocks per chunk, the tenth tap slot carries zero weights): per block and output row a
// lane gathers its 8 window values (immediate LDS offsets), splits them once and multiplies with the parts of both channel tiles.
#if defined(__HIP_DEVICE_COMPILE__)
#define S2_OPAQUE_S(x) asm volatile("" : "+s"(x))
#else
#define S2_OPAQUE_S(x) asm volatile("" : "+r"(x))
#endif
template <int RPW>
__global__ void __launch_bounds__(256, 2) k_conv_s2_fwd3_x3(const float* __restrict__ rec, const float* __restrict__ x, S2Tensor xi, int Hi, int Wi,
                                                            float* __restrict__ y, S2Tensor yo, int Ho, int Wo, int C, int K, int groups,
                                                            int tiles_x, int tiles_y) {
    using G = S2Fwd<3>;
    using L = S2FwdLds<3, RPW>;
    constexpr int CC = G::CC, WR = L::WR, WC = G::WC, ROWF = 2 * S2_PW, CHF = WR * ROWF, WFL = S2_X3_FWD_CHUNK_FLOATS;
    static_assert(CC == 8, "four channel pairs per k-block");
    HIP_DYNAMIC_SHARED(float, smem)
    float* win = smem;
    float* wgt = smem + L::WIN;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    int bid = blockIdx.x;
    const int grp = bid % groups;
    bid /= groups;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y, n = bid / tiles_y;
    const int oy0 = ty * (4 * RPW), ox0 = tx * 32;
    const int iy0 = 2 * oy0, ix0 = 2 * ox0;
    const int chunks = (C + CC - 1) / CC;
    const float* xn = x + n * xi.ns;
    const float* wsrc = rec + (size_t)grp * chunks * WFL;
    const int bbase = h * CHF + (2 * RPW * w) * ROWF + j;
    const bool pair_ok = ((xi.ns | xi.cs | xi.rs) & 1) == 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0;

    s2_f16 acc[RPW][2];
#pragma unroll
    for (int b = 0; b < RPW; ++b)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][t][r] = 0.f;

    constexpr int NROW = (CC * WR + 7) / 8;
    constexpr int XC = WC - 64, NXC = (CC * WR * XC + 255) / 256;
    constexpr int NWG = (WFL / 4 + 255) / 256;                    // 16-byte weight pieces per thread
    s2_f2 pre_w[NROW];
    float pre_x[NXC];
    s2_f4 pre_g[NWG];
    const int l32 = lane & 31, sub = lane >> 5;
    auto fetch = [&](int ch) {
        S2_OPAQUE_S(ch);      // addresses from the chunk index every time: no per-piece pointer kept across the chunk loop
#pragma unroll
        for (int it = 0; it < NROW; ++it) {
            const int rr = it * 8 + 2 * w + sub;
            s2_f2 v = {0.f, 0.f};
            if (rr < CC * WR) {
                const int c = rr / WR, r = rr - c * WR;
                const int gc = ch * CC + c, gy = iy0 + r, gx = ix0 + 2 * l32;
                if (gc < C && gy < Hi) {
                    const float* src = xn + gc * xi.cs + gy * xi.rs + gx;
                    if (pair_ok && gx + 1 < Wi) v = *reinterpret_cast<const s2_f2*>(src);
                    else {
                        if (gx < Wi) v[0] = src[0];
                        if (gx + 1 < Wi) v[1] = src[1];
                    }
                }
            }
            pre_w[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NXC; ++it) {
            const int i = it * 256 + threadIdx.x;
            float v = 0.f;
            if (i < CC * WR * XC) {
                const int rr = i / XC, col = 64 + (i - rr * XC);
                const int c = rr / WR, r = rr - c * WR;
                const int gc = ch * CC + c, gy = iy0 + r, gx = ix0 + col;
                if (gc < C && gy < Hi && gx < Wi) v = xn[gc * xi.cs + gy * xi.rs + gx];
            }
            pre_x[it] = v;
        }
        const s2_f4* src = reinterpret_cast<const s2_f4*>(wsrc + (size_t)ch * WFL);
#pragma unroll
        for (int it = 0; it < NWG; ++it) {
            const int i = it * 256 + threadIdx.x;
            pre_g[it] = i < WFL / 4 ? src[i] : s2_f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < NROW; ++it) {
            const int rr = it * 8 + 2 * w + sub;
            if (rr < CC * WR) {
                const int c = rr / WR, r = rr - c * WR;
                float* dst = win + c * CHF + r * ROWF + l32;
                dst[0] = pre_w[it][0];
                dst[S2_PW] = pre_w[it][1];
            }
        }
#pragma unroll
        for (int it = 0; it < NXC; ++it) {
            const int i = it * 256 + threadIdx.x;
            if (i < CC * WR * XC) {
                const int rr = i / XC, col = 64 + (i - rr * XC);
                const int c = rr / WR, r = rr - c * WR;
                win[c * CHF + r * ROWF + (col & 1) * S2_PW + (col >> 1)] = pre_x[it];
            }
        }
        s2_f4* dst = reinterpret_cast<s2_f4*>(wgt);
#pragma unroll
        for (int it = 0; it < NWG; ++it) {
            const int i = it * 256 + threadIdx.x;
            if (i < WFL / 4) dst[i] = pre_g[it];
        }
    };
    fetch(0);
    for (int ch = 0; ch < chunks; ++ch) {
        __syncthreads();
        commit();
        __syncthreads();
        if (ch + 1 < chunks) fetch(ch + 1);
#pragma unroll
        for (int tp = 0; tp < S2_X3_FWD_TP; ++tp) {
            // taps 2 tp and 2 tp + 1 (tap = 3 a + bb; the tenth does not exist: zero weights, value 0)
            constexpr int dummy = 0;
            (void)dummy;
            const int t0 = 2 * tp, t1 = 2 * tp + 1;
            const int ta0 = t0 / 3, tb0 = t0 - 3 * ta0, ta1 = t1 / 3, tb1 = t1 - 3 * ta1;
            s2_u4 av[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 3; ++q) av[t][q] = *reinterpret_cast<const s2_u4*>(wgt + ((tp * 2 + t) * 3 + q) * 256 + 4 * lane);
#pragma unroll
            for (int b = 0; b < RPW; ++b) {
                s2_u4 bv[3];
#pragma unroll
                for (int p = 0; p < 4; ++p) {      // channel 2 p + h
                    float v0 = win[bbase + 2 * p * CHF + ta0 * ROWF + (tb0 & 1) * S2_PW + (tb0 >> 1) + 2 * b * ROWF];
                    float v1 = t1 < 9 ? win[bbase + 2 * p * CHF + ta1 * ROWF + (tb1 & 1) * S2_PW + (tb1 >> 1) + 2 * b * ROWF] : 0.f;
#pragma unroll
                    for (int q = 0; q < 3; ++q) bv[q][p] = s2_split_pair(v0, v1);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#define S2_PROD(qa, qb) acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s2_bf8, av[t][qa]), __builtin_bit_cast(s2_bf8, bv[qb]), acc[b][t], 0, 0, 0)
                    S2_PROD(0, 0);
                    S2_PROD(0, 1);
                    S2_PROD(1, 0);
                    S2_PROD(0, 2);
                    S2_PROD(2, 0);
                    S2_PROD(1, 1);
#undef S2_PROD
                }
            }
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    }
    float* yn = y + n * yo.ns;
    const int ox = ox0 + j;
#pragma unroll
    for (int b = 0; b < RPW; ++b) {
        const int oy = oy0 + RPW * w + b;
        if (oy < Ho && ox < Wo) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = 64 * grp + 32 * t + s2_nidx(r, h);
                    if (k < K) yn[k * yo.cs + oy * yo.rs + ox] = acc[b][t][r];
                }
        }
    }
}

template <int RPW>
static int s2_launch_fwd3_x3(const float* rec, const float* x, S2Tensor xi, int Hi, int Wi, float* y, S2Tensor yo, int Ho, int Wo, int n_img,
                             int C, int K, hipStream_t st) {
    const int tiles_x = (Wo + 31) / 32, tiles_y = (Ho + 4 * RPW - 1) / (4 * RPW), groups = (K + 63) / 64;
    constexpr size_t smem = sizeof(float) * (S2FwdLds<3, RPW>::WIN + S2_X3_FWD_CHUNK_FLOATS);
    static bool once_on[NF_MAX_DEVICES] = {};
    bool& once = once_on[nf_current_device()];
    if (!once) {
        if (smem > 64 * 1024 &&
            hipFuncSetAttribute((const void*)k_conv_s2_fwd3_x3<RPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
            nf_set_error("nf_conv_s2_fwd_x3: cannot reserve %zu bytes of LDS", smem);
            return 1;
        }
        once = true;
    }
    hipLaunchKernelGGL((k_conv_s2_fwd3_x3<RPW>), dim3((unsigned)(tiles_x * tiles_y * n_img * groups)), dim3(256), smem, st, rec, x, xi, Hi, Wi, y,
                       yo, Ho, Wo, C, K, groups, tiles_x, tiles_y);
    return 0;
}

/* nf_conv_s2_fwd for ks = 3 on the bf16 matrix cores with three-way split operands (fp32-grade); records: nf_conv_s2_x3_pack(..., 0) */
extern "C" int nf_conv_s2_fwd_x3(const float* records, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h, int Hi, int Wi, float* y,
                                 int64_t ys_n, int64_t ys_c, int64_t ys_h, int Ho, int Wo, int n_img, int c_in, int c_out, nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && c_in >= 1 && c_out >= 1 && Hi >= 3 && Wi >= 3, "nf_conv_s2_fwd_x3: bad arguments");
    NF_REQUIRE(Ho == (Hi - 3) / 2 + 1 && Wo == (Wi - 3) / 2 + 1, "nf_conv_s2_fwd_x3: output %d x %d does not match input %d x %d", Ho, Wo, Hi, Wi);
    const S2Tensor xi{xs_n, xs_c, xs_h}, yo{ys_n, ys_c, ys_h};
    // one output row per wave (three waves per SIMD): measured faster than two rows (256 registers, spills) at every layer of config 2
    const int rc = s2_launch_fwd3_x3<1>(records, x, xi, Hi, Wi, y, yo, Ho, Wo, n_img, c_in, c_out, (hipStream_t)stream);
    if (rc) return rc;
    NF_LAUNCH_CHECK("nf_conv_s2_fwd_x3");
    return 0;
}

// ---- the 7x7 stem forward on bf16x3 operands.  K = 3 colours x 7 rows x 7 columns: a k-block of the bf16 instruction is one (tap row a,
// colour c) LINE of the kernel per lane half -- line r = 3 a + c = 2 kb + h, elements = the tap columns b = 0 .. 7 (b = 7: zero weights) --
// eleven blocks, the last half-block (r = 21) of zero weights.  The input window is staged as lines (3 y + c) de-interleaved by column
// parity, so that line r of output row o sits at line 6 o + r: lane base + compile-time immediate for every B read.  All 66 KB of split
// weights stay in LDS for the life of the workgroup, which walks a strip of 4 ITER output rows (window re-staged per 4 rows, prefetched
// into registers under the previous rows' products): the records come from L2 once per strip, not once per 4 rows.
#define S2_STEM_X3_KB 11
#define S2_STEM_X3_FLOATS (S2_STEM_X3_KB * 2 * 3 * 256)       // per group of 64 outputs: [kb][tile t][part][lane][8 bf16]

extern "C" int64_t nf_conv_s2_stem_x3_pack_floats(int c_out) { return (int64_t)((c_out + 63) / 64) * S2_STEM_X3_FLOATS; }

/* HOST: weight [c_out][c_in <= 3][7][7] -> [group of 64 outputs][kb][tile t][part][lane (i, h)][8 bf16]; element j of lane (i, h) =
 * W[64 g + 32 t + i][c][a][b = j], 3 a + c = 2 kb + h (zero for a > 6, b > 6, c >= c_in) */
extern "C" int nf_conv_s2_stem_x3_pack(const float* weight, int c_out, int c_in, float* out) {
    if (c_in < 1 || c_in > 3) return 1;
    const int groups = (c_out + 63) / 64;
    uint16_t* piece = reinterpret_cast<uint16_t*>(out);
    for (int g = 0; g < groups; ++g)
        for (int kb = 0; kb < S2_STEM_X3_KB; ++kb)
            for (int t = 0; t < 2; ++t, piece += 3 * 512)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int k = 64 * g + 32 * t + (lane & 31), r = 2 * kb + (lane >> 5), a = r / 3, c = r - 3 * a;
                        const float v = (k < c_out && a < 7 && j < 7 && c < c_in) ? weight[(((size_t)k * c_in + c) * 7 + a) * 7 + j] : 0.f;
                        s2_x3_emit(piece, lane, j, v);
                    }
    return (reinterpret_cast<float*>(piece) - out) == nf_conv_s2_stem_x3_pack_floats(c_out) ? 0 : 2;
}

// NW waves x RPW output rows each per iteration; the A parts of a k-block are read once for the wave's RPW rows
template <int NW, int RPW>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) k_conv_s2_stem_fwd_x3(const float* __restrict__ rec, const float* __restrict__ x, S2Tensor xi, int Hi, int Wi,
                                                                float* __restrict__ y, S2Tensor yo, int Ho, int Wo, int C, int K, int groups,
                                                                int tiles_x, int strips, int iters) {
    constexpr int ROWS = NW * RPW, NT = 64 * NW;                   // output rows per iteration, threads
    constexpr int ROWF = 2 * S2_PW, LINES = 3 * (2 * (ROWS - 1) + 7) + 1, WINF = LINES * ROWF, WC = 70, XC = WC - 64;      // (+ the line kept zero)
    constexpr int NROW = (LINES - 1 + 2 * NW - 1) / (2 * NW), NXC = ((LINES - 1) * XC + NT - 1) / NT;
    HIP_DYNAMIC_SHARED(float, smem)
    float* win = smem;
    float* wgt = smem + WINF;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    int bid = blockIdx.x;
    const int grp = bid % groups;
    bid /= groups;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int strip = bid % strips, n = bid / strips;
    const int ox0 = tx * 32, ix0 = 2 * ox0, oy_first = strip * ROWS * iters;
    const float* xn = x + n * xi.ns;
    const bool pair_ok = ((xi.ns | xi.cs | xi.rs) & 1) == 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0;
    const int bbase = (6 * RPW * w + h) * ROWF + j;
    const int l32 = lane & 31, sub = lane >> 5;

    // the split weights of this group of 64 outputs: 66 KB, once
    {
        const s2_f4* src = reinterpret_cast<const s2_f4*>(rec + (size_t)grp * S2_STEM_X3_FLOATS);
        s2_f4* dst = reinterpret_cast<s2_f4*>(wgt);
        for (int i = threadIdx.x; i < S2_STEM_X3_FLOATS / 4; i += NT) dst[i] = src[i];
        for (int i = threadIdx.x; i < ROWF; i += NT) win[(LINES - 1) * ROWF + i] = 0.f;       // the line only zero weights read
    }
    s2_f2 pre_w[NROW];
    float pre_x[NXC];
    auto fetch = [&](int it) {
        const int iy0 = 2 * (oy_first + ROWS * it);
        S2_OPAQUE_S(it);
#pragma unroll
        for (int q = 0; q < NROW; ++q) {
            const int L = q * (2 * NW) + 2 * w + sub;
            s2_f2 v = {0.f, 0.f};
            if (L < LINES - 1) {
                const int yl = L / 3, c = L - 3 * yl;
                const int gy = iy0 + yl, gx = ix0 + 2 * l32;
                if (c < C && gy < Hi) {
                    const float* src = xn + c * xi.cs + gy * xi.rs + gx;
                    if (pair_ok && gx + 1 < Wi) v = *reinterpret_cast<const s2_f2*>(src);
                    else {
                        if (gx < Wi) v[0] = src[0];
                        if (gx + 1 < Wi) v[1] = src[1];
                    }
                }
            }
            pre_w[q] = v;
        }
#pragma unroll
        for (int q = 0; q < NXC; ++q) {
            const int i = q * NT + threadIdx.x;
            float v = 0.f;
            if (i < (LINES - 1) * XC) {
                const int L = i / XC, col = 64 + (i - L * XC);
                const int yl = L / 3, c = L - 3 * yl;
                const int gy = iy0 + yl, gx = ix0 + col;
                if (c < C && gy < Hi && gx < Wi) v = xn[c * xi.cs + gy * xi.rs + gx];
            }
            pre_x[q] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < NROW; ++q) {
            const int L = q * (2 * NW) + 2 * w + sub;
            if (L < LINES - 1) {
                float* dst = win + L * ROWF + l32;
                dst[0] = pre_w[q][0];
                dst[S2_PW] = pre_w[q][1];
            }
        }
#pragma unroll
        for (int q = 0; q < NXC; ++q) {
            const int i = q * NT + threadIdx.x;
            if (i < (LINES - 1) * XC) {
                const int L = i / XC, col = 64 + (i - L * XC);
                win[L * ROWF + (col & 1) * S2_PW + (col >> 1)] = pre_x[q];
            }
        }
    };
    float* yn = y + n * yo.ns;
    const int ox = ox0 + j;
    fetch(0);
    for (int it = 0; it < iters; ++it) {
        if (oy_first + ROWS * it >= Ho) break;       // (uniform)
        __syncthreads();
        commit();
        __syncthreads();
        if (it + 1 < iters) fetch(it + 1);
        s2_f16 acc[RPW][2];
#pragma unroll
        for (int b = 0; b < RPW; ++b)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][t][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < S2_STEM_X3_KB; ++kb) {
            s2_u4 av[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 3; ++q) av[t][q] = *reinterpret_cast<const s2_u4*>(wgt + ((kb * 2 + t) * 3 + q) * 256 + 4 * lane);
#pragma unroll
            for (int b = 0; b < RPW; ++b) {
                s2_u4 bv[3];
#pragma unroll
                for (int p = 0; p < 4; ++p) {      // tap columns 2 p, 2 p + 1 (column 7: zero weights, value 0)
                    float v0 = win[bbase + (2 * kb + 6 * b) * ROWF + p];
                    float v1 = p < 3 ? win[bbase + (2 * kb + 6 * b) * ROWF + S2_PW + p] : 0.f;
#pragma unroll
                    for (int q = 0; q < 3; ++q) bv[q][p] = s2_split_pair(v0, v1);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#define S2_PROD(qa, qb) acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s2_bf8, av[t][qa]), __builtin_bit_cast(s2_bf8, bv[qb]), acc[b][t], 0, 0, 0)
                    S2_PROD(0, 0);
                    S2_PROD(0, 1);
                    S2_PROD(1, 0);
                    S2_PROD(0, 2);
                    S2_PROD(2, 0);
                    S2_PROD(1, 1);
#undef S2_PROD
                }
            }
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);      // a k-block's reads stay with its products (hoisted across blocks they cost 60+ registers)
#endif
        }
#pragma unroll
        for (int b = 0; b < RPW; ++b) {
            const int oy = oy_first + ROWS * it + RPW * w + b;
            if (oy < Ho && ox < Wo) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int k = 64 * grp + 32 * t + s2_nidx(r, h);
                        if (k < K) yn[k * yo.cs + oy * yo.rs + ox] = acc[b][t][r];
                    }
            }
        }
    }
}

#ifndef S2_STEM_X3_NW
#define S2_STEM_X3_NW 4
#define S2_STEM_X3_RPW 1
#endif
template <int NW, int RPW>
static int s2_launch_stem_x3(const float* records, const float* x, S2Tensor xi, int Hi, int Wi, float* y, S2Tensor yo, int Ho, int Wo, int n_img,
                             int c_in, int c_out, hipStream_t st) {
    constexpr int ROWS = NW * RPW, LINES = 3 * (2 * (ROWS - 1) + 7) + 1, PER_CU = NW == 4 ? 2 : 1;
    const int tiles_x = (Wo + 31) / 32, groups = (c_out + 63) / 64, row_groups = (Ho + ROWS - 1) / ROWS;
    // iterations per workgroup (a strip of ROWS * iters output rows): the choice with the fewest (rounds over the chip) x (iterations + the
    // ~1.5 iterations a workgroup spends on its 66 KB of records and its first window)
    int iters = 1;
    double best = 1e30;
    for (int it = 1; it <= 16 && it <= row_groups; ++it) {
        const int64_t wgs = (int64_t)tiles_x * ((row_groups + it - 1) / it) * n_img * groups;
        const double cost = (double)((wgs + 256 * PER_CU - 1) / (256 * PER_CU)) * (it + 1.5);
        if (cost < best - 1e-9) best = cost, iters = it;
    }
    const int strips = (row_groups + iters - 1) / iters;
    constexpr size_t smem = sizeof(float) * (LINES * 2 * S2_PW + S2_STEM_X3_FLOATS);
    static_assert(PER_CU * smem <= 160 * 1024, "workgroups per CU");
    static bool once_on[NF_MAX_DEVICES] = {};
    bool& once = once_on[nf_current_device()];
    if (!once) {
        if (hipFuncSetAttribute((const void*)k_conv_s2_stem_fwd_x3<NW, RPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
            nf_set_error("nf_conv_s2_stem_fwd_x3: cannot reserve %zu bytes of LDS", smem);
            return 1;
        }
        once = true;
    }
    hipLaunchKernelGGL((k_conv_s2_stem_fwd_x3<NW, RPW>), dim3((unsigned)(tiles_x * strips * n_img * groups)), dim3(64 * NW), smem, st, records, x, xi,
                       Hi, Wi, y, yo, Ho, Wo, c_in, c_out, groups, tiles_x, strips, iters);
    return 0;
}

/* nf_conv_s2_fwd for the 7x7 stem (c_in <= 3) on the bf16 matrix cores with three-way split operands (fp32-grade); records:
 * nf_conv_s2_stem_x3_pack */
extern "C" int nf_conv_s2_stem_fwd_x3(const float* records, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h, int Hi, int Wi, float* y,
                                      int64_t ys_n, int64_t ys_c, int64_t ys_h, int Ho, int Wo, int n_img, int c_in, int c_out,
                                      nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && c_in >= 1 && c_in <= 3 && c_out >= 1 && Hi >= 7 && Wi >= 7, "nf_conv_s2_stem_fwd_x3: bad arguments (c_in %d)", c_in);
    NF_REQUIRE(Ho == (Hi - 7) / 2 + 1 && Wo == (Wi - 7) / 2 + 1, "nf_conv_s2_stem_fwd_x3: output %d x %d does not match input %d x %d", Ho, Wo, Hi,
               Wi);
    const S2Tensor xi{xs_n, xs_c, xs_h}, yo{ys_n, ys_c, ys_h};
    const int rc = s2_launch_stem_x3<S2_STEM_X3_NW, S2_STEM_X3_RPW>(records, x, xi, Hi, Wi, y, yo, Ho, Wo, n_img, c_in, c_out, (hipStream_t)stream);
    if (rc) return rc;
    NF_LAUNCH_CHECK("nf_conv_s2_stem_fwd_x3");
    return 0;
}

/* nf_conv_s2_bwd for ks = 3 on the bf16 matrix cores with three-way split operands (fp32-grade); records: nf_conv_s2_x3_pack */
extern "C" int nf_conv_s2_bwd_x3(const float* records, const float* dy, int64_t ds_n, int64_t ds_c, int64_t ds_h, int Ho, int Wo, float* dx,
                                 int64_t xs_n, int64_t xs_c, int64_t xs_h, int Hi, int Wi, int n_img, int c_in, int c_out, nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && c_in >= 1 && c_out >= 1 && Hi >= 3 && Wi >= 3, "nf_conv_s2_bwd_x3: bad arguments");
    NF_REQUIRE(Ho == (Hi - 3) / 2 + 1 && Wo == (Wi - 3) / 2 + 1, "nf_conv_s2_bwd_x3: gradient %d x %d does not match input %d x %d", Ho, Wo, Hi, Wi);
    const S2Tensor di{ds_n, ds_c, ds_h}, xo{xs_n, xs_c, xs_h};
    const int tiles_x = ((Wi + 1) / 2 + 31) / 32, tiles_y = ((Hi + 1) / 2 + 3) / 4, groups = (c_in + 31) / 32;
    constexpr size_t smem = sizeof(float) * (S2Bwd<3>::WIN + S2_X3_CHUNK_FLOATS);
    hipLaunchKernelGGL(k_conv_s2_bwd3_x3, dim3((unsigned)(tiles_x * tiles_y * n_img * groups)), dim3(256), smem, (hipStream_t)stream, records, dy,
                       di, Ho, Wo, dx, xo, Hi, Wi, c_in, c_out, groups, tiles_x, tiles_y);
    NF_LAUNCH_CHECK("nf_conv_s2_bwd_x3");
    return 0;
}

// ---- backward-data of the 7x7 stem (at most 4 input channels): the four parity classes share ONE accumulator tile -- row
// m = 4 c + 2 ya + xb -- and one 4 x 4 tap grid (zero weights where a class has only 3 taps), on v_mfma_f32_16x16x4_f32: 16 rows
// (12 used by the 3 colour channels) x 16 pixels x 4 dy channels per instruction, a wave = one row u x 32 columns v = two
// accumulator tiles.  Per (u, v) position and dy channel this is 16 x 0.5 matrix-core cycles against 49 x 1 in the per-class
// 32 x 32 form (measured on the stem: 1.30 ms per-class, 0.45 ms merged on 32 x 32 x 2, see DESIGN for this form).
typedef float s2_f4v __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256, 2) k_conv_s2_bwd_stem(const float* __restrict__ rec, const float* __restrict__ dy, S2Tensor di, int Ho,
                                                             int Wo, float* __restrict__ dx, S2Tensor xo, int Hi, int Wi, int C, int K,
                                                             int tiles_x, int tiles_y) {
    constexpr int CK = S2_STEM_CK, T0 = 4, WR = 4 + T0 - 1, WCOLS = 32 + T0 - 1, CHF = WR * S2_PW, WIN = CK * CHF, STEPS = (CK / 4) * 16;
    HIP_DYNAMIC_SHARED(float, smem)
    float* win = smem;
    float* wgt = smem + WIN;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y, n = bid / tiles_y;
    const int u0 = ty * 4, v0 = tx * 32;
    const int chunks = (K + CK - 1) / CK;
    const float* dn = dy + n * di.ns;
    const int bbase = q * CHF + (T0 - 1 + w) * S2_PW + (T0 - 1) + col;
    s2_f4v acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    S2BwdStage<CK, WR, WCOLS, T0, STEPS * 64> stage;
    stage.fetch(dn, di, Ho, Wo, K, 0, u0, v0, rec, w, lane);
    for (int ch = 0; ch < chunks; ++ch) {
        __syncthreads();
        stage.commit(win, wgt, w, lane);
        __syncthreads();
        if (ch + 1 < chunks) stage.fetch(dn, di, Ho, Wo, K, ch + 1, u0, v0, rec, w, lane);
        int s = 0;
#pragma unroll
        for (int gq = 0; gq < CK / 4; ++gq)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj, ++s) {
                    const float a = wgt[s * 64 + lane];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, win[bbase + 4 * gq * CHF - i * S2_PW - jj + 16 * t], acc[t], 0, 0, 0);
                }
    }
    // accumulator register r of lane (col, q): row m = 4 q + r = (c = q, ya = r >> 1, xb = r & 1), pixel v = v0 + 16 t + col
    float* xn = dx + n * xo.ns;
    const int u = u0 + w;
    if (q < C) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int X = 2 * (v0 + 16 * t + col);
#pragma unroll
            for (int ya = 0; ya < 2; ++ya) {
                const int Y = 2 * u + ya;
                if (Y < Hi && X < Wi) {
                    float* p = xn + q * xo.cs + Y * xo.rs + X;
                    if (X + 1 < Wi) *reinterpret_cast<s2_f2*>(p) = s2_f2{acc[t][2 * ya], acc[t][2 * ya + 1]};
                    else p[0] = acc[t][2 * ya];
                }
            }
        }
    }
}

template <int KS>
static int s2_launch_bwd(const float* rec, const float* dy, S2Tensor di, int Ho, int Wo, float* dx, S2Tensor xo, int Hi, int Wi, int n_img,
                         int C, int K, hipStream_t st) {
    using G = S2Bwd<KS>;
    const int tiles_x = ((Wi + 1) / 2 + 31) / 32, tiles_y = ((Hi + 1) / 2 + 3) / 4, groups = (C + 31) / 32;
    constexpr size_t smem = sizeof(float) * G::FLOATS;
    hipLaunchKernelGGL((k_conv_s2_bwd<KS>), dim3((unsigned)(tiles_x * tiles_y * n_img * groups)), dim3(256), smem, st, rec, dy, di, Ho, Wo, dx,
                       xo, Hi, Wi, C, K, groups, tiles_x, tiles_y);
    return 0;
}

/* dx[n,c,Y,X] = sum_k sum_{a = Y mod 2, b = X mod 2 (mod 2)} W[k][c][a][b] dy[n,k,(Y-a)/2,(X-b)/2] for ALL positions of the
 * Hi x Wi input (rows / columns the convolution never read get zeros).  records: nf_conv_s2_pack(..., backward = 1). */
extern "C" int nf_conv_s2_bwd(const float* records, int ks, const float* dy, int64_t ds_n, int64_t ds_c, int64_t ds_h, int Ho, int Wo,
                              float* dx, int64_t xs_n, int64_t xs_c, int64_t xs_h, int Hi, int Wi, int n_img, int c_in, int c_out,
                              nf_stream_t stream) {
    NF_REQUIRE((ks == 3 || ks == 7) && n_img >= 1 && c_in >= 1 && c_out >= 1 && Hi >= ks && Wi >= ks, "nf_conv_s2_bwd: bad arguments (ks %d)", ks);
    NF_REQUIRE(Ho == (Hi - ks) / 2 + 1 && Wo == (Wi - ks) / 2 + 1, "nf_conv_s2_bwd: gradient %d x %d does not match input %d x %d", Ho, Wo, Hi, Wi);
    hipStream_t st = (hipStream_t)stream;
    const S2Tensor di{ds_n, ds_c, ds_h}, xo{xs_n, xs_c, xs_h};
    if (ks == 7) {
        NF_REQUIRE(c_in <= 4, "nf_conv_s2_bwd: the 7x7 form takes at most 4 input channels (got %d)", c_in);
        const int tiles_x = ((Wi + 1) / 2 + 31) / 32, tiles_y = ((Hi + 1) / 2 + 3) / 4;
        constexpr size_t smem = sizeof(float) * (S2_STEM_CK * 7 * S2_PW + (S2_STEM_CK / 4) * 16 * 64);
        hipLaunchKernelGGL(k_conv_s2_bwd_stem, dim3((unsigned)(tiles_x * tiles_y * n_img)), dim3(256), smem, st, records, dy, di, Ho, Wo, dx, xo,
                           Hi, Wi, c_in, c_out, tiles_x, tiles_y);
    } else {
        s2_launch_bwd<3>(records, dy, di, Ho, Wo, dx, xo, Hi, Wi, n_img, c_in, c_out, st);
    }
    NF_LAUNCH_CHECK("nf_conv_s2_bwd");
    return 0;
}
