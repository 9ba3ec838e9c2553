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
    if (ks == 7) return (int64_t)((c_out + S2_STEM_CK - 1) / S2_STEM_CK) * (S2_STEM_CK / 2) * 16 * 64;      // stem form, see below
    const int ck = 16;
    const int64_t groups = (c_in + 31) / 32, chunks = (c_out + ck - 1) / ck;
    return groups * chunks * (ck / 2) * ks * ks * 64;
}

/* HOST: weight [c_out][c_in][ks][ks] -> records in consumption order.
 * forward:  [group of 64 outputs][chunk][step][tile t][lane (i, h)] = W[64 g + 32 t + i][c][a][b]
 * backward: [group of 32 inputs ][chunk of CK outputs][class (ya, xb)][tap (i, j)][pair p][lane (i_c, h)]
 *           = W[k = chunk CK + 2 p + h][c = 32 g + i_c][a = 2 i + ya][b = 2 j + xb]
 * backward, 7x7 (c_in <= 8): [chunk of 8 outputs][pair p][tap (i, j) of 4 x 4][lane (m, h)], m = 4 c + 2 ya + xb
 *           = W[k = chunk 8 + 2 p + h][c][2 i + ya][2 j + xb]  (zero where the tap lies outside the 7 x 7 kernel) */
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
            for (int p = 0; p < S2_STEM_CK / 2; ++p)
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j)
                        for (int lane = 0; lane < 64; ++lane, ++rec) {
                            const int m = lane & 31, c = m >> 2, ya = (m >> 1) & 1, xb = m & 1;
                            *rec = W(ch * S2_STEM_CK + 2 * p + (lane >> 5), c, 2 * i + ya, 2 * j + xb);
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

    for (int ch = 0; ch < chunks; ++ch) {
        __syncthreads();        // the previous chunk's operands are consumed
        // ---- stage the raw window of this chunk, columns split by parity; zeros outside the tensor.  The window starts at an
        // even column of an even-width row, so a lane moves one 8-byte column pair: 32 lanes cover columns 0..63 of a window
        // row (the two halves of a wave take two rows), element 2 l goes to parity plane 0, 2 l + 1 to plane 1, both at index l
        {
            const int l = lane & 31, sub = lane >> 5;
            for (int r0 = 0; r0 < CC * WR; r0 += 8) {
                const int rr = r0 + 2 * w + sub;                       // window row (channel-major)
                if (rr < CC * WR) {
                    const int c = rr / WR, r = rr - c * WR;
                    const int gc = ch * CC + c, gy = iy0 + r, gx = ix0 + 2 * l;
                    s2_f2 v = {0.f, 0.f};
                    if (gc < C && gy < Hi) {
                        const float* src = xn + gc * xi.cs + gy * xi.rs + gx;
                        if (pair_ok && gx + 1 < Wi) v = *reinterpret_cast<const s2_f2*>(src);
                        else {
                            if (gx < Wi) v[0] = src[0];
                            if (gx + 1 < Wi) v[1] = src[1];
                        }
                    }
                    float* dst = win + c * CHF + r * ROWF + l;
                    dst[0] = v[0];
                    dst[S2_PW] = v[1];
                }
            }
            // the columns beyond 63 (1 for 3x3, 6 for 7x7): one element per thread
            constexpr int XC = WC - 64;
            for (int i = threadIdx.x; i < CC * WR * XC; i += 256) {
                const int rr = i / XC, col = 64 + (i - rr * XC);
                const int c = rr / WR, r = rr - c * WR;
                const int gc = ch * CC + c, gy = iy0 + r, gx = ix0 + col;
                float v = 0.f;
                if (gc < C && gy < Hi && gx < Wi) v = xn[gc * xi.cs + gy * xi.rs + gx];
                win[c * CHF + r * ROWF + (col & 1) * S2_PW + (col >> 1)] = v;
            }
        }
        // ---- and its weight records (contiguous, 16-byte pieces)
        {
            const s2_f4* src = reinterpret_cast<const s2_f4*>(wsrc + (size_t)ch * (G::STEPS * 128));
            s2_f4* dst = reinterpret_cast<s2_f4*>(wgt);
            for (int i = threadIdx.x; i < G::STEPS * 32; i += 256) dst[i] = src[i];
        }
        __syncthreads();
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

    for (int ch = 0; ch < chunks; ++ch) {
        __syncthreads();
        // dy window: the two halves of a wave take two window rows, lane l column l; the columns beyond 31 in a second sweep
        {
            const int l = lane & 31, sub = lane >> 5;
            for (int r0 = 0; r0 < CK * WR; r0 += 8) {
                const int rr = r0 + 2 * w + sub;
                if (rr < CK * WR) {
                    const int c = rr / WR, r = rr - c * WR;
                    const int gk = ch * CK + c, gu = u0 - (T0 - 1) + r, gv = v0 - (T0 - 1) + l;
                    float v = 0.f;
                    if (gk < K && gu >= 0 && gu < Ho && gv >= 0 && gv < Wo) v = dn[gk * di.cs + gu * di.rs + gv];
                    win[c * CHF + r * S2_PW + l] = v;
                }
            }
            constexpr int XC = G::WCOLS - 32;
            for (int i = threadIdx.x; i < CK * WR * XC; i += 256) {
                const int rr = i / XC, col = 32 + (i - rr * XC);
                const int c = rr / WR, r = rr - c * WR;
                const int gk = ch * CK + c, gu = u0 - (T0 - 1) + r, gv = v0 - (T0 - 1) + col;
                float v = 0.f;
                if (gk < K && gu >= 0 && gu < Ho && gv >= 0 && gv < Wo) v = dn[gk * di.cs + gu * di.rs + gv];
                win[c * CHF + r * S2_PW + col] = v;
            }
        }
        {
            const s2_f4* src = reinterpret_cast<const s2_f4*>(wsrc + (size_t)ch * (G::STEPS * 64));
            s2_f4* dst = reinterpret_cast<s2_f4*>(wgt);
            for (int i = threadIdx.x; i < G::STEPS * 16; i += 256) dst[i] = src[i];
        }
        __syncthreads();
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

// ---- backward-data of the 7x7 stem (at most 8 input channels): the four parity classes share ONE accumulator tile -- row
// m = 4 c + 2 ya + xb -- and one 4 x 4 tap grid (zero weights where a class has only 3 taps), so a (u, v) position costs 16
// matrix-core products per dy channel pair instead of the 49 of the per-class form above (measured: 1.30 ms -> see DESIGN)
__global__ void __launch_bounds__(256, 2) k_conv_s2_bwd_stem(const float* __restrict__ rec, const float* __restrict__ dy, S2Tensor di, int Ho,
                                                             int Wo, float* __restrict__ dx, S2Tensor xo, int Hi, int Wi, int C, int K,
                                                             int tiles_x, int tiles_y) {
    constexpr int CK = S2_STEM_CK, T0 = 4, WR = 4 + T0 - 1, WCOLS = 32 + T0 - 1, CHF = WR * S2_PW, WIN = CK * CHF, STEPS = (CK / 2) * 16;
    HIP_DYNAMIC_SHARED(float, smem)
    float* win = smem;
    float* wgt = smem + WIN;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y, n = bid / tiles_y;
    const int u0 = ty * 4, v0 = tx * 32;
    const int chunks = (K + CK - 1) / CK;
    const float* dn = dy + n * di.ns;
    const int bbase = h * CHF + (T0 - 1 + w) * S2_PW + (T0 - 1) + j;
    s2_f16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int ch = 0; ch < chunks; ++ch) {
        __syncthreads();
        {
            const int l = lane & 31, sub = lane >> 5;
            for (int r0 = 0; r0 < CK * WR; r0 += 8) {
                const int rr = r0 + 2 * w + sub;
                if (rr < CK * WR) {
                    const int c = rr / WR, r = rr - c * WR;
                    const int gk = ch * CK + c, gu = u0 - (T0 - 1) + r, gv = v0 - (T0 - 1) + l;
                    float v = 0.f;
                    if (gk < K && gu >= 0 && gu < Ho && gv >= 0 && gv < Wo) v = dn[gk * di.cs + gu * di.rs + gv];
                    win[c * CHF + r * S2_PW + l] = v;
                }
            }
            constexpr int XC = WCOLS - 32;
            for (int i = threadIdx.x; i < CK * WR * XC; i += 256) {
                const int rr = i / XC, col = 32 + (i - rr * XC);
                const int c = rr / WR, r = rr - c * WR;
                const int gk = ch * CK + c, gu = u0 - (T0 - 1) + r, gv = v0 - (T0 - 1) + col;
                float v = 0.f;
                if (gk < K && gu >= 0 && gu < Ho && gv >= 0 && gv < Wo) v = dn[gk * di.cs + gu * di.rs + gv];
                win[c * CHF + r * S2_PW + col] = v;
            }
        }
        {
            const s2_f4* src = reinterpret_cast<const s2_f4*>(rec + (size_t)ch * (STEPS * 64));
            s2_f4* dst = reinterpret_cast<s2_f4*>(wgt);
            for (int i = threadIdx.x; i < STEPS * 16; i += 256) dst[i] = src[i];
        }
        __syncthreads();
        int s = 0;
#pragma unroll
        for (int p = 0; p < CK / 2; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj, ++s) acc = S2_MFMA(wgt[s * 64 + lane], win[bbase + 2 * p * CHF - i * S2_PW - jj], acc);
    }
    float* xn = dx + n * xo.ns;
    const int u = u0 + w, v = v0 + j, X = 2 * v;
#pragma unroll
    for (int rr = 0; rr < 16; rr += 2) {
        const int m = s2_nidx(rr, h), c = m >> 2, ya = (m >> 1) & 1;       // rows m, m + 1 = the two column classes
        const int Y = 2 * u + ya;
        if (c < C && Y < Hi && X < Wi) {
            float* p = xn + c * xo.cs + Y * xo.rs + X;
            if (X + 1 < Wi) *reinterpret_cast<s2_f2*>(p) = s2_f2{acc[rr], acc[rr + 1]};
            else p[0] = acc[rr];
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
        NF_REQUIRE(c_in <= 8, "nf_conv_s2_bwd: the 7x7 form takes at most 8 input channels (got %d)", c_in);
        const int tiles_x = ((Wi + 1) / 2 + 31) / 32, tiles_y = ((Hi + 1) / 2 + 3) / 4;
        constexpr size_t smem = sizeof(float) * (S2_STEM_CK * 7 * S2_PW + (S2_STEM_CK / 2) * 16 * 64);
        hipLaunchKernelGGL(k_conv_s2_bwd_stem, dim3((unsigned)(tiles_x * tiles_y * n_img)), dim3(256), smem, st, records, dy, di, Ho, Wo, dx, xo,
                           Hi, Wi, c_in, c_out, tiles_x, tiles_y);
    } else {
        s2_launch_bwd<3>(records, dy, di, Ho, Wo, dx, xo, Hi, Wi, n_img, c_in, c_out, st);
    }
    NF_LAUNCH_CHECK("nf_conv_s2_bwd");
    return 0;
}
