// 3x3 stride-1 convolutions as Winograd F(4x4, 3x3) on the fp32 matrix cores: 36 element-wise products per 6x6 input tile and
// 4x4 output tile -- 2.25 matrix-core products per output where F(2x2, 3x3) (csrc/nf_wino.hip) needs 4 and the direct form 9.
// ref: ibrnet/feature_network.py:28-36, 38-78, 127-151 (the stride-1 3x3 convolutions on pre-padded activations; backward-data =
// the same kernel on the zero-extended gradient with rotated weights, exactly as nf_wino.hip).
//
//   Y = A^T [ sum_c (G g G^T) . (B^T d B) ] A,   B^T 6x6, G 6x3, A^T 4x6 (Lavin & Gray's matrices for the points 0, +-1, +-2, inf)
//
// The 36 products are 36 independent GEMMs M_xi[k][tile] = sum_c U_xi[k][c] V_xi[c][tile] on v_mfma_f32_32x32x2_f32: rows = 32
// output channels (A operand = transformed weights, host-packed), columns = 32 tiles (a 4 x 8 block of 4x4-output tiles =
// 16 x 32 outputs), k = input channel pairs (lane half).  A workgroup is SIX waves; wave r owns row r of the 6x6 transformed
// tile: it builds its six B operands of a k-step from the raw window in LDS -- a combination of at most four window rows with
// row r of B^T, then the 6-point column transform, ~40 vector instructions and twelve 8-byte LDS reads per six products -- and
// keeps six accumulator tiles (96 registers).  V never exists in memory.  Output transform: the column half (6 -> 4) inside the
// wave, the row half (6 waves -> 4 output rows) through LDS in two rounds of 8 accumulator registers, 16-byte stores.
// Staging in the style of csrc/nf_conv_s2.hip: 8-channel chunks (4 k-steps); the next chunk's window is fetched into registers
// under the current chunk's matrix-core work and committed behind one barrier pair; the weight records go from L2 straight into
// registers one k-step ahead (a lane needs exactly one dword per product).  fp32 error of F(4x4): 2-4e-6 of full scale (F(2x2): 2e-7) -- inside every tolerance of the parity tests;
// chosen per layer shape against the F(2x2) kernel by the executor's one-off timing.
#include "nf_common.h"
#include "nf_wino4.h"

typedef float q16 __attribute__((ext_vector_type(16)));
typedef float q2a __attribute__((ext_vector_type(2)));                    // 8-byte aligned LDS pair
typedef float q4 __attribute__((ext_vector_type(4)));
typedef float q4u __attribute__((ext_vector_type(4), aligned(4)));
#define W4_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__host__ __device__ constexpr int w4_nidx(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

#ifndef W4_CC
#define W4_CC 8             // input channels per chunk (4 k-steps)
#endif
#ifndef W4_OCC
#define W4_OCC 2            // waves per SIMD the register allocation aims at.  3 (two 6-wave workgroups per CU) needs <= 168 registers:
                            // 12 vector + 41 scalar spills and 141 us on the 64 -> 64 layer; 2 (one workgroup per CU, no spills): 116 us
#endif
#define W4_WR 18            // window rows of a 16-row output block
#define W4_WC 34            // window columns of a 32-column output block
#define W4_WS 36            // LDS row stride of the window
#define W4_CHF (W4_WR * W4_WS)
#define W4_WIN (W4_CC * W4_CHF)
#define W4_WGT ((W4_CC / 2) * 36 * 64)                                    // floats of weight records per chunk (streamed, not staged)
#define W4_EX (6 * 4 * 8 * 64)                                            // one round of the output exchange, floats
#define W4_LDS_FLOATS (W4_EX > W4_WIN ? W4_EX : W4_WIN)

extern "C" int64_t nf_wino4_pack_floats(int c_out, int c_in) {
    return (int64_t)((c_out + 31) / 32) * ((c_in + W4_CC - 1) / W4_CC) * W4_WGT + 36 * 64;      // + one step of slack for the look-ahead
}

/* HOST: weight [c_out][c_in][3][3] -> U = G g G^T as records [group of 32 outputs][chunk of 4 inputs][step][row r][column s][lane (i, h)]
 * = U[r][s] of (k = 32 g + i, c = 4 chunk + 2 step + h).  backward != 0 packs the backward-data convolution:
 * g'[c][k][a][b] = g[k][c][2-a][2-b], roles of c_out / c_in swapped (the records then describe a convolution with c_in OUTPUTS). */
extern "C" int nf_wino4_pack(const float* weight, int c_out, int c_in, int backward, float* out) {
    static const double G[6][3] = {{0.25, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int N = backward ? c_in : c_out, C = backward ? c_out : c_in;      // outputs / inputs of the packed convolution
    const int groups = (N + 31) / 32, chunks = (C + W4_CC - 1) / W4_CC;
    float* rec = out;
    for (int g = 0; g < groups; ++g)
        for (int ch = 0; ch < chunks; ++ch)
            for (int st = 0; st < W4_CC / 2; ++st)
                for (int r = 0; r < 6; ++r)
                    for (int s = 0; s < 6; ++s)
                        for (int lane = 0; lane < 64; ++lane, ++rec) {
                            const int k = 32 * g + (lane & 31), c = ch * W4_CC + 2 * st + (lane >> 5);
                            double u = 0.0;
                            if (k < N && c < C)
                                for (int a = 0; a < 3; ++a)
                                    for (int b = 0; b < 3; ++b) {
                                        const float gv = backward ? weight[(((size_t)c * c_in + k) * 3 + (2 - a)) * 3 + (2 - b)]
                                                                  : weight[(((size_t)k * c_in + c) * 3 + a) * 3 + b];
                                        u += G[r][a] * (double)gv * G[s][b];
                                    }
                            *rec = (float)u;
                        }
    for (int i = 0; i < 36 * 64; ++i) *rec++ = 0.f;
    return (rec - out) == nf_wino4_pack_floats(N, C) ? 0 : 2;
}

struct W4Tensor { int64_t ns, cs, rs; };        // element strides: image, channel, row (unit column stride)

// keeps the compiler from merging the k-steps of a chunk into one block (it then reads all their window rows up front: 100+ live
// registers on top of the 96 accumulators, i.e. spills)
__device__ __forceinline__ void w4_step_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
}

__device__ __forceinline__ int w4_uniform(int v) {        // wave-uniform value in a scalar register
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
}

__global__ void __launch_bounds__(384, W4_OCC) k_wino4(const float* __restrict__ rec, const float* __restrict__ x, W4Tensor xi, int Hi, int Wi, int pad,
                                                  float* __restrict__ y, W4Tensor yo, int Ho, int Wo, int C, int K, int groups,
                                                  int tiles_x, int tiles_y) {
    HIP_DYNAMIC_SHARED(float, smem)
    float* win = smem;
    const int lane = threadIdx.x & 63, w = w4_uniform(threadIdx.x >> 6);
    const int t = lane & 31, h = lane >> 5, tr = t >> 3, tc = t & 7;
    int bid = blockIdx.x;
    const int grp = bid % groups;
    bid /= groups;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y, n = bid / tiles_y;
    const int oy0 = ty * 16, ox0 = tx * 32, iy0 = oy0 - pad, ix0 = ox0 - pad;
    const int chunks = (C + W4_CC - 1) / W4_CC;
    const float* xn = x + n * xi.ns;
    const float* wsrc = rec + (size_t)grp * chunks * W4_WGT;

    // row r = w of B^T: at most four window rows and their coefficients (wave-uniform)
    //   r0: 4 d0 - 5 d2 + d4      r1: -4 d1 - 4 d2 + d3 + d4     r2: 4 d1 - 4 d2 - d3 + d4
    //   r3: -2 d1 - d2 + 2 d3 + d4   r4: 2 d1 - d2 - 2 d3 + d4   r5: 4 d1 - 5 d3 + d5
    int ri0, ri1, ri2, ri3;
    float rc0, rc1, rc2, rc3;
    switch (w) {
        case 0: ri0 = 0, ri1 = 2, ri2 = 4, ri3 = 4, rc0 = 4.f, rc1 = -5.f, rc2 = 1.f, rc3 = 0.f; break;
        case 1: ri0 = 1, ri1 = 2, ri2 = 3, ri3 = 4, rc0 = -4.f, rc1 = -4.f, rc2 = 1.f, rc3 = 1.f; break;
        case 2: ri0 = 1, ri1 = 2, ri2 = 3, ri3 = 4, rc0 = 4.f, rc1 = -4.f, rc2 = -1.f, rc3 = 1.f; break;
        case 3: ri0 = 1, ri1 = 2, ri2 = 3, ri3 = 4, rc0 = -2.f, rc1 = -1.f, rc2 = 2.f, rc3 = 1.f; break;
        case 4: ri0 = 1, ri1 = 2, ri2 = 3, ri3 = 4, rc0 = 2.f, rc1 = -1.f, rc2 = -2.f, rc3 = 1.f; break;
        default: ri0 = 1, ri1 = 3, ri2 = 5, ri3 = 5, rc0 = 4.f, rc1 = -5.f, rc2 = 1.f, rc3 = 0.f; break;
    }
    const int wbase = h * W4_CHF + (4 * tr) * W4_WS + 4 * tc;
    const int ro0 = wbase + ri0 * W4_WS, ro1 = wbase + ri1 * W4_WS, ro2 = wbase + ri2 * W4_WS, ro3 = wbase + ri3 * W4_WS;

    q16 acc[6];
#pragma unroll
    for (int s = 0; s < 6; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

    // ---- window staging, software-pipelined: 72 row pairs (8 channels x 9) over 6 waves = 12 per wave, each wave instruction
    // moves two window rows x 32 columns (lane half = row of the pair); columns 32, 33 in one extra item per thread
    constexpr int NROW = (W4_CC * (W4_WR / 2) + 5) / 6;
    constexpr int NXC = (W4_CC * W4_WR * (W4_WC - 32) + 383) / 384;
    float pre_w[NROW], pre_x[NXC];
    const int sub = lane >> 5;
    auto fetch = [&](int ch) {
#pragma unroll
        for (int it = 0; it < NROW; ++it) {
            const int pi = it * 6 + w;                                  // wave-uniform pair index
            const int c = pi / (W4_WR / 2), r = 2 * (pi - c * (W4_WR / 2)) + sub;
            const int gc = ch * W4_CC + c, gy = iy0 + r, gx = ix0 + t;
            float v = 0.f;
            if (pi < W4_CC * (W4_WR / 2) && gc < C && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi) v = xn[gc * xi.cs + gy * xi.rs + gx];
            pre_w[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NXC; ++it) {
            const int i = it * 384 + (int)threadIdx.x;
            float v = 0.f;
            if (i < W4_CC * W4_WR * (W4_WC - 32)) {
                const int rr = i / (W4_WC - 32), col = 32 + (i - rr * (W4_WC - 32));
                const int c = rr / W4_WR, r = rr - c * W4_WR;
                const int gc = ch * W4_CC + c, gy = iy0 + r, gx = ix0 + col;
                if (gc < C && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi) v = xn[gc * xi.cs + gy * xi.rs + gx];
            }
            pre_x[it] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < NROW; ++it) {
            const int pi = it * 6 + w;
            const int c = pi / (W4_WR / 2), r = 2 * (pi - c * (W4_WR / 2)) + sub;
            if (pi < W4_CC * (W4_WR / 2)) win[c * W4_CHF + r * W4_WS + t] = pre_w[it];
        }
#pragma unroll
        for (int it = 0; it < NXC; ++it) {
            const int i = it * 384 + (int)threadIdx.x;
            if (i < W4_CC * W4_WR * (W4_WC - 32)) {
                const int rr = i / (W4_WC - 32), col = 32 + (i - rr * (W4_WC - 32));
                const int c = rr / W4_WR, r = rr - c * W4_WR;
                win[c * W4_CHF + r * W4_WS + col] = pre_x[it];
            }
        }
    };
    // ---- weight records: this wave's six A operands of a k-step straight from L2 (one dword per lane and product, 256
    // contiguous bytes per wave instruction), one step ahead of their use -- they never touch LDS
    const float* wp = wsrc + (w * 6) * 64 + lane;
    float a_nxt[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) a_nxt[s] = wp[s * 64];
    wp += 36 * 64;

#ifndef W4_PREFETCH
#define W4_PREFETCH 1          // next chunk's window fetched into registers under this chunk's products (125 -> 116 us at W4_OCC 2)
#endif
    if (W4_PREFETCH) fetch(0);
    for (int ch = 0; ch < chunks; ++ch) {
        __syncthreads();
        if (!W4_PREFETCH) fetch(ch);
        commit();
        __syncthreads();
        if (W4_PREFETCH && ch + 1 < chunks) fetch(ch + 1);
#pragma unroll
        for (int st = 0; st < W4_CC / 2; ++st) {
            float a_cur[6];
#pragma unroll
            for (int s = 0; s < 6; ++s) a_cur[s] = a_nxt[s];
#pragma unroll
            for (int s = 0; s < 6; ++s) a_nxt[s] = wp[s * 64];          // the pack carries one step of zeros behind the last one
            wp += 36 * 64;
            // row combination: tj[j] = sum_q rc_q d[ri_q][j], j = 0..5 (three 8-byte pairs per window row)
            float tj[6];
#pragma unroll
            for (int jp = 0; jp < 3; ++jp) {
                const int off = 2 * st * W4_CHF + 2 * jp;
                const q2a a0 = *reinterpret_cast<const q2a*>(win + ro0 + off), a1 = *reinterpret_cast<const q2a*>(win + ro1 + off);
                const q2a a2 = *reinterpret_cast<const q2a*>(win + ro2 + off), a3 = *reinterpret_cast<const q2a*>(win + ro3 + off);
                tj[2 * jp] = fmaf(rc3, a3[0], fmaf(rc2, a2[0], fmaf(rc1, a1[0], rc0 * a0[0])));
                tj[2 * jp + 1] = fmaf(rc3, a3[1], fmaf(rc2, a2[1], fmaf(rc1, a1[1], rc0 * a0[1])));
            }
            // column transform v = B^T tj
            const float p = fmaf(-4.f, tj[2], tj[4]), q = fmaf(-4.f, tj[1], tj[3]);        // t4 - 4 t2, t3 - 4 t1
            const float cc = tj[4] - tj[2], dd = 2.f * (tj[3] - tj[1]);
            float v[6];
            v[0] = fmaf(4.f, tj[0], fmaf(-5.f, tj[2], tj[4]));
            v[1] = p + q;
            v[2] = p - q;
            v[3] = cc + dd;
            v[4] = cc - dd;
            v[5] = fmaf(4.f, tj[1], fmaf(-5.f, tj[3], tj[5]));
#pragma unroll
            for (int s = 0; s < 6; ++s) acc[s] = W4_MFMA(a_cur[s], v[s], acc[s]);
            w4_step_fence();
        }
    }
    __syncthreads();      // every wave is done with the window and the records: the same memory now carries the output transform

    // ---- output transform Y = A^T M A.  Column half inside the wave: z_j = sum_s A^T[j][s] M[r][s]
    //      A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
    float* ex = smem;
    float at0, at1, at2, at3, at4, at5;          // row i = w of A^T (waves 0..3 produce output row i of every tile)
    switch (w) {
        case 0: at0 = 1.f, at1 = 1.f, at2 = 1.f, at3 = 1.f, at4 = 1.f, at5 = 0.f; break;
        case 1: at0 = 0.f, at1 = 1.f, at2 = -1.f, at3 = 2.f, at4 = -2.f, at5 = 0.f; break;
        case 2: at0 = 0.f, at1 = 1.f, at2 = 1.f, at3 = 4.f, at4 = 4.f, at5 = 0.f; break;
        default: at0 = 0.f, at1 = 1.f, at2 = -1.f, at3 = 8.f, at4 = -8.f, at5 = 1.f; break;
    }
    float* yn = y + n * yo.ns;
    const int orow = oy0 + 4 * tr + w, ocol = ox0 + 4 * tc;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        // exchange image of a round: [wave r][j][q4 = 0..1][lane][4]: four consecutive accumulator registers as one 16-byte access
#pragma unroll
        for (int g4 = 0; g4 < 2; ++g4) {
            q4 z0, z1, z2, z3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 8 * hf + 4 * g4 + e;
                const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                z0[e] = m0 + s12 + s34;
                z1[e] = fmaf(2.f, d34, d12);
                z2[e] = fmaf(4.f, s34, s12);
                z3[e] = fmaf(8.f, d34, d12) + m5;
            }
            float* e0 = ex + (((w * 4 + 0) * 2 + g4) * 64 + lane) * 4;
            *reinterpret_cast<q4*>(e0) = z0;
            *reinterpret_cast<q4*>(e0 + 2 * 256) = z1;
            *reinterpret_cast<q4*>(e0 + 4 * 256) = z2;
            *reinterpret_cast<q4*>(e0 + 6 * 256) = z3;
        }
        __syncthreads();
        if (w < 4) {
#pragma unroll
            for (int g4 = 0; g4 < 2; ++g4) {
                q4 o[4];          // o[j][e]: output column j, register 8 hf + 4 g4 + e
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* b = ex + ((j * 2 + g4) * 64 + lane) * 4;
                    const q4 r0 = *reinterpret_cast<const q4*>(b), r1 = *reinterpret_cast<const q4*>(b + 1 * 2048);
                    const q4 r2 = *reinterpret_cast<const q4*>(b + 2 * 2048), r3 = *reinterpret_cast<const q4*>(b + 3 * 2048);
                    const q4 r4 = *reinterpret_cast<const q4*>(b + 4 * 2048), r5 = *reinterpret_cast<const q4*>(b + 5 * 2048);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o[j][e] = fmaf(at5, r5[e], fmaf(at4, r4[e], fmaf(at3, r3[e], fmaf(at2, r2[e], fmaf(at1, r1[e], at0 * r0[e])))));
                }
                if (orow < Ho) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = 32 * grp + w4_nidx(8 * hf + 4 * g4 + e, h);
                        if (k < K) {
                            float* yp = yn + k * yo.cs + orow * yo.rs + ocol;
                            if (ocol + 3 < Wo) *reinterpret_cast<q4u*>(yp) = q4u{o[0][e], o[1][e], o[2][e], o[3][e]};
                            else {
                                if (ocol < Wo) yp[0] = o[0][e];
                                if (ocol + 1 < Wo) yp[1] = o[1][e];
                                if (ocol + 2 < Wo) yp[2] = o[2][e];
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();      // the exchange image is free again
    }
}

/* y[n, k, oy, ox] = sum_c sum_{a,b<3} x[n, c, oy + a - pad, ox + b - pad] W[k][c][a][b]  (zeros outside x), output Ho x Wo.
 * Forward of the network's padding-0 convolutions: pad = 0, Ho = Hi - 2.  Backward-data: x = d y, pad = 2, Ho = Hi + 2,
 * records packed with backward != 0.  x, y: element strides (image, channel, row), unit column stride. */
extern "C" int nf_conv3x3_wino4(const float* records, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h, int Hi, int Wi, int pad,
                                float* y, int64_t ys_n, int64_t ys_c, int64_t ys_h, int Ho, int Wo, int n_img, int c_in, int c_out,
                                nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && c_in >= 1 && c_out >= 1 && Hi >= 1 && Wi >= 1 && Ho >= 1 && Wo >= 1 && (pad == 0 || pad == 2),
               "nf_conv3x3_wino4: bad arguments");
    NF_REQUIRE(Ho == Hi - 2 + 2 * pad && Wo == Wi - 2 + 2 * pad, "nf_conv3x3_wino4: output %d x %d does not match input %d x %d, pad %d", Ho, Wo, Hi,
               Wi, pad);
    const int tiles_x = (Wo + 31) / 32, tiles_y = (Ho + 15) / 16, groups = (c_out + 31) / 32;
    constexpr size_t smem = sizeof(float) * W4_LDS_FLOATS;
    static bool once_on[NF_MAX_DEVICES] = {};
    bool& once = once_on[nf_current_device()];
    if (!once) {
        if (smem > 64 * 1024 && hipFuncSetAttribute((const void*)k_wino4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
            nf_set_error("nf_conv3x3_wino4: cannot reserve %zu bytes of LDS", smem);
            return 1;
        }
        once = true;
    }
    const W4Tensor xi{xs_n, xs_c, xs_h}, yo{ys_n, ys_c, ys_h};
    hipLaunchKernelGGL(k_wino4, dim3((unsigned)(tiles_x * tiles_y * n_img * groups)), dim3(384), smem, (hipStream_t)stream, records, x, xi, Hi, Wi,
                       pad, y, yo, Ho, Wo, c_in, c_out, groups, tiles_x, tiles_y);
    NF_LAUNCH_CHECK("nf_conv3x3_wino4");
    return 0;
}
