// 3x3 stride-1 convolutions of the ResUNet (a14) as Winograd F(2x2, 3x3) on the fp32 matrix cores.
// ref: ibrnet/feature_network.py:28-36, 38-78, 127-151 (every 3x3 convolution of the network that has stride 1; the
// reflect padding is already in the input -- csrc/nf_cnn.hip writes pre-padded activations -- so the convolution itself
// runs with padding 0, and its backward-data pass is the same kernel on the zero-extended gradient with rotated weights).
//
//   Y = A^T [ sum_c (G g G^T) . (B^T d B) ] A      per 4x4 input tile d -> 2x2 outputs, 16 element-wise products
// The 16 products are 16 independent GEMMs  M_xi[k][tile] = sum_c U_xi[k][c] V_xi[c][tile]  on v_mfma_f32_32x32x2_f32:
// rows = 32 output channels (A operand = transformed weights, host-packed records streamed from L2), columns = 32 tiles
// (a 4 x 8 block of tiles), k = input channels.  Wave w of the 4-wave workgroup owns row w of the
// 4x4 transformed tile (xi = (w, nu), nu = 0..3): it builds ITS rows of B^T d B straight from the raw input patch in LDS
// (two patch rows, one 4-wide column transform: 8 adds per 4 B operands) -- V never exists in memory -- and keeps
// 4 x KB accumulator tiles.  The column half of A^T M A is local to a wave, the row half crosses the waves through LDS.
// 2.25x fewer multiplies than the direct form.  Each wave carries TWO blocks of 32 tiles, so a weight record feeds two MFMAs;
// the records run through a four-deep register ring (three steps = 1.3 us in flight), the next chunk's patch is fetched
// into registers before the current chunk's multiplications and committed to LDS behind them.
#include "nf_common.h"

typedef float w16 __attribute__((ext_vector_type(16)));
typedef float w2f __attribute__((ext_vector_type(2), aligned(4)));
#define WN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__host__ __device__ constexpr int wn_nidx(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

#define WN_CC 16            // input channels per chunk
#define WN_PR_MAX 18        // patch rows of the two-block variant (8 tile rows: 16 + 2); one block: 10
#define WN_PC 18            // patch columns used (8 tiles: 16 + 2)
#define WN_PS 24            // patch row stride in LDS (the four tile rows of a wave then hit disjoint bank groups)

extern "C" int64_t nf_wino_pack_floats(int c_out, int c_in, int k_per_group) {
    const int groups = (c_out + k_per_group - 1) / k_per_group, chunks = (c_in + WN_CC - 1) / WN_CC;
    return (int64_t)groups * 4 * chunks * 8 * 4 * (k_per_group / 32) * 64 + 4096;      // + room for the last prefetch past the end
}

/* HOST: weight [c_out][c_in][3][3] -> U = G g G^T in record order [group][wave][chunk][step][nu][kb][64 lanes].
 * backward != 0 packs the backward-data convolution: g'[c][k][a][b] = g[k][c][2-a][2-b], roles of c_out / c_in swapped
 * (the records then describe a convolution with `c_in` OUTPUT channels). */
extern "C" int nf_wino_pack(const float* weight, int c_out, int c_in, int backward, int k_per_group, float* out) {
    static const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const int N = backward ? c_in : c_out, C = backward ? c_out : c_in;      // output / input channels of the packed convolution
    if (k_per_group % 32 != 0 || k_per_group < 32) return 1;
    const int KB = k_per_group / 32, groups = (N + k_per_group - 1) / k_per_group, chunks = (C + WN_CC - 1) / WN_CC;
    const int64_t total = nf_wino_pack_floats(N, C, k_per_group);
    for (int64_t i = total - 4096; i < total; ++i) out[i] = 0.f;
    float* rec = out;
    for (int g = 0; g < groups; ++g)
        for (int w = 0; w < 4; ++w)
            for (int ch = 0; ch < chunks; ++ch)
                for (int s = 0; s < 8; ++s)
                    for (int nu = 0; nu < 4; ++nu)
                        for (int kb = 0; kb < KB; ++kb, rec += 64)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int k = g * k_per_group + kb * 32 + (lane & 31), c = ch * WN_CC + 2 * s + (lane >> 5);
                                float u = 0.f;
                                if (k < N && c < C) {
                                    for (int a = 0; a < 3; ++a)
                                        for (int b = 0; b < 3; ++b) {
                                            const float gv = backward ? weight[(((size_t)c * c_in + k) * 3 + (2 - a)) * 3 + (2 - b)]
                                                                      : weight[(((size_t)k * c_in + c) * 3 + a) * 3 + b];
                                            u += G[w][a] * gv * G[nu][b];
                                        }
                                }
                                rec[lane] = u;
                            }
    return 0;
}

struct WnTensor { int64_t ns, cs, rs; };        // element strides: image, channel, row (unit column stride)

// TB = tile blocks per wave: 2 -> the workgroup covers 8 x 8 tiles (16 x 16 output pixels), every weight record feeds two
// MFMAs, one workgroup per CU; 1 -> 4 x 8 tiles (8 x 16 pixels), 128 accumulator registers, two workgroups per CU -- the
// better choice when the image has too few tiles to fill the chip with the big blocks (layer 3: 48 x 63)
template <int KB, int TB>
__global__ void __launch_bounds__(256, TB == 1 ? 2 : 1) k_wino3x3(const float* __restrict__ rec, const float* __restrict__ x, WnTensor xi, int Hi, int Wi,
                                                 int pad, float* __restrict__ y, WnTensor yo, int Ho, int Wo, int C, int K,
                                                 int groups) {
    // the raw input patch of the current 16-channel chunk; after the last chunk the same memory carries the row half of the
    // output transform from wave to wave
    __shared__ float smem[4 * 2 * 16 * 64];
    float* patch = smem;
    float* ex = smem;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = lane & 31, hh = lane >> 5, tr = t >> 3, tc = t & 7;
    const int n = blockIdx.z / groups, grp = blockIdx.z - n * groups;
    constexpr int WN_TB = TB, WN_PR = 8 * TB + 2, WN_CH = WN_PR * WN_PS;
    const int oy0 = blockIdx.y * (8 * TB), ox0 = blockIdx.x * 16;
    const int iy0 = oy0 - pad, ix0 = ox0 - pad;
    const int chunks = (C + WN_CC - 1) / WN_CC;
    const float* xn = x + n * xi.ns;
    // this wave's weight stream: [chunk][step][nu][kb] records, consumed strictly in order
    const float* wr = rec + ((size_t)(grp * 4 + w) * chunks) * (8 * 4 * KB * 64) + lane;
    // B^T d B, row w: which two patch rows, and the sign of the second
    const int ra = w == 0 ? 0 : (w == 2 ? 2 : 1), rb = w == 0 ? 2 : (w == 1 ? 2 : (w == 2 ? 1 : 3));
    const float sb = w == 1 ? 1.f : -1.f;
    const int lbase = (2 * tr) * WN_PS + 2 * tc;

    // staging: thread j < 288 owns patch row (channel j / 18, row j % 18) of every chunk: 18 consecutive floats of the image
    // (five wide loads when the row lies inside the image; element-wise with zero fill at the image border), fetched into
    // registers half a chunk ahead and committed to LDS behind the chunk's multiplications
    const int sj = threadIdx.x, sj2 = threadIdx.x + 256;           // second row for threads 0..31 (two-block variant)
    const bool one_row = sj < WN_CC * WN_PR, two_rows = sj2 < WN_CC * WN_PR;
    float pre[2][WN_PC];
    auto fetch_row = [&](int chunk, int j, float (&dst)[WN_PC]) {
        const int ch = j / WN_PR, pr = j - ch * WN_PR;
        const int c = chunk * WN_CC + ch, gy = iy0 + pr;
        const bool row_ok = c < C && gy >= 0 && gy < Hi;
        const float* src = xn + (row_ok ? c : 0) * xi.cs + (row_ok ? gy : 0) * xi.rs + ix0;
        if (row_ok && ix0 >= 0 && ix0 + WN_PC <= Wi) {
#pragma unroll
            for (int q = 0; q < WN_PC / 2; ++q) {
                const w2f v = *reinterpret_cast<const w2f*>(src + 2 * q);
                dst[2 * q] = v[0];
                dst[2 * q + 1] = v[1];
            }
        } else {
#pragma unroll
            for (int q = 0; q < WN_PC; ++q) {
                const int gx = ix0 + q;
                const bool ok = row_ok && gx >= 0 && gx < Wi;
                const float v = src[ok ? q : -ix0];                 // a valid address either way (column 0 of the row)
                dst[q] = ok ? v : 0.f;
            }
        }
    };
    auto fetch = [&](int chunk) {
        if (one_row) fetch_row(chunk, sj, pre[0]);
        if (two_rows) fetch_row(chunk, sj2, pre[1]);
    };
    auto commit_row = [&](int j, const float (&src)[WN_PC]) {
        const int ch = j / WN_PR, pr = j - ch * WN_PR;
        float* dst = patch + ch * WN_CH + pr * WN_PS;
#pragma unroll
        for (int q = 0; q < WN_PC / 2; ++q) *reinterpret_cast<w2f*>(dst + 2 * q) = w2f{src[2 * q], src[2 * q + 1]};
    };
    auto commit = [&]() {
        if (one_row) commit_row(sj, pre[0]);
        if (two_rows) commit_row(sj2, pre[1]);
    };

    w16 acc[4][KB][WN_TB];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int b = 0; b < WN_TB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nu][kb][b][r] = 0.f;

    // weight ring: four steps in registers, three in flight (each step = 4 KB records feeding 4 KB x WN_TB MFMAs)
    float wb[4][4 * KB];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 4 * KB; ++i) wb[q][i] = wr[(q * 4 * KB + i) * 64];
    fetch(0);
    commit();
    __syncthreads();
    for (int chunk = 0; chunk < chunks; ++chunk) {
        const float* wc = wr + (size_t)chunk * (8 * 4 * KB * 64);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            // step s + 3 (possibly in the next chunk, or the zero pad behind the last one) goes into the slot step s - 1 freed
            asm volatile("" ::: "memory");      // keep the issue order of the loads: the vm counter retires them in order
#pragma unroll
            for (int i = 0; i < 4 * KB; ++i) wb[(s + 3) & 3][i] = wc[((s + 3) * 4 * KB + i) * 64];
            if (s == 3 && chunk + 1 < chunks) fetch(chunk + 1);      // needed four steps (1.7 us) from now
            asm volatile("" ::: "memory");
            float v[WN_TB][4];
#pragma unroll
            for (int b = 0; b < WN_TB; ++b) {
                const float* pa = patch + lbase + (2 * s + hh) * WN_CH + (8 * b) * WN_PS;
                const w2f a0 = *reinterpret_cast<const w2f*>(pa + ra * WN_PS), a1 = *reinterpret_cast<const w2f*>(pa + ra * WN_PS + 2);
                const w2f b0 = *reinterpret_cast<const w2f*>(pa + rb * WN_PS), b1 = *reinterpret_cast<const w2f*>(pa + rb * WN_PS + 2);
                const float e0 = a0[0] + sb * b0[0], e1 = a0[1] + sb * b0[1], e2 = a1[0] + sb * b1[0], e3 = a1[1] + sb * b1[1];
                v[b][0] = e0 - e2;
                v[b][1] = e1 + e2;
                v[b][2] = e2 - e1;
                v[b][3] = e1 - e3;
            }
#pragma unroll
            for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int b = 0; b < WN_TB; ++b) acc[nu][kb][b] = WN_MFMA(wb[s & 3][nu * KB + kb], v[b][nu], acc[nu][kb][b]);
        }
        __syncthreads();                 // every wave is done reading this chunk's patch
        if (chunk + 1 < chunks) {
            commit();
            __syncthreads();
        }
    }
    // ---- output transform Y = A^T M A: columns (nu) inside the wave, rows (w) across the waves through LDS
    const int kbase = grp * (32 * KB);
    float* yn = y + n * yo.ns;
#pragma unroll
    for (int b = 0; b < WN_TB; ++b) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a0 = acc[0][kb][b][r], a1 = acc[1][kb][b][r], a2 = acc[2][kb][b][r], a3 = acc[3][kb][b][r];
                ex[((w * 2 + 0) * 16 + r) * 64 + lane] = a0 + a1 + a2;
                ex[((w * 2 + 1) * 16 + r) * 64 + lane] = a1 - a2 - a3;
            }
            __syncthreads();
            {
                const int oy = w & 1, rsel = 8 * (w >> 1);
                const int row = oy0 + 2 * (4 * b + tr) + oy, col = ox0 + 2 * tc;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = rsel + q;
                    float o[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float m0 = ex[((0 * 2 + j) * 16 + r) * 64 + lane], m1 = ex[((1 * 2 + j) * 16 + r) * 64 + lane];
                        const float m2 = ex[((2 * 2 + j) * 16 + r) * 64 + lane], m3 = ex[((3 * 2 + j) * 16 + r) * 64 + lane];
                        o[j] = oy == 0 ? m0 + m1 + m2 : m1 - m2 - m3;
                    }
                    const int k = kbase + 32 * kb + wn_nidx(r, hh);
                    if (k < K && row < Ho) {
                        float* yp = yn + k * yo.cs + row * yo.rs + col;
                        if (col + 1 < Wo) *reinterpret_cast<w2f*>(yp) = w2f{o[0], o[1]};
                        else if (col < Wo) yp[0] = o[0];
                    }
                }
            }
            __syncthreads();
        }
    }
}

/* y[n, k, oy, ox] = sum_c sum_{a,b<3} x[n, c, oy + a - pad, ox + b - pad] W[k][c][a][b]  (zeros outside x), output Ho x Wo.
 * Forward of the network's padding-0 convolutions: pad = 0, Ho = Hi - 2.  Backward-data: x = d y, pad = 2, Ho = Hi + 2,
 * records packed with backward != 0.  x, y: element strides (image, channel, row), unit column stride. */
template <int KB, int TB>
static void wn_launch(const float* records, const float* x, WnTensor xi, int Hi, int Wi, int pad, float* y, WnTensor yo, int Ho, int Wo,
                      int n_img, int c_in, int c_out, int groups, hipStream_t st) {
    dim3 grid((unsigned)((Wo + 15) / 16), (unsigned)((Ho + 8 * TB - 1) / (8 * TB)), (unsigned)(n_img * groups));
    hipLaunchKernelGGL((k_wino3x3<KB, TB>), grid, dim3(256), 0, st, records, x, xi, Hi, Wi, pad, y, yo, Ho, Wo, c_in, c_out, groups);
}

extern "C" int nf_conv3x3_wino(const float* records, int k_per_group, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h, int Hi,
                               int Wi, int pad, float* y, int64_t ys_n, int64_t ys_c, int64_t ys_h, int Ho, int Wo, int n_img,
                               int c_in, int c_out, int tile_blocks, nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && c_in >= 1 && c_out >= 1 && Hi >= 1 && Wi >= 1 && Ho >= 1 && Wo >= 1 && (k_per_group == 64 || k_per_group == 32),
               "nf_conv3x3_wino: bad arguments (k_per_group %d)", k_per_group);
    const int groups = (c_out + k_per_group - 1) / k_per_group;
    const WnTensor xi = {xs_n, xs_c, xs_h}, yo = {ys_n, ys_c, ys_h};
    hipStream_t st = (hipStream_t)stream;
    // measured (tools/bench_conv3x3.py): one tile block per wave at two workgroups per CU beats two blocks at one workgroup per
    // CU on every ResUNet shape (100 vs 130 us at 64 -> 64, 189 x 252), so that is the default
    const int tb = tile_blocks == 2 ? 2 : 1;
#define WN_GO(KB)                                                                                                                \
    do {                                                                                                                          \
        if (tb == 1) wn_launch<KB, 1>(records, x, xi, Hi, Wi, pad, y, yo, Ho, Wo, n_img, c_in, c_out, groups, st);               \
        else wn_launch<KB, 2>(records, x, xi, Hi, Wi, pad, y, yo, Ho, Wo, n_img, c_in, c_out, groups, st);                       \
    } while (0)
    if (k_per_group == 32) WN_GO(1);
    else WN_GO(2);
#undef WN_GO
    NF_LAUNCH_CHECK("nf_conv3x3_wino");
    return 0;
}
