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
// 2.25x fewer multiplies than the direct form.  The weight records run through a four-deep register ring (three steps in
// flight); the next chunk's input window is fetched into registers half a chunk ahead and committed to the second LDS buffer
// behind the current chunk's multiplications (one barrier per chunk); two workgroups per CU.
// Measured and dropped (tools/bench_conv3x3.py, 64 -> 64 at 189 x 252: 100 us for this form): two tile blocks per wave at one
// workgroup per CU (130 us), B operands loaded straight from global memory without LDS (152 us), deeper weight ring (no change).
#include "nf_common.h"

typedef float w16 __attribute__((ext_vector_type(16)));
typedef float w2f __attribute__((ext_vector_type(2), aligned(4)));
#define WN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__host__ __device__ constexpr int wn_nidx(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

#define WN_CC 16            // input channels per chunk
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

template <int KB>
__global__ void __launch_bounds__(256, 2) k_wino3x3(const float* __restrict__ rec, const float* __restrict__ x, WnTensor xi, int Hi, int Wi,
                                                    int pad, float* __restrict__ y, WnTensor yo, int Ho, int Wo, int C, int K, int groups) {
    // two buffers of the raw input window of a 16-channel chunk; after the last chunk the same memory carries the row half of
    // the output transform from wave to wave
    constexpr int WN_PR = 10, WN_CH = WN_PR * WN_PS, WN_BUF = WN_CC * WN_CH;
    __shared__ float smem[(KB * 8192 > 2 * WN_BUF) ? KB * 8192 : 2 * WN_BUF];
    float* ex = smem;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = lane & 31, hh = lane >> 5, tr = t >> 3, tc = t & 7;
    const int n = blockIdx.z / groups, grp = blockIdx.z - n * groups;
    const int oy0 = blockIdx.y * 8, ox0 = blockIdx.x * 16;
    const int iy0 = oy0 - pad, ix0 = ox0 - pad;
    const int chunks = (C + WN_CC - 1) / WN_CC;
    const float* xn = x + n * xi.ns;
    // this wave's weight stream: [chunk][step][nu][kb] records, consumed strictly in order
    const float* wr = rec + ((size_t)(grp * 4 + w) * chunks) * (8 * 4 * KB * 64) + lane;
    // B^T d B, row w: which two window rows, and the sign of the second
    const int ra = w == 0 ? 0 : (w == 2 ? 2 : 1), rb = w == 0 ? 2 : (w == 1 ? 2 : (w == 2 ? 1 : 3));
    const float sb = w == 1 ? 1.f : -1.f;
    const int lbase = (2 * tr) * WN_PS + 2 * tc;

    // staging: the 160 rows (16 channels x 10) of a chunk's window are cut into 480 segments of 6 floats; thread j owns segments
    // j and j + 256: three 8-byte loads each when the segment lies inside the image, element-wise with zero fill at the border.
    // Fetched into registers half a chunk ahead, committed to the OTHER buffer behind the chunk's multiplications.
    const int seg_a = threadIdx.x, seg_b = threadIdx.x + 256;
    const bool has_b = seg_b < WN_CC * WN_PR * 3;
    float pre[2][6];
    auto fetch_seg = [&](int chunk, int sg, float (&dst)[6]) {
        const int j = sg / 3, part = sg - 3 * j;
        const int ch = j / WN_PR, pr = j - ch * WN_PR;
        const int c = chunk * WN_CC + ch, gy = iy0 + pr, gx0 = ix0 + 6 * part;
        const bool row_ok = c < C && gy >= 0 && gy < Hi;
        const float* src = xn + (row_ok ? c : 0) * xi.cs + (row_ok ? gy : 0) * xi.rs;
        if (row_ok && gx0 >= 0 && gx0 + 6 <= Wi) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const w2f v = *reinterpret_cast<const w2f*>(src + gx0 + 2 * q);
                dst[2 * q] = v[0];
                dst[2 * q + 1] = v[1];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int gx = gx0 + q;
                const bool ok = row_ok && gx >= 0 && gx < Wi;
                const float v = src[ok ? gx : 0];
                dst[q] = ok ? v : 0.f;
            }
        }
    };
    auto commit_seg = [&](int buf, int sg, const float (&src)[6]) {
        const int j = sg / 3, part = sg - 3 * j;
        const int ch = j / WN_PR, pr = j - ch * WN_PR;
        float* dst = smem + buf * WN_BUF + ch * WN_CH + pr * WN_PS + 6 * part;
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<w2f*>(dst + 2 * q) = w2f{src[2 * q], src[2 * q + 1]};
    };
    auto fetch = [&](int chunk) {
        fetch_seg(chunk, seg_a, pre[0]);
        if (has_b) fetch_seg(chunk, seg_b, pre[1]);
    };
    auto commit = [&](int buf) {
        commit_seg(buf, seg_a, pre[0]);
        if (has_b) commit_seg(buf, seg_b, pre[1]);
    };

    w16 acc[4][KB];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nu][kb][r] = 0.f;

    // weight ring: four steps in registers, three in flight (each step = 4 KB records feeding 4 KB MFMAs)
    float wb[4][4 * KB];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 4 * KB; ++i) wb[q][i] = wr[(q * 4 * KB + i) * 64];
    fetch(0);
    commit(0);
    __syncthreads();
    for (int chunk = 0; chunk < chunks; ++chunk) {
        const float* wc = wr + (size_t)chunk * (8 * 4 * KB * 64);
        const float* pbuf = smem + (chunk & 1) * WN_BUF + lbase;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            // step s + 3 (possibly in the next chunk, or the zero pad behind the last one) goes into the slot step s - 1 freed
            asm volatile("" ::: "memory");      // keep the issue order of the loads: the vm counter retires them in order
#pragma unroll
            for (int i = 0; i < 4 * KB; ++i) wb[(s + 3) & 3][i] = wc[((s + 3) * 4 * KB + i) * 64];
            if (s == 3 && chunk + 1 < chunks) fetch(chunk + 1);      // needed four steps from now
            asm volatile("" ::: "memory");
            const float* pa = pbuf + (2 * s + hh) * WN_CH;
            const w2f a0 = *reinterpret_cast<const w2f*>(pa + ra * WN_PS), a1 = *reinterpret_cast<const w2f*>(pa + ra * WN_PS + 2);
            const w2f b0 = *reinterpret_cast<const w2f*>(pa + rb * WN_PS), b1 = *reinterpret_cast<const w2f*>(pa + rb * WN_PS + 2);
            const float e0 = a0[0] + sb * b0[0], e1 = a0[1] + sb * b0[1], e2 = a1[0] + sb * b1[0], e3 = a1[1] + sb * b1[1];
            const float v[4] = {e0 - e2, e1 + e2, e2 - e1, e1 - e3};
#pragma unroll
            for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) acc[nu][kb] = WN_MFMA(wb[s & 3][nu * KB + kb], v[nu], acc[nu][kb]);
        }
        // the other buffer was last read one chunk ago (every wave has passed the barrier behind it): fill it, then one barrier
        // both publishes it and retires this chunk's buffer
        if (chunk + 1 < chunks) commit((chunk + 1) & 1);
        __syncthreads();
    }
    // ---- output transform Y = A^T M A: columns (nu) inside the wave, rows (w) across the waves through LDS (all KB at once)
    const int kbase = grp * (32 * KB);
    float* yn = y + n * yo.ns;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a0 = acc[0][kb][r], a1 = acc[1][kb][r], a2 = acc[2][kb][r], a3 = acc[3][kb][r];
            ex[kb * 8192 + ((w * 2 + 0) * 16 + r) * 64 + lane] = a0 + a1 + a2;
            ex[kb * 8192 + ((w * 2 + 1) * 16 + r) * 64 + lane] = a1 - a2 - a3;
        }
    __syncthreads();
    {
        const int oy = w & 1, rsel = 8 * (w >> 1);
        const int row = oy0 + 2 * tr + oy, col = ox0 + 2 * tc;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = rsel + q;
                float o[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float* e = ex + kb * 8192 + (j * 16 + r) * 64 + lane;
                    const float m0 = e[0 * 2048], m1 = e[1 * 2048], m2 = e[2 * 2048], m3 = e[3 * 2048];
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
}

/* y[n, k, oy, ox] = sum_c sum_{a,b<3} x[n, c, oy + a - pad, ox + b - pad] W[k][c][a][b]  (zeros outside x), output Ho x Wo.
 * Forward of the network's padding-0 convolutions: pad = 0, Ho = Hi - 2.  Backward-data: x = d y, pad = 2, Ho = Hi + 2,
 * records packed with backward != 0.  x, y: element strides (image, channel, row), unit column stride. */
template <int KB>
static void wn_launch(const float* records, const float* x, WnTensor xi, int Hi, int Wi, int pad, float* y, WnTensor yo, int Ho, int Wo,
                      int n_img, int c_in, int c_out, int groups, hipStream_t st) {
    dim3 grid((unsigned)((Wo + 15) / 16), (unsigned)((Ho + 7) / 8), (unsigned)(n_img * groups));
    hipLaunchKernelGGL((k_wino3x3<KB>), grid, dim3(256), 0, st, records, x, xi, Hi, Wi, pad, y, yo, Ho, Wo, c_in, c_out, groups);
}

extern "C" int nf_conv3x3_wino(const float* records, int k_per_group, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h, int Hi,
                               int Wi, int pad, float* y, int64_t ys_n, int64_t ys_c, int64_t ys_h, int Ho, int Wo, int n_img,
                               int c_in, int c_out, int tile_blocks, nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && c_in >= 1 && c_out >= 1 && Hi >= 1 && Wi >= 1 && Ho >= 1 && Wo >= 1 && (k_per_group == 64 || k_per_group == 32),
               "nf_conv3x3_wino: bad arguments (k_per_group %d)", k_per_group);
    const int groups = (c_out + k_per_group - 1) / k_per_group;
    const WnTensor xi = {xs_n, xs_c, xs_h}, yo = {ys_n, ys_c, ys_h};
    hipStream_t st = (hipStream_t)stream;
    (void)tile_blocks;      // one block of 32 tiles per wave (two blocks at one workgroup per CU measured 30 % slower)
    if (k_per_group == 32) wn_launch<1>(records, x, xi, Hi, Wi, pad, y, yo, Ho, Wo, n_img, c_in, c_out, groups, st);
    else wn_launch<2>(records, x, xi, Hi, Wi, pad, y, yo, Ho, Wo, n_img, c_in, c_out, groups, st);
    NF_LAUNCH_CHECK("nf_conv3x3_wino");
    return 0;
}
