// 1x1 convolutions of the ResUNet (a14) as fp32 MFMA GEMMs over the pixels, with the data movement the library path
// needs around them folded into the addressing: the stride-2 subsampling of the downsample branches is read in place, and
// out_conv writes the channels-last feature maps the gather kernels want directly (the vendor path runs
// transpose -> GEMM -> transpose plus a layout copy per call: ~0.55 ms per PGD step for four tiny GEMMs).
// ref: ibrnet/feature_network.py:196-203 (out_conv 1x1 with bias), :62-70 (downsample: conv1x1(stride) + norm).
//
// Y[co][p] = bias[co] + sum_ci W[co][ci] X[ci][p]: A operand = weights as host-packed records (rows = 32 output channels,
// k = input channels in fragment order), B operand = the pixel's channels (lane = pixel).  Input and output are addressed as
// base + n * ns + c * cs + row * rs + col * ws (elements), so NCHW, NHWC and strided views are all the same kernel.
// The backward-data pass is the same kernel on the transposed weight records.
#include "nf_common.h"

typedef float c16 __attribute__((ext_vector_type(16)));
typedef float c1_f4u __attribute__((ext_vector_type(4), aligned(4)));
#define C1_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__host__ __device__ constexpr int c1_nidx(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

extern "C" int64_t nf_conv1x1_pack_floats(int c_out, int c_in) { return (int64_t)((c_out + 31) / 32) * ((c_in + 31) / 32) * 16 * 64; }

/* HOST: weight [c_out][c_in] (transposed != 0: the matrix used is weight^T, i.e. the backward-data GEMM) -> records */
extern "C" int nf_conv1x1_pack(const float* weight, int c_out, int c_in, int transposed, float* out) {
    const int N = transposed ? c_in : c_out, K = transposed ? c_out : c_in;          // GEMM rows / reduction length
    float* rec = out;
    for (int nt = 0; nt < (N + 31) / 32; ++nt)
        for (int kt = 0; kt < (K + 31) / 32; ++kt)
            for (int r = 0; r < 16; ++r, rec += 64)
                for (int lane = 0; lane < 64; ++lane) {
                    const int n = nt * 32 + (lane & 31), k = kt * 32 + c1_nidx(r, lane >> 5);
                    float v = 0.f;
                    if (n < N && k < K) v = transposed ? weight[(size_t)k * c_in + n] : weight[(size_t)n * c_in + k];
                    rec[lane] = v;
                }
    return 0;
}

struct C1Tensor { int64_t ns, cs, rs, ws; };      // element strides: image, channel, row, column

// one wave = 32 pixels; KT = input-channel tiles of 32 (compile time: the pixel's channels live in registers)
template <int KT>
__global__ void __launch_bounds__(256) k_conv1x1(const float* __restrict__ rec, const float* __restrict__ bias, const float* __restrict__ x,
                                                 C1Tensor xi, float* __restrict__ y, C1Tensor yo, int n_img, int H, int W, int c_in,
                                                 int c_out, const float* __restrict__ x2, int c_split) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int64_t HW = (int64_t)H * W, n_pix = (int64_t)n_img * HW;
    const int64_t n_tiles = (n_pix + 31) / 32;
    const int nt_out = (c_out + 31) / 32;
    // gridDim.y splits the output-channel tiles over workgroups (small pixel grids: layer3 has 378 pixel tiles, the chip
    // 1024 SIMDs); a workgroup handles the tiles t = blockIdx.y, blockIdx.y + gridDim.y, ...
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        int64_t p = tile * 32 + m;
        const bool live = p < n_pix;
        if (!live) p = n_pix - 1;
        const int n = (int)(p / HW);
        const int q = (int)(p - (int64_t)n * HW);
        const int row = q / W, col = q - row * W;
        // wave-uniform channel offsets + ONE per-lane offset (pixel position and the lane half's 4-channel shift)
        const int64_t xoff = n * xi.ns + row * xi.rs + col * xi.ws + 4 * h * xi.cs;
        const float* xl = x + xoff;
        // second source: channels c_split .. c_in - 1 live in x2 (same strides) -- the two feature-map gradients of out_conv's
        // backward arrive as separate tensors; c_split is a multiple of 32, so a k-tile reads one source
        const float* xl2 = x2 ? x2 + xoff - (int64_t)c_split * xi.cs : xl;
        float* yl = y + n * yo.ns + row * yo.rs + col * yo.ws + 4 * h * yo.cs;
        c16 xv[KT];
        if (xi.cs == 1 && (c_in & 7) == 0) {        // channels-last input: registers r..r+3 are 4 consecutive channels
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = 32 * kt + 8 * g;
                    const float* xs = c >= c_split ? xl2 : xl;
                    c1_f4u v = (c + 4 * h < c_in) ? *reinterpret_cast<const c1_f4u*>(xs + c) : c1_f4u{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) xv[kt][4 * g + j] = v[j];
                }
        } else {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = 32 * kt + c1_nidx(r, 0);
                    const float* xs = c >= c_split ? xl2 : xl;
                    xv[kt][r] = (c + 4 * h < c_in) ? xs[c * xi.cs] : 0.f;
                }
        }
        for (int t = blockIdx.y; t < nt_out; t += gridDim.y) {
            c16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * t + c1_nidx(r, h);
                acc[r] = (bias && co < c_out) ? bias[co] : 0.f;
            }
            const float* rt = rec + (size_t)t * KT * 16 * 64;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc = C1_MFMA(rt[(kt * 16 + r) * 64 + lane], xv[kt][r], acc);
            if (live) {
                if (yo.cs == 1 && (c_out & 7) == 0) {      // channels-last output: 16-byte stores of 4 consecutive channels
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = 32 * t + 8 * g;
                        if (c + 4 * h < c_out)
                            *reinterpret_cast<c1_f4u*>(yl + c) = c1_f4u{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = 32 * t + c1_nidx(r, 0);
                        if (c + 4 * h < c_out) yl[c * yo.cs] = acc[r];
                    }
                }
            }
        }
    }
}

/* y[n, co, row, col] = bias[co] + sum_ci M[co][ci] x[n, ci, row, col] over an H x W pixel grid; x and y are addressed with
 * element strides (image, channel, row, column), so a stride-2 subsampled view, NCHW and NHWC are the same call.
 * records: nf_conv1x1_pack of the weight (transposed for the backward-data pass, where c_in / c_out swap roles).
 * x2 (nullable): input channels c_split .. c_in - 1 are read from x2[n, ci - c_split, row, col] (same strides as x). */
extern "C" int nf_conv1x1(const float* records, const float* bias, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h,
                          int64_t xs_w, float* y, int64_t ys_n, int64_t ys_c, int64_t ys_h, int64_t ys_w, int n_img, int H, int W,
                          int c_in, int c_out, const float* x2, int c_split, nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && H >= 1 && W >= 1 && c_in >= 1 && c_in <= 256 && c_out >= 1 && c_out <= 1024,
               "nf_conv1x1: need 1 <= c_in <= 256, 1 <= c_out <= 1024 (got c_in %d c_out %d)", c_in, c_out);
    NF_REQUIRE(!x2 || (c_split > 0 && c_split < c_in && c_split % 32 == 0),
               "nf_conv1x1: a second source needs 0 < c_split < c_in, c_split a multiple of 32 (got %d of %d)", c_split, c_in);
    if (!x2) c_split = c_in;
    const int64_t tiles = ((int64_t)n_img * H * W + 31) / 32;
    int64_t blocks = (tiles + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    const int nt_out = (c_out + 31) / 32;
    int split = 1;                          // output tiles over gridDim.y until ~2 waves per SIMD are in flight
    while (split < nt_out && blocks * split * 4 < 2048) split *= 2;
    if (split > nt_out) split = nt_out;
    const C1Tensor xi = {xs_n, xs_c, xs_h, xs_w}, yo = {ys_n, ys_c, ys_h, ys_w};
    hipStream_t st = (hipStream_t)stream;
    const int kt = (c_in + 31) / 32;
#define C1_LAUNCH(KT) \
    hipLaunchKernelGGL(k_conv1x1<KT>, dim3((unsigned)blocks, (unsigned)split), dim3(256), 0, st, records, bias, x, xi, y, yo, n_img, H, W, c_in, c_out, x2, c_split)
    switch (kt) {
        case 1: C1_LAUNCH(1); break;
        case 2: C1_LAUNCH(2); break;
        case 3: C1_LAUNCH(3); break;
        case 4: C1_LAUNCH(4); break;
        case 5: C1_LAUNCH(5); break;
        case 6: C1_LAUNCH(6); break;
        case 7: C1_LAUNCH(7); break;
        default: C1_LAUNCH(8); break;
    }
#undef C1_LAUNCH
    NF_LAUNCH_CHECK("nf_conv1x1");
    return 0;
}
