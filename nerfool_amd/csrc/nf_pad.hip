// Layout / padding glue of the ResUNet executor that used to run as ATen kernels (permute + contiguous, F.pad, slicing,
// upsample_bilinear2d_backward): ref ibrnet/feature_network.py:188 (reflect padding of the input image in front of the 7x7 stem),
// :231-243 (skipconnect: zero-pad the encoder tensor up to the decoder size, concatenate), :143-151 (bilinear x2 upsampling in
// front of a reflect-padded 3x3 convolution -- here its backward).
//
//   nf_pad_gather_fwd   out[n][c][ph][pw] = Z(reflect(ph - pad, H), reflect(pw - pad, W)), Z = the H x W plane that holds the
//                       eh x ew source window at (top, left) and zeros elsewhere; the source is addressed by element strides, so
//                       a channels-last image [V,H,W,3] or the interior view of a padded activation is read where it lies
//   nf_pad_gather_bwd   the adjoint: din[n][c][y][x] = sum of dout over the padded positions that read source element (y, x);
//                       din is written through element strides too (the gradient of the channels-last image directly)
//   nf_upsample2x_pad_bwd   adjoint of nf_upsample2x_pad_fwd: folds the reflect padding and applies the transposed bilinear
//                       interpolation in one pass over LDS tiles (weights recomputed with the forward's arithmetic)
// All three are bandwidth kernels: one read of the larger tensor, one write of the smaller.
#include "nf_common.h"

#define PG_BLOCK 256

__device__ __forceinline__ int pg_reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

__global__ void __launch_bounds__(PG_BLOCK) k_pad_gather_fwd(const float* __restrict__ src, int64_t ss_n, int64_t ss_c, int64_t ss_h, int64_t ss_w,
                                                             int eh, int ew, int top, int left, int H, int W, int pad, int C,
                                                             float* __restrict__ out, int64_t os_n, int64_t os_c) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int64_t plane = blockIdx.y;
    const int n = (int)(plane / C), c = (int)(plane - (int64_t)n * C);
    const float* s = src + n * ss_n + c * ss_c;
    float* o = out + n * os_n + c * os_c;
    for (int idx = blockIdx.x * PG_BLOCK + threadIdx.x; idx < Hp * Wp; idx += gridDim.x * PG_BLOCK) {
        const int ph = idx / Wp, pw = idx - ph * Wp;
        const int y = pg_reflect(ph - pad, H) - top, x = pg_reflect(pw - pad, W) - left;
        o[idx] = (y >= 0 && y < eh && x >= 0 && x < ew) ? s[y * ss_h + x * ss_w] : 0.f;
    }
}

// padded positions along one axis that read logical position q (0 <= q < n): q + pad, and the mirror images inside the ring
__device__ __forceinline__ int pg_sources(int q, int n, int pad, int (&p)[3]) {
    int k = 0;
    p[k++] = q + pad;
    if (q >= 1 && q <= pad) p[k++] = pad - q;
    if (q >= n - 1 - pad && q <= n - 2) p[k++] = pad + 2 * (n - 1) - q;
    return k;
}

__global__ void __launch_bounds__(PG_BLOCK) k_pad_gather_bwd(const float* __restrict__ dout, int64_t os_n, int64_t os_c, int H, int W, int pad,
                                                             int C, int eh, int ew, int top, int left, float* __restrict__ din,
                                                             int64_t ds_n, int64_t ds_c, int64_t ds_h, int64_t ds_w) {
    const int Wp = W + 2 * pad;
    const int64_t plane = blockIdx.y;
    const int n = (int)(plane / C), c = (int)(plane - (int64_t)n * C);
    const float* g = dout + n * os_n + c * os_c;
    float* d = din + n * ds_n + c * ds_c;
    for (int idx = blockIdx.x * PG_BLOCK + threadIdx.x; idx < eh * ew; idx += gridDim.x * PG_BLOCK) {
        const int y = idx / ew, x = idx - y * ew;
        int py[3], px[3];
        const int ny = pg_sources(y + top, H, pad, py), nx = pg_sources(x + left, W, pad, px);
        float acc = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) acc += g[(int64_t)py[a] * Wp + px[b]];
        d[y * ds_h + x * ds_w] = acc;
    }
}

extern "C" int nf_pad_gather_fwd(const float* src, int64_t ss_n, int64_t ss_c, int64_t ss_h, int64_t ss_w, int n_img, int C, int eh, int ew,
                                 int top, int left, int H, int W, int pad, float* out, int64_t os_n, int64_t os_c, nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && C >= 1 && eh >= 1 && ew >= 1 && top >= 0 && left >= 0 && top + eh <= H && left + ew <= W && pad >= 0 &&
                   pad < H && pad < W, "nf_pad_gather_fwd: bad arguments (window %d x %d at (%d, %d) in %d x %d, pad %d)", eh, ew, top, left, H, W, pad);
    const int HWp = (H + 2 * pad) * (W + 2 * pad);
    unsigned bx = (unsigned)((HWp + PG_BLOCK * 4 - 1) / (PG_BLOCK * 4));
    hipLaunchKernelGGL(k_pad_gather_fwd, dim3(bx, (unsigned)(n_img * C)), dim3(PG_BLOCK), 0, (hipStream_t)stream, src, ss_n, ss_c, ss_h, ss_w,
                       eh, ew, top, left, H, W, pad, C, out, os_n, os_c);
    NF_LAUNCH_CHECK("nf_pad_gather_fwd");
    return 0;
}

extern "C" int nf_pad_gather_bwd(const float* dout, int64_t os_n, int64_t os_c, int n_img, int C, int H, int W, int pad, int eh, int ew,
                                 int top, int left, float* din, int64_t ds_n, int64_t ds_c, int64_t ds_h, int64_t ds_w, nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && C >= 1 && eh >= 1 && ew >= 1 && top >= 0 && left >= 0 && top + eh <= H && left + ew <= W && pad >= 0 &&
                   pad < H && pad < W, "nf_pad_gather_bwd: bad arguments (window %d x %d at (%d, %d) in %d x %d, pad %d)", eh, ew, top, left, H, W, pad);
    unsigned bx = (unsigned)((eh * ew + PG_BLOCK * 4 - 1) / (PG_BLOCK * 4));
    hipLaunchKernelGGL(k_pad_gather_bwd, dim3(bx, (unsigned)(n_img * C)), dim3(PG_BLOCK), 0, (hipStream_t)stream, dout, os_n, os_c, H, W, pad, C,
                       eh, ew, top, left, din, ds_n, ds_c, ds_h, ds_w);
    NF_LAUNCH_CHECK("nf_pad_gather_bwd");
    return 0;
}

// ---- adjoint of the fused upsample + reflect pad ---------------------------------------------------------------------
// forward (k_upsample2x_pad): yp[ph][pw] = sum_{i,j} wy(Y -> i) wx(X -> j) x[i][j], (Y, X) = reflect(ph - pad, pw - pad) in the
// 2h x 2w plane; wy(Y -> i) = [h1 == i] (1 - l) + [h1 + h1p == i] l with h1 = (int)(rh Y), l = rh Y - h1, h1p = h1 < h - 1.
// Here: dx[i][j] = sum_{Y, X} wy(Y -> i) wx(X -> j) G[Y][X], G = the padded gradient folded onto the 2h x 2w plane.
// A workgroup owns UT x UT source elements; the rows Y with h1 in [i0 - 1, i0 + UT - 1] lie in [2 i0 - 2, 2 (i0 + UT - 1) + 4]
// (rh is just under 1/2), so the folded tile has 2 UT + 5 rows / columns.
// Tile = UT rows x UTX columns of source elements (folded tile 37 x 133: 1.2x the tile's own gradient elements, rows of 532 contiguous
// bytes; the 16 x 16 tile of rounds 2-3 read 1.34x in rows of 148 bytes and moved 1.6x its algorithmic bytes through HBM).  The folded
// tile sits in LDS split by column parity, so that the stride-2 window reads of consecutive lanes hit consecutive banks.
#define UT 16
#define UTX 64
#define UG (2 * UT + 5)
#define UGX (2 * UTX + 5)
#define UGXH ((UGX + 1) / 2 + 1)

__device__ __forceinline__ float up_weight(int Y, int i, int h, int H2, float r) {       // wy(Y -> i); 0 outside the plane
    if (Y < 0 || Y >= H2) return 0.f;
    const float h1r = r * (float)Y;
    const int h1 = (int)h1r;
    const int h1p = h1 < h - 1 ? 1 : 0;
    const float l1 = h1r - (float)h1, l0 = 1.f - l1;
    return (h1 == i ? l0 : 0.f) + (h1 + h1p == i ? l1 : 0.f);
}

__global__ void __launch_bounds__(PG_BLOCK) k_upsample2x_pad_bwd(const float* __restrict__ dyp, int h, int w, int pad, float* __restrict__ dx,
                                                                 int64_t xs_plane, int64_t xs_row, int tiles_x) {
    __shared__ float G[2][UG][UGXH];          // [column parity][row][column / 2]
    const int H2 = 2 * h, W2 = 2 * w, Wp = W2 + 2 * pad, Hp = H2 + 2 * pad;
    const int64_t plane = blockIdx.y;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int i0 = ty * UT, j0 = tx * UTX, Y0 = 2 * i0 - 2, X0 = 2 * j0 - 2;
    const float* g = dyp + plane * (int64_t)Hp * Wp;
    for (int idx = threadIdx.x; idx < UG * UGX; idx += PG_BLOCK) {
        const int a = idx / UGX, b = idx - a * UGX, Y = Y0 + a, X = X0 + b;
        float acc = 0.f;
        if (Y >= 0 && Y < H2 && X >= 0 && X < W2) {
            int py[3], px[3];
            const int ny = pg_sources(Y, H2, pad, py), nx = pg_sources(X, W2, pad, px);
            for (int p = 0; p < ny; ++p)
                for (int q = 0; q < nx; ++q) acc += g[(int64_t)py[p] * Wp + px[q]];
        }
        G[b & 1][a][b >> 1] = acc;
    }
    __syncthreads();
    const float rh = H2 > 1 ? (float)(h - 1) / (float)(H2 - 1) : 0.f;
    const float rw = W2 > 1 ? (float)(w - 1) / (float)(W2 - 1) : 0.f;
    const int lj = threadIdx.x % UTX, lrow = threadIdx.x / UTX;           // 256 threads = 4 rows x 64 columns per pass
    const int j = j0 + lj;
    float wx[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) wx[k] = up_weight(2 * j - 2 + k, j, w, W2, rw);
    static_assert(PG_BLOCK % UTX == 0 && UT % (PG_BLOCK / UTX) == 0, "whole passes over the tile");
#pragma unroll 1
    for (int li = lrow; li < UT; li += PG_BLOCK / UTX) {
        const int i = i0 + li;
        if (i < h && j < w) {
            float acc = 0.f;
#pragma unroll
            for (int a = 0; a < 7; ++a) {
                const float wy = up_weight(2 * i - 2 + a, i, h, H2, rh);
                if (wy != 0.f) {
                    float row = 0.f;
#pragma unroll
                    for (int b = 0; b < 7; ++b) row = fmaf(wx[b], G[b & 1][2 * li + a][lj + (b >> 1)], row);      // column 2 lj + b
                    acc = fmaf(wy, row, acc);
                }
            }
            dx[plane * xs_plane + (int64_t)i * xs_row + j] = acc;
        }
    }
}

/* d(yp) [planes, 2h + 2 pad, 2w + 2 pad] (contiguous) -> dx [planes, h, w] written at (plane stride, row stride): the gradient of
 * nf_upsample2x_pad_fwd's input */
extern "C" int nf_upsample2x_pad_bwd(const float* d_y_padded, int64_t planes, int h, int w, int pad, float* dx, int64_t xs_plane,
                                     int64_t xs_row, nf_stream_t stream) {
    NF_REQUIRE(planes >= 1 && planes <= 0x7fffffff && h >= 1 && w >= 1 && pad >= 0 && pad < 2 * h && pad < 2 * w && xs_row >= w,
               "nf_upsample2x_pad_bwd: bad arguments (planes %lld h %d w %d pad %d)", (long long)planes, h, w, pad);
    const int tiles_x = (w + UTX - 1) / UTX, tiles_y = (h + UT - 1) / UT;
    hipLaunchKernelGGL(k_upsample2x_pad_bwd, dim3((unsigned)(tiles_x * tiles_y), (unsigned)planes), dim3(PG_BLOCK), 0, (hipStream_t)stream,
                       d_y_padded, h, w, pad, dx, xs_plane, xs_row, tiles_x);
    NF_LAUNCH_CHECK("nf_upsample2x_pad_bwd");
    return 0;
}
