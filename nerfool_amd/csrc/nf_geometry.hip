// Kernels for ray sampling (a2) and the projector (a3): sample_along_camera_ray, camera table, fused
// project + bilinear gather + direction deltas + validity mask, and the scatter-add backward of the feature gather.
// ref: ibrnet/render_ray.py:73-116, ibrnet/projection.py:89-132.
//
// Data layout: a (point, view) pair is served by 8 adjacent lanes; with a channels-last feature map each lane reads
// one float4 per tap so the 8 lanes cover one 128-byte pixel record (32 channels) -- one cache line per tap.
#include "nf_geometry.h"

#include <stdarg.h>
#include <stdio.h>

// ---------------------------------------------------------------------------------------------------------------
// error plumbing shared by all translation units
// ---------------------------------------------------------------------------------------------------------------
static thread_local char g_nf_error[512] = "";

void nf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_nf_error, sizeof(g_nf_error), fmt, ap);
    va_end(ap);
}

extern "C" int nf_abi_version(void) { return NF_ABI_VERSION; }
extern "C" const char* nf_last_error(void) { return g_nf_error; }
extern "C" int nf_device_cu_count(void) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return cus;
}

// ---------------------------------------------------------------------------------------------------------------
// a2
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_sample_along_ray(const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                          const float* __restrict__ depth_range, int64_t n_rays, int S,
                                                          int inv_uniform, const float* __restrict__ t_rand,
                                                          float* __restrict__ pts, float* __restrict__ z_vals) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays * S) return;
    int64_t r = i / S;
    int s = (int)(i - r * S);
    float near = depth_range[0], far = depth_range[1];
    float z = nf_coarse_depth(near, far, s, S, inv_uniform);
    if (t_rand) {   // stratified jitter, ibrnet/render_ray.py:103-110
        float zp = nf_coarse_depth(near, far, s > 0 ? s - 1 : 0, S, inv_uniform);
        float zn = nf_coarse_depth(near, far, s < S - 1 ? s + 1 : S - 1, S, inv_uniform);
        float lower = s > 0 ? 0.5f * (z + zp) : z;
        float upper = s < S - 1 ? 0.5f * (zn + z) : z;
        z = lower + (upper - lower) * t_rand[i];
    }
    z_vals[i] = z;
    pts[i * 3 + 0] = z * ray_d[r * 3 + 0] + ray_o[r * 3 + 0];
    pts[i * 3 + 1] = z * ray_d[r * 3 + 1] + ray_o[r * 3 + 1];
    pts[i * 3 + 2] = z * ray_d[r * 3 + 2] + ray_o[r * 3 + 2];
}

__global__ void __launch_bounds__(256) k_points_from_depths(const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                            const float* __restrict__ z_vals, int64_t n_rays, int S,
                                                            float* __restrict__ pts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays * S) return;
    int64_t r = i / S;
    float z = z_vals[i];
    pts[i * 3 + 0] = z * ray_d[r * 3 + 0] + ray_o[r * 3 + 0];
    pts[i * 3 + 1] = z * ray_d[r * 3 + 1] + ray_o[r * 3 + 1];
    pts[i * 3 + 2] = z * ray_d[r * 3 + 2] + ray_o[r * 3 + 2];
}

extern "C" int nf_sample_along_ray(const float* ray_o, const float* ray_d, const float* depth_range, int64_t n_rays,
                                   int n_samples, int inv_uniform, const float* t_rand, float* pts, float* z_vals,
                                   nf_stream_t stream) {
    NF_REQUIRE(n_rays >= 0 && n_samples >= 2, "nf_sample_along_ray: need n_rays >= 0 and n_samples >= 2 (got %lld, %d)",
               (long long)n_rays, n_samples);
    if (n_rays == 0) return 0;
    hipLaunchKernelGGL(k_sample_along_ray, dim3(nf_blocks(n_rays * n_samples, 256)), dim3(256), 0, (hipStream_t)stream,
                       ray_o, ray_d, depth_range, n_rays, n_samples, inv_uniform, t_rand, pts, z_vals);
    NF_LAUNCH_CHECK("nf_sample_along_ray");
    return 0;
}

extern "C" int nf_points_from_depths(const float* ray_o, const float* ray_d, const float* z_vals, int64_t n_rays,
                                     int n_samples, float* pts, nf_stream_t stream) {
    NF_REQUIRE(n_rays >= 0 && n_samples >= 1, "nf_points_from_depths: bad sizes");
    if (n_rays == 0) return 0;
    hipLaunchKernelGGL(k_points_from_depths, dim3(nf_blocks(n_rays * n_samples, 256)), dim3(256), 0, (hipStream_t)stream,
                       ray_o, ray_d, z_vals, n_rays, n_samples, pts);
    NF_LAUNCH_CHECK("nf_points_from_depths");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// a3: camera table
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_camera_setup(const float* __restrict__ query_camera, const float* __restrict__ src_cameras, int V,
                               float* __restrict__ cam_ws) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < V) {
        nf_camera_entry(src_cameras + (int64_t)v * 34, cam_ws + (int64_t)v * NF_CAM_STRIDE);
    } else if (v == V) {
        float* q = cam_ws + (int64_t)V * NF_CAM_STRIDE;
        for (int i = 0; i < NF_CAM_STRIDE; ++i) q[i] = 0.f;
        q[0] = src_cameras[0];   // h, w of the SOURCE images: train_cameras[0][:2] (ibrnet/projection.py:112)
        q[1] = src_cameras[1];
        q[12] = query_camera[18 + 3];
        q[13] = query_camera[18 + 7];
        q[14] = query_camera[18 + 11];
    }
}

extern "C" int nf_camera_setup(const float* query_camera, const float* src_cameras, int n_views, float* cam_ws,
                               nf_stream_t stream) {
    NF_REQUIRE(n_views >= 1 && n_views <= 64, "nf_camera_setup: n_views must be in [1,64] (got %d)", n_views);
    hipLaunchKernelGGL(k_camera_setup, dim3(1), dim3(128), 0, (hipStream_t)stream, query_camera, src_cameras, n_views,
                       cam_ws);
    NF_LAUNCH_CHECK("nf_camera_setup");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// a3: fused projection + gather (forward)
// ---------------------------------------------------------------------------------------------------------------
#define NF_SUB 8   // lanes cooperating on one (point, view) pair

template <bool VEC>
__global__ void __launch_bounds__(256) k_project_gather_fwd(const float* __restrict__ xyz, int64_t n_pts,
                                                            const float* __restrict__ cam_ws, int V,
                                                            const float* __restrict__ src_rgbs, int H, int W,
                                                            const float* __restrict__ featmap, int C, int Hf, int Wf,
                                                            int64_t fs_v, int64_t fs_c, int64_t fs_h, int64_t fs_w,
                                                            float* __restrict__ rgb_feat, float* __restrict__ ray_diff,
                                                            float* __restrict__ mask, float* __restrict__ pix) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t pv = gid / NF_SUB;
    int sub = (int)(gid - pv * NF_SUB);
    if (pv >= n_pts * V) return;
    int64_t n = pv / V;
    int v = (int)(pv - n * V);
    const float* cam = cam_ws + (int64_t)v * NF_CAM_STRIDE;
    const float* qc = cam_ws + (int64_t)V * NF_CAM_STRIDE;
    float h = qc[0], w = qc[1];
    float x = xyz[n * 3 + 0], y = xyz[n * 3 + 1], z = xyz[n * 3 + 2];
    float px, py;
    bool front;
    nf_project_point(cam, x, y, z, px, py, front);
    float* out = rgb_feat + pv * (int64_t)(3 + C);

    NfTaps tf = nf_bilinear_taps(px, py, h, w, Hf, Wf);
    const float* fbase = featmap + (int64_t)v * fs_v;
    if (VEC) {
        for (int q = sub; q < C / 4; q += NF_SUB) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (tf.in[t]) {
                    int xx = tf.x0 + (t & 1), yy = tf.y0 + (t >> 1);
                    float4 val = *reinterpret_cast<const float4*>(fbase + (int64_t)yy * fs_h + (int64_t)xx * fs_w + q * 4);
                    acc.x = acc.x + val.x * tf.w[t];
                    acc.y = acc.y + val.y * tf.w[t];
                    acc.z = acc.z + val.z * tf.w[t];
                    acc.w = acc.w + val.w * tf.w[t];
                }
            }
            out[3 + q * 4 + 0] = acc.x;
            out[3 + q * 4 + 1] = acc.y;
            out[3 + q * 4 + 2] = acc.z;
            out[3 + q * 4 + 3] = acc.w;
        }
    } else {
        for (int c = sub; c < C; c += NF_SUB) {
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (tf.in[t]) {
                    int xx = tf.x0 + (t & 1), yy = tf.y0 + (t >> 1);
                    acc = acc + fbase[(int64_t)c * fs_c + (int64_t)yy * fs_h + (int64_t)xx * fs_w] * tf.w[t];
                }
            }
            out[3 + c] = acc;
        }
    }
    if (sub == 0) {
        NfTaps ti = nf_bilinear_taps(px, py, h, w, H, W);
        const float* ibase = src_rgbs + (int64_t)v * H * W * 3;
        float r = 0.f, g = 0.f, b = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (ti.in[t]) {
                const float* p = ibase + ((int64_t)(ti.y0 + (t >> 1)) * W + (ti.x0 + (t & 1))) * 3;
                r = r + p[0] * ti.w[t];
                g = g + p[1] * ti.w[t];
                b = b + p[2] * ti.w[t];
            }
        }
        out[0] = r; out[1] = g; out[2] = b;
        float rd[4];
        nf_ray_diff(qc + 12, cam + 12, x, y, z, rd);
        float* rdo = ray_diff + pv * 4;
        rdo[0] = rd[0]; rdo[1] = rd[1]; rdo[2] = rd[2]; rdo[3] = rd[3];
        mask[pv] = (nf_inbound(px, py, h, w) && front) ? 1.f : 0.f;
        if (pix) {
            pix[((int64_t)v * n_pts + n) * 2 + 0] = px;
            pix[((int64_t)v * n_pts + n) * 2 + 1] = py;
        }
    }
}

static bool nf_vec_ok(const float* featmap, int C, int64_t fs_v, int64_t fs_c, int64_t fs_h, int64_t fs_w) {
    return fs_c == 1 && (C % 4) == 0 && (fs_v % 4) == 0 && (fs_h % 4) == 0 && (fs_w % 4) == 0 &&
           (((uintptr_t)featmap) % 16) == 0;
}

extern "C" int nf_project_gather_fwd(const float* xyz, int64_t n_pts, const float* cam_ws, int n_views,
                                     const float* src_rgbs, int H, int W, const float* featmap, int C, int Hf, int Wf,
                                     int64_t fs_v, int64_t fs_c, int64_t fs_h, int64_t fs_w, float* rgb_feat,
                                     float* ray_diff, float* mask, float* pix, nf_stream_t stream) {
    NF_REQUIRE(n_pts >= 0 && n_views >= 1 && C >= 1 && H >= 2 && W >= 2 && Hf >= 1 && Wf >= 1,
               "nf_project_gather_fwd: bad sizes (n_pts %lld V %d C %d H %d W %d Hf %d Wf %d)", (long long)n_pts,
               n_views, C, H, W, Hf, Wf);
    if (n_pts == 0) return 0;
    int64_t threads = n_pts * n_views * NF_SUB;
    dim3 grid(nf_blocks(threads, 256)), block(256);
    if (nf_vec_ok(featmap, C, fs_v, fs_c, fs_h, fs_w))
        hipLaunchKernelGGL(k_project_gather_fwd<true>, grid, block, 0, (hipStream_t)stream, xyz, n_pts, cam_ws, n_views,
                           src_rgbs, H, W, featmap, C, Hf, Wf, fs_v, fs_c, fs_h, fs_w, rgb_feat, ray_diff, mask, pix);
    else
        hipLaunchKernelGGL(k_project_gather_fwd<false>, grid, block, 0, (hipStream_t)stream, xyz, n_pts, cam_ws, n_views,
                           src_rgbs, H, W, featmap, C, Hf, Wf, fs_v, fs_c, fs_h, fs_w, rgb_feat, ray_diff, mask, pix);
    NF_LAUNCH_CHECK("nf_project_gather_fwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// a3: backward of the feature gather (scatter-add).  ref: autograd of F.grid_sample at ibrnet/projection.py:120-121
// ---------------------------------------------------------------------------------------------------------------
#define NF_SUB_BWD 32   // lanes per (point, view) in the scatter: lane = channel => 128 contiguous bytes per tap (C = 32)

__global__ void __launch_bounds__(256) k_project_gather_bwd(const float* __restrict__ xyz, int64_t n_pts,
                                                            const float* __restrict__ cam_ws, int V, int H, int W,
                                                            const float* __restrict__ d_rgb_feat, int C, int Hf, int Wf,
                                                            int64_t fs_v, int64_t fs_c, int64_t fs_h, int64_t fs_w,
                                                            float* __restrict__ d_featmap) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t pv = gid / NF_SUB_BWD;
    int sub = (int)(gid - pv * NF_SUB_BWD);
    if (pv >= n_pts * V) return;
    int64_t n = pv / V;
    int v = (int)(pv - n * V);
    const float* cam = cam_ws + (int64_t)v * NF_CAM_STRIDE;
    const float* qc = cam_ws + (int64_t)V * NF_CAM_STRIDE;
    float px, py;
    bool front;
    nf_project_point(cam, xyz[n * 3 + 0], xyz[n * 3 + 1], xyz[n * 3 + 2], px, py, front);
    NfTaps tf = nf_bilinear_taps(px, py, qc[0], qc[1], Hf, Wf);
    if (!(tf.in[0] || tf.in[1] || tf.in[2] || tf.in[3])) return;
    const float* g = d_rgb_feat + pv * (int64_t)(3 + C) + 3;
    float* fbase = d_featmap + (int64_t)v * fs_v;
    for (int c = sub; c < C; c += NF_SUB_BWD) {
        float gv = g[c];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (tf.in[t]) {
                int xx = tf.x0 + (t & 1), yy = tf.y0 + (t >> 1);
                atomicAdd(fbase + (int64_t)c * fs_c + (int64_t)yy * fs_h + (int64_t)xx * fs_w, gv * tf.w[t]);
            }
        }
    }
}

extern "C" int nf_project_gather_bwd(const float* xyz, int64_t n_pts, const float* cam_ws, int n_views, int H, int W,
                                     const float* d_rgb_feat, int C, int Hf, int Wf, int64_t fs_v, int64_t fs_c,
                                     int64_t fs_h, int64_t fs_w, float* d_featmap, nf_stream_t stream) {
    NF_REQUIRE(n_pts >= 0 && n_views >= 1 && C >= 1 && Hf >= 1 && Wf >= 1, "nf_project_gather_bwd: bad sizes");
    if (n_pts == 0) return 0;
    int64_t threads = n_pts * n_views * NF_SUB_BWD;
    hipLaunchKernelGGL(k_project_gather_bwd, dim3(nf_blocks(threads, 256)), dim3(256), 0, (hipStream_t)stream, xyz, n_pts,
                       cam_ws, n_views, H, W, d_rgb_feat, C, Hf, Wf, fs_v, fs_c, fs_h, fs_w, d_featmap);
    NF_LAUNCH_CHECK("nf_project_gather_bwd");
    return 0;
}

// ---- deterministic form of the scatter (opt-in): keys -> stable sort by feature-map pixel (host side: torch.sort) -> one
// 32-lane group per pixel segment sums its contributions in sorted order.  No atomics: bitwise reproducible run to run.
__global__ void __launch_bounds__(256) k_project_gather_keys(const float* __restrict__ xyz, int64_t n_pts, const float* __restrict__ cam_ws,
                                                             int V, int Hf, int Wf, int* __restrict__ keys, float* __restrict__ wts) {
    int64_t pv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pv >= n_pts * V) return;
    int64_t n = pv / V;
    int v = (int)(pv - n * V);
    const float* cam = cam_ws + (int64_t)v * NF_CAM_STRIDE;
    const float* qc = cam_ws + (int64_t)V * NF_CAM_STRIDE;
    float px, py;
    bool front;
    nf_project_point(cam, xyz[n * 3 + 0], xyz[n * 3 + 1], xyz[n * 3 + 2], px, py, front);
    NfTaps tf = nf_bilinear_taps(px, py, qc[0], qc[1], Hf, Wf);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int xx = tf.x0 + (t & 1), yy = tf.y0 + (t >> 1);
        keys[pv * 4 + t] = tf.in[t] ? (v * Hf + yy) * Wf + xx : 0x7fffffff;      // taps outside the map sort to the end
        wts[pv * 4 + t] = tf.w[t];
    }
}

__global__ void __launch_bounds__(256) k_project_gather_bwd_sorted(const int* __restrict__ skeys, const int64_t* __restrict__ perm,
                                                                   const float* __restrict__ wts, int64_t n_taps,
                                                                   const float* __restrict__ d_rgb_feat, int C, int Hf, int Wf,
                                                                   int64_t fs_v, int64_t fs_c, int64_t fs_h, int64_t fs_w,
                                                                   float* __restrict__ d_featmap) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = gid / NF_SUB_BWD;
    int sub = (int)(gid - i * NF_SUB_BWD);
    if (i >= n_taps) return;
    const int key = skeys[i];
    if (key == 0x7fffffff || (i > 0 && skeys[i - 1] == key)) return;      // not the head of a pixel's segment
    const int xx = key % Wf, yy = (key / Wf) % Hf, v = key / (Wf * Hf);
    for (int c = sub; c < C; c += NF_SUB_BWD) {
        float acc = 0.f;
        for (int64_t j = i; j < n_taps && skeys[j] == key; ++j) {
            const int64_t src = perm[j];                                       // tap index: (point, view) * 4 + tap
            acc = fmaf(wts[src], d_rgb_feat[(src >> 2) * (int64_t)(3 + C) + 3 + c], acc);
        }
        d_featmap[(int64_t)v * fs_v + (int64_t)c * fs_c + (int64_t)yy * fs_h + (int64_t)xx * fs_w] = acc;
    }
}

/* Deterministic variant of nf_project_gather_bwd in two calls around a stable sort the caller performs:
 *   nf_project_gather_keys: keys [n_pts * V * 4] int32 (feature-map pixel index (v * Hf + y) * Wf + x of each bilinear tap,
 *     0x7fffffff for taps outside the map) and weights [n_pts * V * 4]
 *   nf_project_gather_bwd_sorted: sorted keys + the sort's permutation (int64) -> d_featmap (zero-initialised by the caller;
 *     every touched pixel is written once, summed in sorted order) */
extern "C" int nf_project_gather_keys(const float* xyz, int64_t n_pts, const float* cam_ws, int n_views, int Hf, int Wf, int* keys,
                                      float* weights, nf_stream_t stream) {
    NF_REQUIRE(n_pts >= 0 && n_views >= 1 && Hf >= 1 && Wf >= 1 && (int64_t)n_views * Hf * Wf < 0x7fffffff, "nf_project_gather_keys: bad sizes");
    if (n_pts == 0) return 0;
    hipLaunchKernelGGL(k_project_gather_keys, dim3(nf_blocks(n_pts * n_views, 256)), dim3(256), 0, (hipStream_t)stream, xyz, n_pts, cam_ws,
                       n_views, Hf, Wf, keys, weights);
    NF_LAUNCH_CHECK("nf_project_gather_keys");
    return 0;
}

extern "C" int nf_project_gather_bwd_sorted(const int* sorted_keys, const int64_t* perm, const float* weights, int64_t n_taps,
                                            const float* d_rgb_feat, int C, int Hf, int Wf, int64_t fs_v, int64_t fs_c, int64_t fs_h,
                                            int64_t fs_w, float* d_featmap, nf_stream_t stream) {
    NF_REQUIRE(n_taps >= 0 && C >= 1 && Hf >= 1 && Wf >= 1, "nf_project_gather_bwd_sorted: bad sizes");
    if (n_taps == 0) return 0;
    hipLaunchKernelGGL(k_project_gather_bwd_sorted, dim3(nf_blocks(n_taps * NF_SUB_BWD, 256)), dim3(256), 0, (hipStream_t)stream, sorted_keys,
                       perm, weights, n_taps, d_rgb_feat, C, Hf, Wf, fs_v, fs_c, fs_h, fs_w, d_featmap);
    NF_LAUNCH_CHECK("nf_project_gather_bwd_sorted");
    return 0;
}

// pixel_mask[n] = (sum_v mask[n,v]) > 1        ref: ibrnet/render_ray.py:210
__global__ void __launch_bounds__(256) k_pixel_mask(const float* __restrict__ mask, int64_t n_pts, int V,
                                                    uint8_t* __restrict__ out) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_pts) return;
    float s = 0.f;
    for (int v = 0; v < V; ++v) s += mask[n * V + v];
    out[n] = s > 1.f ? 1 : 0;
}

extern "C" int nf_pixel_mask(const float* mask, int64_t n_pts, int n_views, uint8_t* pixel_mask, nf_stream_t stream) {
    NF_REQUIRE(n_pts >= 0 && n_views >= 1, "nf_pixel_mask: bad sizes");
    if (n_pts == 0) return 0;
    hipLaunchKernelGGL(k_pixel_mask, dim3(nf_blocks(n_pts, 256)), dim3(256), 0, (hipStream_t)stream, mask, n_pts, n_views,
                       pixel_mask);
    NF_LAUNCH_CHECK("nf_pixel_mask");
    return 0;
}
