/*
 * nerfool_hip.h -- C ABI of libnerfool_hip.so: the MI355X (gfx950) implementation of NeRFool's adversarial inner
 * loop (IBRNet flavour).  Plain pointers and sizes only; every pointer is a DEVICE pointer unless marked HOST.
 * Every call enqueues work on `stream` (a hipStream_t passed as void*) and returns immediately; 0 = ok, any other
 * value = error (text via nf_last_error()).  The library never allocates or frees device memory: outputs and
 * workspaces are provided by the caller (the Python host layer borrows torch storage).
 *
 * Each entry point names the reference function it replaces (paths relative to the GATECH-EIC/NeRFool tree).
 * The reference is pure Python/PyTorch and has no FFI of its own; INTEGRATION.md shows the ctypes stub a
 * maintainer would add.
 */
#ifndef NERFOOL_HIP_H
#define NERFOOL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NF_ABI_VERSION 1

typedef void* nf_stream_t; /* hipStream_t */

int nf_abi_version(void);
const char* nf_last_error(void);
/* number of compute units of the current device (HOST query, used by the host layer to size grids/chunks) */
int nf_device_cu_count(void);

/* ------------------------------------------------------------------------------------------------------------------
 * a1  random pixel pick of one PGD iteration    ref: ibrnet/sample_ray.py:12 (RandomState(234)), :149-171
 *     (`rng.choice(H*W, size=(N_rand,), replace=False)`).  HOST-only, no GPU work, safe to call from any thread.
 * out[0..size) = numpy.random.RandomState.choice(pop, size, replace=False) for the MT19937 state (key[624], *pos) --
 * i.e. permutation(pop)[:size] -- and the state is advanced in place exactly as numpy advances it.
 * All pointers are HOST pointers; scratch holds pop int64.
 * ---------------------------------------------------------------------------------------------------------------- */
int nf_legacy_choice(uint32_t* key, int32_t* pos, int64_t pop, int64_t size, int64_t* out, int64_t* scratch);

/* ------------------------------------------------------------------------------------------------------------------
 * a2  sample_along_camera_ray            ref: ibrnet/render_ray.py:73-116
 * depth_range: device [2] = (near, far).  t_rand: nullable [R,S] uniform numbers (det=False stratified jitter).
 * Writes z_vals [R,S] and pts [R,S,3].
 * ---------------------------------------------------------------------------------------------------------------- */
int nf_sample_along_ray(const float* ray_o, const float* ray_d, const float* depth_range, int64_t n_rays,
                        int n_samples, int inv_uniform, const float* t_rand, float* pts, float* z_vals,
                        nf_stream_t stream);

/* pts[r,s,:] = z[r,s] * ray_d[r] + ray_o[r]        ref: ibrnet/render_ray.py:241-243 */
int nf_points_from_depths(const float* ray_o, const float* ray_d, const float* z_vals, int64_t n_rays,
                          int n_samples, float* pts, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a3  Projector.compute                  ref: ibrnet/projection.py:89-132 (+ :42-62, :64-87, :24-35)
 *
 * nf_camera_setup turns the reference camera vectors ([34] = H, W, K 4x4, c2w 4x4 row-major) into the per-view
 * table the gather kernels read: cam_ws [(V+1) * 16] floats; view v: [0..11] rows 0..2 of K*inv(c2w), [12..14]
 * camera centre; entry V (query camera): [0]=h, [1]=w of the SOURCE images, [12..14] query camera centre.
 * ---------------------------------------------------------------------------------------------------------------- */
int nf_camera_setup(const float* query_camera, const float* src_cameras, int n_views, float* cam_ws,
                    nf_stream_t stream);

/* xyz [N,3] (N = R*S points), src_rgbs [V,H,W,3], featmap: V x C x Hf x Wf addressed through element strides
 * (fs_v, fs_c, fs_h, fs_w) so NCHW and channels-last maps are both accepted without a copy.
 * Outputs: rgb_feat [N,V,3+C], ray_diff [N,V,4], mask [N,V] (0/1 floats); pix (nullable) [V,N,2] pixel locations. */
int nf_project_gather_fwd(const float* xyz, int64_t n_pts, const float* cam_ws, int n_views, const float* src_rgbs,
                          int H, int W, const float* featmap, int C, int Hf, int Wf, int64_t fs_v, int64_t fs_c,
                          int64_t fs_h, int64_t fs_w, float* rgb_feat, float* ray_diff, float* mask, float* pix,
                          nf_stream_t stream);

/* pixel_mask[n] = (sum_v mask[n,v]) > 1   (bytes 0/1)        ref: ibrnet/render_ray.py:210,249 */
int nf_pixel_mask(const float* mask, int64_t n_pts, int n_views, uint8_t* pixel_mask, nf_stream_t stream);

/* Backward of the bilinear FEATURE gather (the rgb taps and ray_diff carry no gradient in the attack):
 * d_featmap[v, c, y, x] += tap weights * d_rgb_feat[n, v, 3 + c]   (float atomics; d_featmap must be pre-zeroed or
 * hold the gradient being accumulated; same stride convention as the forward). */
int nf_project_gather_bwd(const float* xyz, int64_t n_pts, const float* cam_ws, int n_views, int H, int W,
                          const float* d_rgb_feat, int C, int Hf, int Wf, int64_t fs_v, int64_t fs_c, int64_t fs_h,
                          int64_t fs_w, float* d_featmap, nf_stream_t stream);
/* deterministic (sorted / segmented) form of nf_project_gather_bwd: no float atomics, bitwise reproducible.  Two calls around a
 * STABLE ascending sort of `keys` the caller performs (its permutation as int64): keys / weights [n_pts * V * 4] per bilinear tap
 * (key = (v * Hf + y) * Wf + x, 0x7fffffff outside the map); then every feature-map pixel that received a tap is written once
 * with its contributions summed in sorted order.  d_featmap must be zero-initialised. */
int nf_project_gather_keys(const float* xyz, int64_t n_pts, const float* cam_ws, int n_views, int Hf, int Wf, int* keys, float* weights,
                           nf_stream_t stream);
int nf_project_gather_bwd_sorted(const int* sorted_keys, const int64_t* perm, const float* weights, int64_t n_taps,
                                 const float* d_rgb_feat, int C, int Hf, int Wf, int64_t fs_v, int64_t fs_c, int64_t fs_h, int64_t fs_w,
                                 float* d_featmap, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a4/a5  IBRNet.forward (+ MultiHeadAttention)      ref: ibrnet/mlp_network.py:222-274, :69-119, :23-43
 *
 * The 20 136 parameters travel as ONE packed blob; nf_ibrnet_blob_entry enumerates its layout so that the host
 * packs a reference state-dict by key (HOST function, no GPU needed):
 *   idx -> name (state-dict key), offset (floats), rows, cols, transposed (1: stored [in][out]), returns 0 while
 *   idx is valid, 1 past the end.  nf_ibrnet_blob_floats() = total length.
 * pos_enc [S,16] is the `pos_encoding` buffer (depends on S, not part of checkpoints).
 * rgb_feat [R,S,V,35], ray_diff [R,S,V,4], mask [R,S,V] -> raw [R,S,4] (rgb, sigma).
 * workspace: nf_ibrnet_workspace_floats(R,S,V,backward) floats.
 * ---------------------------------------------------------------------------------------------------------------- */
int64_t nf_ibrnet_blob_floats(void);
int nf_ibrnet_blob_entry(int idx, char* name, int name_cap, int64_t* offset, int* rows, int* cols, int* transposed);
int64_t nf_ibrnet_workspace_floats(int64_t n_rays, int n_samples, int n_views, int backward);
int nf_ibrnet_fwd(const float* blob, const float* pos_enc, const float* rgb_feat, const float* ray_diff,
                  const float* mask, int64_t n_rays, int n_samples, int n_views, int anti_alias_pooling,
                  float* raw, float* workspace, nf_stream_t stream);
/* d_raw [R,S,4] -> d_rgb_feat [R,S,V,35] (all 35 channels; the weights are constants of the attack) */
int nf_ibrnet_bwd(const float* blob, const float* pos_enc, const float* rgb_feat, const float* ray_diff,
                  const float* mask, const float* d_raw, int64_t n_rays, int n_samples, int n_views,
                  int anti_alias_pooling, float* d_rgb_feat, float* workspace, nf_stream_t stream);

/* Matrix-core (MFMA, exact fp32) path of the same function for any V <= 32 (a sample's views sit on the next power of two of V
 * adjacent lanes; the reference's default num_source_views = 10 runs on 16 with neutral padding lanes): two kernels per level
 * (per-(sample,view) rows on v_mfma_f32_32x32x2_f32, then per-ray attention).  mfma_blob = the natural blob re-ordered
 * into MFMA operand order by nf_ibrnet_pack_mfma (HOST pointers in, HOST pointers out; no GPU needed).
 * workspace: nf_ibrnet_mfma_workspace_floats(R, S) floats. */
int64_t nf_ibrnet_mfma_blob_floats(void);
int nf_ibrnet_pack_mfma(const float* natural_blob_host, float* mfma_blob_host);
int nf_ibrnet_mfma_supported(int n_samples, int n_views);
/* TEST / DIAGNOSTIC hook.  The per-(sample, view) part of the matrix-core FORWARD exists in two forms: sample-on-the-lane (a wave
 * owns 32 samples and walks their views in time; the view-invariant part of base_fc.0 -- ibrnet/mlp_network.py:245-247 -- is
 * multiplied once per sample; 2 <= V <= 10) and the row form (a wave owns 32 (sample, view) rows; any V <= 32, also the bf16-operand
 * precision of config 5 and every backward).  The sample-on-the-lane kernels multiply on the bf16 matrix cores with every fp32
 * operand split into three bf16 parts (six cross terms of order <= 2^-16: error at fp32 rounding level) or, for comparison, on
 * the fp32 matrix cores.  form 0 = sample-on-the-lane with the split operands where it exists (default), 1 = the row form always,
 * 2 = sample-on-the-lane with fp32 operands (V <= 4).  Returns the previous setting.  Process-wide. */
int nf_ibrnet_rows_form(int form);
/* 1 when the sample-on-the-lane form runs for this view count / operand precision under the current setting */
int nf_ibrnet_sol_selected(int n_views, int bf16_operands);
int64_t nf_ibrnet_mfma_workspace_floats(int64_t n_rays, int n_samples);
int nf_ibrnet_fwd_mfma(const float* mfma_blob, const float* blob, const float* pos_enc, const float* rgb_feat,
                       const float* ray_diff, const float* mask, int64_t n_rays, int n_samples, int n_views,
                       int anti_alias_pooling, float* raw, float* workspace, nf_stream_t stream);
/* backward of the MFMA path: smp = the per-sample records the forward left in its workspace (keep that buffer);
 * d_workspace: same size, scratch.  d_raw [R,S,4] -> d_rgb_feat [R,S,V,35]. */
int nf_ibrnet_bwd_mfma(const float* mfma_blob, const float* blob, const float* pos_enc, const float* rgb_feat,
                       const float* ray_diff, const float* mask, const float* smp, const float* d_raw, int64_t n_rays,
                       int n_samples, int n_views, int anti_alias_pooling, float* d_rgb_feat, float* d_workspace,
                       nf_stream_t stream);
/* The same backward with the adjoint of Projector.compute's feature gather (autograd of F.grid_sample, ibrnet/projection.py:116-125)
 * fused into its output stage: d rgb_feat is not written, its 32 feature channels are scattered straight into the feature-map
 * gradient.  xyz [n_rays * n_samples][3] and cam_ws exactly as given to nf_project_gather_fwd; d_featmap [V][32][Hf][Wf]
 * through element strides, ZEROED by the caller (the kernel adds, float atomics like nf_project_gather_bwd).  bf16_blob: nullptr =
 * exact fp32 rows; else the bf16-operand row network of nf_ibrnet_bwd_mfma_bf16 (8-wave workgroups: image + one staging tile per wave). */
int nf_ibrnet_bwd_mfma_scatter(const float* bf16_blob, const float* mfma_blob, const float* blob, const float* pos_enc, const float* rgb_feat,
                               const float* ray_diff, const float* mask, const float* smp, const float* d_raw, int64_t n_rays,
                               int n_samples, int n_views, int anti_alias_pooling, float* d_workspace, const float* xyz,
                               const float* cam_ws, float* d_featmap, int64_t fs_v, int64_t fs_c, int64_t fs_h, int64_t fs_w,
                               int Hf, int Wf, nf_stream_t stream);
/* The forward with Projector.compute (ibrnet/projection.py:42-132: projection, bilinear taps of the feature maps and of the source
 * images, ray_diff, mask) folded into the row kernel -- the per-ray source features go from the maps to the lanes that consume
 * them, rgb_feat / ray_diff are never written (no-grad rendering).  bf16_blob: nullptr = exact fp32 rows.  featmap channels-last
 * (fs_c == 1, 16-byte aligned pixel records); mask_out [n_rays * n_samples * V] receives the validity mask. */
int nf_ibrnet_fwd_mfma_gather(const float* bf16_blob, const float* mfma_blob, const float* blob, const float* pos_enc, const float* xyz,
                              const float* cam_ws, const float* src_rgbs, int H, int W, const float* featmap, int Hf, int Wf,
                              int64_t fs_v, int64_t fs_c, int64_t fs_h, int64_t fs_w, int64_t n_rays, int n_samples, int n_views,
                              int anti_alias_pooling, float* raw, float* workspace, float* mask_out, nf_stream_t stream);
/* bf16-operand variant of the matrix-core path (BASELINE config 5, "bf16 MFMA path"): the per-(sample, view) row network of
 * IBRNet.forward (ibrnet/mlp_network.py:231-257, 268-273: ray_dir_fc, base_fc, vis_fc, vis_fc2, rgb_fc) runs on
 * v_mfma_f32_32x32x16_bf16 -- weights and activations rounded to bf16 at the matrix-core inputs, fp32 accumulation; pooling,
 * ELU / sigmoid / softmax and the per-ray part (geometry_fc, ray attention, LayerNorm, density head) stay fp32.
 * bf16_blob = nf_ibrnet_pack_mfma_bf16(mfma blob) (HOST pointers; nf_ibrnet_mfma_bf16_blob_floats() floats), copied to the
 * device by the caller.  Same arguments, workspaces and outputs as nf_ibrnet_fwd_mfma / nf_ibrnet_bwd_mfma. */
int64_t nf_ibrnet_mfma_bf16_blob_floats(void);
int nf_ibrnet_pack_mfma_bf16(const float* mfma_blob_host, float* bf16_blob_host);
int nf_ibrnet_fwd_mfma_bf16(const float* bf16_blob, const float* mfma_blob, const float* blob, const float* pos_enc,
                            const float* rgb_feat, const float* ray_diff, const float* mask, int64_t n_rays, int n_samples,
                            int n_views, int anti_alias_pooling, float* raw, float* workspace, nf_stream_t stream);
int nf_ibrnet_bwd_mfma_bf16(const float* bf16_blob, const float* mfma_blob, const float* blob, const float* pos_enc,
                            const float* rgb_feat, const float* ray_diff, const float* mask, const float* smp,
                            const float* d_raw, int64_t n_rays, int n_samples, int n_views, int anti_alias_pooling,
                            float* d_rgb_feat, float* d_workspace, nf_stream_t stream);
/* diagnostics: d[lane][16] = mfma_f32_32x32x2_f32(a[lane], b[lane], c[lane][16]) for one wave -- pins the fragment
 * layout the kernels (and the CPU stand-in of the test-suite) assume */
int nf_debug_mfma32(const float* a, const float* b, const float* c, float* d, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a6  raw2outputs                          ref: ibrnet/render_ray.py:123-170
 * raw [R,S,4], z_vals [R,S], pixel_mask [R,S] bytes (0/1) -> rgb [R,3], depth [R], weights [R,S], alpha [R,S],
 * ray_mask [R] bytes (sum_s pixel_mask > 8).  Exactly one of pixel_mask and view_mask is given: view_mask [R,S,n_views] are the
 * projector's per-view validity flags, from which the kernel forms pixel_mask = (sum_v view_mask > 1) itself (:210).
 * ---------------------------------------------------------------------------------------------------------------- */
int nf_composite_fwd(const float* raw, const float* z_vals, const uint8_t* pixel_mask, const float* view_mask, int n_views,
                     int64_t n_rays, int n_samples, int white_bkgd, float* rgb, float* depth, float* weights, float* alpha,
                     uint8_t* ray_mask, nf_stream_t stream);
/* upstream gradients (each nullable): d_rgb [R,3], d_depth [R], d_weights [R,S], d_alpha [R,S] -> d_raw [R,S,4] */
int nf_composite_bwd(const float* raw, const float* z_vals, int64_t n_rays, int n_samples, int white_bkgd,
                     const float* d_rgb, const float* d_depth, const float* d_weights, const float* d_alpha,
                     float* d_raw, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a7  sample_pdf + fine-sample assembly    ref: ibrnet/render_ray.py:24-70, :216-237
 * z_vals [R,S] (ascending), weights [R,S] (detached), u_rand nullable [R,N_imp] (det=False)
 * -> z_out [R,S+N_imp] ascending (the sorted union).
 * ---------------------------------------------------------------------------------------------------------------- */
int nf_sample_fine(const float* z_vals, const float* weights, int64_t n_rays, int n_samples, int n_importance,
                   int inv_uniform, const float* u_rand, float* z_out, nf_stream_t stream);
/* sample_pdf alone (ibrnet/render_ray.py:24-70): bins [R,M+1], weights [R,M] (1e-5 is added inside, as the reference
 * does), u_rand nullable [R,N] -> samples [R,N] */
int nf_sample_pdf(const float* bins, const float* weights, int64_t n_rays, int n_bins, int n_samples,
                  const float* u_rand, float* samples, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a8  Criterion / img2mse (masked)         ref: ibrnet/criterion.py:23-33, utils.py:48-58
 * out [3] = (loss, sum_r mask_r * |rgb_r - gt_r|^2, sum_r mask_r).  mask nullable (plain mean).
 * cnt_override (nullable, device [1]): global mask count of a ray-sharded batch; then
 * loss = local_sum / (cnt_override*3 + 1e-6).
 * ---------------------------------------------------------------------------------------------------------------- */
int nf_masked_mse_fwd(const float* rgb, const float* gt, const uint8_t* mask, int64_t n_rays,
                      const float* cnt_override, float* out, nf_stream_t stream);
/* d_rgb[r,c] = d_loss * 2 (rgb-gt) mask_r / (cnt*3 + 1e-6); cnt = device [1] (out+2 of the forward or the override) */
int nf_masked_mse_bwd(const float* rgb, const float* gt, const uint8_t* mask, int64_t n_rays, const float* cnt,
                      const float* d_loss, float* d_rgb, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a14 (glue only)  decoder `upconv`: F.interpolate(scale 2, bilinear, align_corners=True) fused with the reflect padding of
 * the convolution behind it.   ref: ibrnet/feature_network.py:143-151.
 *   x: `planes` planes of h x w floats addressed by (xs_plane, xs_row) element strides (unit column stride; may be the
 *   interior of a padded tensor); y_padded [planes, 2h+2p, 2w+2p].  The backward is nf_in_act_pad_bwd (fold, act 0)
 *   followed by the transposed interpolation.
 * ---------------------------------------------------------------------------------------------------------------- */
int nf_upsample2x_pad_fwd(const float* x, int64_t planes, int64_t xs_plane, int64_t xs_row, int h, int w, int pad,
                          float* y_padded, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a14  ResUNet 1x1 convolutions (out_conv with bias, the stride-2 downsample branches) as fp32 MFMA GEMMs over the pixels.
 *      ref: ibrnet/feature_network.py:196-203, :62-70.
 *   y[n, co, r, c] = bias[co] + sum_ci M[co][ci] x[n, ci, r, c] on an H x W pixel grid; x and y are addressed with element
 *   strides (image, channel, row, column): NCHW, NHWC and stride-2 subsampled views are the same call.  `records` =
 *   nf_conv1x1_pack(weight [c_out][c_in]) (HOST pointers); with transposed != 0 the records hold weight^T and the call
 *   nf_conv1x1(..., c_in := c_out, c_out := c_in) is the backward-data pass.  x2 (nullable): input channels
 *   c_split .. c_in - 1 come from x2[n, ci - c_split, r, c] (same strides as x; c_split a multiple of 32) -- out_conv's backward
 *   reads the gradients of the coarse and the fine feature maps (feature_network.py:265-268) where autograd delivers them.
 * ---------------------------------------------------------------------------------------------------------------- */
int64_t nf_conv1x1_pack_floats(int c_out, int c_in);
int nf_conv1x1_pack(const float* weight_host, int c_out, int c_in, int transposed, float* records_host);
int nf_conv1x1(const float* records, const float* bias, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h, int64_t xs_w,
               float* y, int64_t ys_n, int64_t ys_c, int64_t ys_h, int64_t ys_w, int n_img, int H, int W, int c_in, int c_out,
               const float* x2, int c_split, nf_stream_t stream);

/* Layout / padding glue of the ResUNet executor (csrc/nf_pad.hip; ibrnet/feature_network.py:188 reflect padding of the input,
 * :231-243 skipconnect zero padding, :143-151 upsampling in front of a reflect-padded convolution -- its backward):
 *   nf_pad_gather_fwd: out[n][c] (contiguous (H+2p) x (W+2p) planes at os_n / os_c) = reflect-pad_p( Z ), Z = the H x W plane holding
 *       the eh x ew source window at (top, left), zeros elsewhere; source addressed by element strides (n, c, row, column)
 *   nf_pad_gather_bwd: its adjoint, din written through element strides
 *   nf_upsample2x_pad_bwd: adjoint of nf_upsample2x_pad_fwd (folds the reflect padding, transposed bilinear interpolation) */
int nf_pad_gather_fwd(const float* src, int64_t ss_n, int64_t ss_c, int64_t ss_h, int64_t ss_w, int n_img, int C, int eh, int ew, int top,
                      int left, int H, int W, int pad, float* out, int64_t os_n, int64_t os_c, nf_stream_t stream);
int nf_pad_gather_bwd(const float* dout, int64_t os_n, int64_t os_c, int n_img, int C, int H, int W, int pad, int eh, int ew, int top,
                      int left, float* din, int64_t ds_n, int64_t ds_c, int64_t ds_h, int64_t ds_w, nf_stream_t stream);
int nf_upsample2x_pad_bwd(const float* d_y_padded, int64_t planes, int h, int w, int pad, float* dx, int64_t xs_plane, int64_t xs_row,
                          nf_stream_t stream);

/* The network's stride-2 convolutions -- the 7x7 stem and the first 3x3 convolution of layer1 / layer2 / layer3
 * (ibrnet/feature_network.py:188, :192-195 via :51) -- on pre-padded activations (padding 0), forward and backward-data, as
 * direct implicit-GEMM convolutions on the fp32 matrix cores (csrc/nf_conv_s2.hip; replaces MIOpen / rocBLAS + col2im).
 *   records = nf_conv_s2_pack(weight [c_out][c_in][ks][ks], ks in {3, 7}, backward) (HOST pointers, nf_conv_s2_pack_floats floats)
 *   forward:  y [N, c_out, Ho, Wo], Ho = (Hi - ks) / 2 + 1;  the 7x7 form takes c_in <= 3
 *   backward: dx [N, c_in, Hi, Wi] from dy [N, c_out, Ho, Wo]; every element of dx is written (zeros where the convolution
 *             never read the input)
 * x / y / dx / dy: element strides (image, channel, row), unit column stride. */
int64_t nf_conv_s2_pack_floats(int c_out, int c_in, int ks, int backward);
int nf_conv_s2_pack(const float* weight_host, int c_out, int c_in, int ks, int backward, float* records_host);
int nf_conv_s2_fwd(const float* records, int ks, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h, int Hi, int Wi, float* y,
                   int64_t ys_n, int64_t ys_c, int64_t ys_h, int Ho, int Wo, int n_img, int c_in, int c_out, nf_stream_t stream);
int nf_conv_s2_bwd(const float* records, int ks, const float* dy, int64_t ds_n, int64_t ds_c, int64_t ds_h, int Ho, int Wo, float* dx,
                   int64_t xs_n, int64_t xs_c, int64_t xs_h, int Hi, int Wi, int n_img, int c_in, int c_out, nf_stream_t stream);
/* nf_conv_s2_fwd / nf_conv_s2_bwd for the 3x3 convolutions on the BF16 matrix cores, every fp32 operand as three bf16 parts (six cross
 * terms, error at fp32 rounding level; ref ibrnet/feature_network.py:192-195 via :51).  records = nf_conv_s2_x3_pack(weight
 * [c_out][c_in][3][3], backward) (HOST pointers, nf_conv_s2_x3_pack_floats floats; backward = 0: forward records). */
int64_t nf_conv_s2_x3_pack_floats(int c_out, int c_in, int backward);
int nf_conv_s2_x3_pack(const float* weight_host, int c_out, int c_in, int backward, float* records_host);
int nf_conv_s2_fwd_x3(const float* records, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h, int Hi, int Wi, float* y, int64_t ys_n,
                      int64_t ys_c, int64_t ys_h, int Ho, int Wo, int n_img, int c_in, int c_out, nf_stream_t stream);
/* the 7x7 stem (c_in <= 3) forward on the same operand split: records = nf_conv_s2_stem_x3_pack(weight [c_out][c_in][7][7]) (HOST pointers,
 * nf_conv_s2_stem_x3_pack_floats floats); ref ibrnet/feature_network.py:188 */
int64_t nf_conv_s2_stem_x3_pack_floats(int c_out);
int nf_conv_s2_stem_x3_pack(const float* weight_host, int c_out, int c_in, float* records_host);
int nf_conv_s2_stem_fwd_x3(const float* records, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h, int Hi, int Wi, float* y, int64_t ys_n,
                           int64_t ys_c, int64_t ys_h, int Ho, int Wo, int n_img, int c_in, int c_out, nf_stream_t stream);
int nf_conv_s2_bwd_x3(const float* records, const float* dy, int64_t ds_n, int64_t ds_c, int64_t ds_h, int Ho, int Wo, float* dx, int64_t xs_n,
                      int64_t xs_c, int64_t xs_h, int Hi, int Wi, int n_img, int c_in, int c_out, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a14  ResUNet 3x3 stride-1 convolutions as Winograd F(2x2, 3x3) on the fp32 matrix cores.
 *      ref: ibrnet/feature_network.py:28-36, 38-78, 127-151 (reflect padding is already in the input, see nf_in_act_pad_fwd).
 *   y[n, k, oy, ox] = sum_c sum_{a,b<3} x[n, c, oy + a - pad, ox + b - pad] W[k][c][a][b], zeros outside x, output Ho x Wo.
 *   forward: pad = 0, Ho = Hi - 2; backward-data: x = d y, pad = 2, Ho = Hi + 2, records packed with backward != 0.
 *   `records` = nf_wino_pack(weight [c_out][c_in][3][3], ..., k_per_group in {32, 64}) (HOST pointers).
 *   tile_blocks: 1 or 2 blocks of 32 Winograd tiles per wave (8 x 16 or 16 x 16 output pixels per workgroup), 0 = choose.
 *   x, y: element strides (image, channel, row), unit column stride.
 * ---------------------------------------------------------------------------------------------------------------- */
int64_t nf_wino_pack_floats(int c_out, int c_in, int k_per_group);
int nf_wino_pack(const float* weight_host, int c_out, int c_in, int backward, int k_per_group, float* records_host);
int nf_conv3x3_wino(const float* records, int k_per_group, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h, int Hi, int Wi,
                    int pad, float* y, int64_t ys_n, int64_t ys_c, int64_t ys_h, int Ho, int Wo, int n_img, int c_in, int c_out,
                    int tile_blocks, nf_stream_t stream);
/* The same convolutions with the Winograd-domain products on the BF16 matrix cores (csrc/nf_wino_bf.hip; ref: the same lines,
 * ibrnet/feature_network.py:28-36, 38-78, 127-151).  n_split = 3: every fp32 operand is the sum of three bf16 parts (8 significant
 * bits each) and the product is the six cross terms of order <= 2^-16 -- dropped terms <= 2^-24 of the product, i.e. fp32 rounding
 * level: a drop-in for nf_conv3x3_wino at 6 / 16 of its matrix-pipe time.  n_split = 2 (round 5; the executor's choice for the
 * BACKWARD-DATA passes): hi + mid, three cross terms, operands of 16 significant bits (dropped terms <= 3 x 2^-16 of a product).
 * n_split = 1: plain bf16 operands (diagnostic: 8 significant bits per operand, measured and rejected for the CNN).  records = nf_wino_bf_pack(weight, ...,
 * k_per_group in {32, 64}, n_split) (HOST pointers); tensors, geometry and strides as nf_conv3x3_wino. */
int64_t nf_wino_bf_pack_floats(int c_out, int c_in, int k_per_group, int n_split);
int nf_wino_bf_pack(const float* weight_host, int c_out, int c_in, int backward, int k_per_group, int n_split, float* records_host);
int nf_conv3x3_wino_bf(const float* records, int k_per_group, int n_split, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h, int Hi,
                       int Wi, int pad, float* y, int64_t ys_n, int64_t ys_c, int64_t ys_h, int Ho, int Wo, int n_img, int c_in, int c_out,
                       nf_stream_t stream);
/* The one-pixel border ring of a backward-data pass, g = d(padded input) [N, c_dx, H + 2, W + 2] from dy [N, c_dy, H, W]: the ring
 * sees one row / column of dy, i.e. four 1-D convolutions -- computed apart so that the Winograd kernel covers an H x W region
 * with the forward pass's block count (48 x 63: 24 blocks instead of 35) at pad 1 on the output shifted by (1, 1).
 *   kinds: bit 0 top row, 1 bottom row, 2 left column, 3 right column (columns: rows 1 .. H, and H + 1 too when the bottom row is not
 *   asked for -- the caller's Winograd region then ends at row H + 1 and the corners are the columns');  records = nf_wino_ring_pack(weight
 *   [c_dy = c_out][c_dx = c_in][3][3]) (HOST pointers).  ref: the same lines as nf_conv3x3_wino (backward of :28-36, 38-78). */
int64_t nf_wino_ring_pack_floats(int c_out, int c_in);
int nf_wino_ring_pack(const float* weight_host, int c_out, int c_in, float* records_host);
int nf_conv3x3_bwd_ring(const float* ring_records, const float* dy, int64_t ds_n, int64_t ds_c, int64_t ds_h, int H, int W, float* g,
                        int64_t gs_n, int64_t gs_c, int64_t gs_h, int n_img, int c_dy, int c_dx, int kinds, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a11/a13  perturbation update             ref: eval/ibrnet/eval_adv.py:28-29, :248-254, :805-839
 * All tensors flat [n] (delta, grad, exp_avg, exp_avg_sq, src all shaped [1,V,H,W,3]).
 *   nf_project_perturb : delta = max(min(delta, eps), -eps) (skipped if eps < 0); delta = max(min(delta, hi-src), lo-src)
 *   nf_pgd_adam_step   : g = -grad; torch.optim.Adam single-tensor update with neg_step_size = -(lr/bias_corr1),
 *                        bc2_sqrt = sqrt(bias_corr2), one_minus_beta{1,2} = 1 - beta (all computed in HOST double and
 *                        rounded to float once, exactly as torch does); then both clamps
 *   nf_pgd_adam_step_dev: the same update with {neg_step_size, bc2_sqrt} read from DEVICE memory (hyper_dev[2]; the host writes
 *                        them per iteration with a stream-ordered copy): the one launch of a PGD step whose scalar arguments change
 *                        with the iteration count, so that a whole step captured into a hipGraph (PGDAttack.step, INTEGRATION.md
 *                        section 4) can be replayed unchanged
 *   nf_pgd_sign_step   : delta += alpha * sign(grad); then both clamps
 * ---------------------------------------------------------------------------------------------------------------- */
int nf_project_perturb(float* delta, const float* src, int64_t n, float epsilon, float lower, float upper,
                       nf_stream_t stream);
int nf_pgd_adam_step(float* delta, const float* grad, float* exp_avg, float* exp_avg_sq, const float* src, int64_t n,
                     float neg_step_size, float one_minus_beta1, float beta2, float one_minus_beta2, float bc2_sqrt,
                     float adam_eps, float epsilon, float lower, float upper, nf_stream_t stream);
int nf_pgd_adam_step_dev(float* delta, const float* grad, float* exp_avg, float* exp_avg_sq, const float* src, int64_t n,
                         const float* hyper_dev, float one_minus_beta1, float beta2, float one_minus_beta2, float adam_eps,
                         float epsilon, float lower, float upper, nf_stream_t stream);
int nf_pgd_sign_step(float* delta, const float* grad, const float* src, int64_t n, float alpha, float epsilon,
                     float lower, float upper, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a15  GNT.forward (ret_alpha = False, eval mode)      ref: gnt/transformer_network.py:270-309 (+ :55-89, :93-113,
 * :121-171, :175-202).  Parameters as one blob whose layout nf_gnt_blob_entry enumerates (HOST; an empty name marks a
 * slot without parameters, e.g. q_fcs on odd layers).  rgb_feat [R,S,V,35], ray_diff [R,S,V,4], mask [R,S,V],
 * pts [R,S,3], ray_d [R,3] -> rgb [R,3]; alpha (nullable, [R,S]): the `ret_alpha` output, the attention row of the first
 * sample in the last ray transformer averaged over the heads (:196-200, :303-309).  save != 0 keeps all activations in `workspace`
 * (nf_gnt_workspace_floats(R,S,V,depth,1) floats) for nf_gnt_bwd: d_rgb [R,3] -> d_rgb_feat [R,S,V,35].
 * ---------------------------------------------------------------------------------------------------------------- */
int64_t nf_gnt_blob_floats(int depth);
int nf_gnt_blob_entry(int depth, int idx, char* name, int name_cap, int64_t* offset, int* rows, int* cols, int* transposed);
int64_t nf_gnt_workspace_floats(int64_t n_rays, int n_samples, int n_views, int depth, int save);
int nf_gnt_fwd(const float* blob, const float* rgb_feat, const float* ray_diff, const float* mask, const float* pts,
               const float* ray_d, int64_t n_rays, int n_samples, int n_views, int depth, int save, float* rgb,
               float* alpha, float* workspace, nf_stream_t stream);
int nf_gnt_bwd(const float* blob, const float* ray_diff, const float* mask, const float* d_rgb, int64_t n_rays, int n_samples,
               int n_views, int depth, float* d_rgb_feat, float* workspace, nf_stream_t stream);
/* The same pair in TRAINING mode (round 5): the reference's universal GNT loop runs before model.switch_to_eval()
 * (eval/gnt/eval_adv.py:739-878 vs :959), i.e. with the eight nn.Dropout(p = 0.1) sites of every layer live
 * (gnt/transformer_network.py:45-48 feed-forward hidden / output, :85-88 view-attention probabilities / output, :162-166 ray-attention
 * probabilities / output; with ret_alpha the returned attention is the dropped one).  The masks come from a counter-based generator --
 * keep(seed, site = 8 layer + {0..7}, flat index of the element in the tensor the reference hands to nn.Dropout), a 32-bit integer hash
 * (csrc/nf_gnt.h: gnt_keep) -- so nf_gnt_bwd_train regenerates the masks of its forward from the same (seed, p), and the CPU oracle
 * and the reference itself (masks injected into its modules, tests/golden/make_golden_gnt_train.py) evaluate identical ones.  A new
 * seed per call is the host's job (nerfool_amd/gnt/transformer_network.py). */
int nf_gnt_fwd_train(const float* blob, const float* rgb_feat, const float* ray_diff, const float* mask, const float* pts,
                     const float* ray_d, int64_t n_rays, int n_samples, int n_views, int depth, int save, float* rgb,
                     float* alpha, float* workspace, uint32_t seed, double p, nf_stream_t stream);
int nf_gnt_bwd_train(const float* blob, const float* ray_diff, const float* mask, const float* d_rgb, int64_t n_rays, int n_samples,
                     int n_views, int depth, float* d_rgb_feat, float* workspace, uint32_t seed, double p, nf_stream_t stream);

/* GNT forward on the matrix cores (S in {32, 64, 96, 128}): same arguments and workspace size as nf_gnt_fwd, with the weights
 * re-ordered by nf_gnt_pack_mfma (HOST pointers: natural blob -> MFMA-order blob of nf_gnt_mfma_blob_floats(depth) floats: the fp32
 * records, and behind them the bf16x3 image of the streamed records -- every weight as three bf16 parts -- that the streamed GEMMs
 * multiply on the bf16 matrix cores with fp32-grade results).
 * What it saves is for nf_gnt_bwd_mfma ONLY: the view softmax as masked logits + per-channel maximum / reciprocal sum, the ReLU
 * layers as sign words, the view-attention output (nf_gnt.h); nf_gnt_bwd reads the workspace of nf_gnt_fwd.
 *                                                                             ref: gnt/transformer_network.py:270-309 */
int64_t nf_gnt_mfma_blob_floats(int depth);
int nf_gnt_pack_mfma(int depth, const float* natural_blob_host, float* mfma_blob_host);
int nf_gnt_mfma_supported(int n_samples, int n_views);
int nf_gnt_fwd_mfma(const float* mfma_blob, const float* rgb_feat, const float* ray_diff, const float* mask, const float* pts,
                    const float* ray_d, int64_t n_rays, int n_samples, int n_views, int depth, int save, float* rgb,
                    float* alpha, float* workspace, nf_stream_t stream);
int nf_gnt_bwd_mfma(const float* mfma_blob, const float* mask, const float* d_rgb, int64_t n_rays, int n_samples, int n_views,
                    int depth, float* d_rgb_feat, float* workspace, nf_stream_t stream);

/* The matrix-core pair in TRAINING mode (round 6): the Dropout-active network of the reference's universal GNT loop
 * (eval/gnt/eval_adv.py:739-878, before switch_to_eval at :959; sites gnt/transformer_network.py:45-48, :85-88, :162-166) with the
 * masks of nf_gnt_fwd_train (same generator, same indices: the two pairs evaluate identical masks for one seed).
 * seed_dev: optional DEVICE pointer to the seed word -- when non-null the kernels read the seed from it instead of `seed`, so a
 * hipGraph-captured PGD step takes a fresh seed per replay (the host refreshes the word in stream order before each replay).  The
 * backward must be given the seed / word of its forward.  Workspace and blob as nf_gnt_fwd_mfma / nf_gnt_bwd_mfma. */
int nf_gnt_fwd_train_mfma(const float* mfma_blob, const float* rgb_feat, const float* ray_diff, const float* mask, const float* pts,
                          const float* ray_d, int64_t n_rays, int n_samples, int n_views, int depth, int save, float* rgb,
                          float* alpha, float* workspace, uint32_t seed, double p, const uint32_t* seed_dev, nf_stream_t stream);
int nf_gnt_bwd_train_mfma(const float* mfma_blob, const float* mask, const float* d_rgb, int64_t n_rays, int n_samples, int n_views,
                          int depth, float* d_rgb_feat, float* workspace, uint32_t seed, double p, const uint32_t* seed_dev,
                          nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a14 (glue only)  ResUNet: InstanceNorm + affine + residual + ReLU/ELU + reflect padding fused into one pass over a
 * convolution output, producing the pre-padded input of the next convolution.   ref: ibrnet/feature_network.py:38-78,
 * :127-140 (padding_mode='reflect', InstanceNorm2d(affine, no running stats)).
 *   x [N,C,H,W] contiguous; gamma/beta [C] (both NULL: no normalisation); res (nullable) addressed by element strides;
 *   act 0 none / 1 ReLU / 2 ELU; y_padded [N,C,H+2p,W+2p]; mean, rstd [N*C] out (saved for the backward).
 * Backward: dy_padded (nullable) = gradient w.r.t. y_padded, d_extra (nullable, [N,C,H,W]) = additional gradient w.r.t.
 * the unpadded activated output (residual / skip consumers); d_res (nullable out) = gradient of the residual input;
 * y_n_stride / dy_n_stride: elements between consecutive images of y_padded / dy_padded (0 = packed); lets the C planes
 * be a channel slice of a wider tensor (the decoder's skip concatenation is written / folded slice by slice, no torch.cat).
 * d_extra_sub (nullable, [N,C,ceil(H/2),ceil(W/2)]): gradient of a stride-2 consumer of the unpadded output (the 1x1
 * downsample convolution), added at the even rows / columns.
 * beta (nullable): when given and no residual went into the activation, its derivative is recomputed from x and
 * y_padded is not read (may be NULL);
 * dx [N,C,H,W] = gradient of x.   scratch: 512 bytes per (image, channel) plane (fp64 partial sums of up to 32 workgroups; no atomics).
 * ---------------------------------------------------------------------------------------------------------------- */
int nf_in_act_pad_fwd(const float* x, int n_img, int C, int H, int W, const float* gamma, const float* beta, float eps,
                      const float* res, int64_t rs_n, int64_t rs_c, int64_t rs_h, int64_t rs_w, int act, int pad,
                      float* y_padded, int64_t y_n_stride, float* mean, float* rstd, void* scratch, nf_stream_t stream);
int nf_in_act_pad_bwd(const float* dy_padded, const float* d_extra, const float* y_padded, const float* x, int n_img, int C,
                      int H, int W, const float* gamma, const float* beta, const float* mean, const float* rstd, int act, int pad,
                      float* d_res, float* dx, void* scratch, int64_t dy_n_stride, const float* d_extra_sub, nf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFOOL_HIP_H */
