"""Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): IBRNet ray renderer on PyTorch-CPU fp32.

Every function restates one reference function and cites it (paths relative to the NeRFool tree).
The code is deliberately functional (state-dict in, tensors out), differentiable through autograd, and
free of any import from nerfool_amd/.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------------
# a2  depth sampling                                   ref: ibrnet/render_ray.py:73-116
# --------------------------------------------------------------------------------------------------


def sample_along_camera_ray(ray_o, ray_d, depth_range, N_samples, inv_uniform=False, det=False):
    near, far = depth_range[0, 0], depth_range[0, 1]
    assert near > 0 and far > 0 and far > near
    n_rays = ray_d.shape[0]
    near_v = near * torch.ones(n_rays, dtype=ray_d.dtype, device=ray_d.device)
    far_v = far * torch.ones(n_rays, dtype=ray_d.dtype, device=ray_d.device)
    idx = torch.arange(N_samples, dtype=ray_d.dtype, device=ray_d.device)[None, :]
    if inv_uniform:
        first = 1.0 / near_v
        delta = (1.0 / far_v - first) / (N_samples - 1)
        z_vals = 1.0 / (first[:, None] + idx * delta[:, None])
    else:
        delta = (far_v - near_v) / (N_samples - 1)
        z_vals = near_v[:, None] + idx * delta[:, None]
    if not det:   # stratified jitter (training only; eval_adv.py:530 forces det=True)
        mids = 0.5 * (z_vals[:, 1:] + z_vals[:, :-1])
        hi = torch.cat([mids, z_vals[:, -1:]], dim=-1)
        lo = torch.cat([z_vals[:, :1], mids], dim=-1)
        z_vals = lo + (hi - lo) * torch.rand_like(z_vals)
    pts = z_vals[:, :, None] * ray_d[:, None, :] + ray_o[:, None, :]
    return pts, z_vals


# --------------------------------------------------------------------------------------------------
# a3  projection + bilinear gather + direction deltas   ref: ibrnet/projection.py:24-132
# --------------------------------------------------------------------------------------------------


def camera_matrices(cams):
    """cams [V,34] -> (hw [2], K [V,4,4], c2w [V,4,4]).  ref: ibrnet/sample_ray.py:27-32 layout."""
    return cams[0, :2], cams[:, 2:18].reshape(-1, 4, 4), cams[:, 18:34].reshape(-1, 4, 4)


def compute_projections(xyz, train_cameras):
    """ref: ibrnet/projection.py:42-62.  xyz [R,S,3] -> pixel_locations [V,R,S,2], in_front [V,R,S]."""
    lead = xyz.shape[:2]
    pts = xyz.reshape(-1, 3)
    _, K, c2w = camera_matrices(train_cameras)
    P = K.bmm(torch.inverse(c2w))                                    # [V,4,4]
    pts_h = torch.cat([pts, torch.ones_like(pts[:, :1])], dim=-1)    # [N,4]
    proj = P.bmm(pts_h.t()[None].expand(P.shape[0], -1, -1)).permute(0, 2, 1)   # [V,N,4]
    pix = proj[..., :2] / torch.clamp(proj[..., 2:3], min=1e-8)
    pix = torch.clamp(pix, min=-1e6, max=1e6)
    in_front = proj[..., 2] > 0
    V = P.shape[0]
    return pix.reshape((V,) + lead + (2,)), in_front.reshape((V,) + lead)


def compute_angle(xyz, query_camera, train_cameras):
    """ref: ibrnet/projection.py:64-87.  -> ray_diff [V,R,S,4]."""
    lead = xyz.shape[:2]
    pts = xyz.reshape(-1, 3)
    c_src = train_cameras[:, 18:34].reshape(-1, 4, 4)[:, :3, 3]      # [V,3]
    c_tar = query_camera[18:34].reshape(4, 4)[:3, 3]                 # [3]
    to_tar = c_tar[None, None, :] - pts[None]                        # [1,N,3]
    to_tar = to_tar / (torch.norm(to_tar, dim=-1, keepdim=True) + 1e-6)
    to_src = c_src[:, None, :] - pts[None]                           # [V,N,3]
    to_src = to_src / (torch.norm(to_src, dim=-1, keepdim=True) + 1e-6)
    diff = to_tar - to_src
    diff_len = torch.norm(diff, dim=-1, keepdim=True)
    dot = torch.sum(to_tar * to_src, dim=-1, keepdim=True)
    out = torch.cat([diff / torch.clamp(diff_len, min=1e-6), dot], dim=-1)
    return out.reshape((c_src.shape[0],) + lead + (4,))


def bilinear_zero_pad(img, pix_x, pix_y):
    """Hand-rolled `grid_sample(align_corners=True, zeros)` in pixel units, used to document/check the
    tap arithmetic the HIP kernel implements.  img [V,C,Hm,Wm]; pix_x/pix_y [V,N] -> [V,C,N]."""
    V, C, Hm, Wm = img.shape
    x0 = torch.floor(pix_x)
    y0 = torch.floor(pix_y)
    wx1 = pix_x - x0
    wy1 = pix_y - y0
    out = torch.zeros(V, C, pix_x.shape[1], dtype=img.dtype)
    flat = img.reshape(V, C, Hm * Wm)
    for dy, wy in ((0, 1 - wy1), (1, wy1)):
        for dx, wx in ((0, 1 - wx1), (1, wx1)):
            xi = (x0 + dx).long()
            yi = (y0 + dy).long()
            ok = (xi >= 0) & (xi <= Wm - 1) & (yi >= 0) & (yi <= Hm - 1)
            lin = (yi.clamp(0, Hm - 1) * Wm + xi.clamp(0, Wm - 1))[:, None, :].expand(-1, C, -1)
            out = out + torch.gather(flat, 2, lin) * (wx * wy * ok.to(img.dtype))[:, None, :]
    return out


def projector_compute(xyz, query_camera, train_imgs, train_cameras, featmaps, return_pixels=False):
    """ref: ibrnet/projection.py:89-132 (Projector.compute).

    xyz [R,S,3]; query_camera [1,34]; train_imgs [1,V,H,W,3]; train_cameras [1,V,34]; featmaps [V,C,Hf,Wf]
    -> rgb_feat [R,S,V,3+C], ray_diff [R,S,V,4], mask [R,S,V,1]
    """
    assert train_imgs.shape[0] == 1 and train_cameras.shape[0] == 1 and query_camera.shape[0] == 1
    cams = train_cameras.detach()[0]
    qcam = query_camera[0]
    imgs = train_imgs[0].permute(0, 3, 1, 2)                         # [V,3,H,W]
    h, w = cams[0][:2]
    pix, in_front = compute_projections(xyz, cams)
    scale = torch.stack([w - 1.0, h - 1.0]).to(pix)[None, None, None, :]
    grid = 2 * pix / scale - 1.0                                     # the SAME grid samples both maps
    rgb = F.grid_sample(imgs, grid, align_corners=True).permute(2, 3, 0, 1)
    feat = F.grid_sample(featmaps, grid, align_corners=True).permute(2, 3, 0, 1)
    rgb_feat = torch.cat([rgb, feat], dim=-1)
    inb = (pix[..., 0] <= w - 1.0) & (pix[..., 0] >= 0) & (pix[..., 1] <= h - 1.0) & (pix[..., 1] >= 0)
    ray_diff = compute_angle(xyz, qcam, cams).permute(1, 2, 0, 3)
    mask = (inb * in_front).float().permute(1, 2, 0)[..., None]
    if return_pixels:
        return rgb_feat, ray_diff, mask, pix
    return rgb_feat, ray_diff, mask


# --------------------------------------------------------------------------------------------------
# a4/a5  IBRNet per-sample network                      ref: ibrnet/mlp_network.py:152-274
# --------------------------------------------------------------------------------------------------


def posenc_table(n_samples, d_hid=16):
    """ref: ibrnet/mlp_network.py:210-220 (float64 table -> float32, leading batch dim)."""
    pos = np.arange(n_samples, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    ang = pos / np.power(10000.0, 2 * (j // 2) / d_hid)
    tab = np.where(j % 2 == 0, np.sin(ang), np.cos(ang))
    return torch.from_numpy(tab).float()[None]


IBRNET_LAYERS = [
    # (prefix, [(in, out), ...]) in module order -- used by random_ibrnet_params and the packers
    ('ray_dir_fc', [(0, 4, 16), (2, 16, 35)]),
    ('base_fc', [(0, 105, 64), (2, 64, 32)]),
    ('vis_fc', [(0, 32, 32), (2, 32, 33)]),
    ('vis_fc2', [(0, 32, 32), (2, 32, 1)]),
    ('geometry_fc', [(0, 65, 64), (2, 64, 16)]),
    ('out_geometry_fc', [(0, 16, 16), (2, 16, 1)]),
    ('rgb_fc', [(0, 37, 16), (2, 16, 8), (4, 8, 1)]),
]


def random_ibrnet_params(n_samples, seed, sigma_bias=1.0, gain=1.0):
    """Fixture weights: Kaiming-normal like the reference init (mlp_network.py:136-141) with small random
    biases so that every bias path is exercised, and out_geometry_fc.2.bias shifted by `sigma_bias` so that
    sigma/alpha/T are non-trivial (SURVEY 8c caveat)."""
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    p['s'] = torch.tensor(0.2)
    for prefix, layers in IBRNET_LAYERS:
        for idx, fan_in, fan_out in layers:
            p['%s.%d.weight' % (prefix, idx)] = torch.randn(fan_out, fan_in, generator=g) * gain * (2.0 / fan_in) ** 0.5
            p['%s.%d.bias' % (prefix, idx)] = torch.randn(fan_out, generator=g) * 0.05
    p['out_geometry_fc.2.bias'] = p['out_geometry_fc.2.bias'] + sigma_bias
    for name in ('w_qs', 'w_ks', 'w_vs', 'fc'):
        p['ray_attention.%s.weight' % name] = torch.randn(16, 16, generator=g) * 0.25
    p['ray_attention.layer_norm.weight'] = 1.0 + 0.1 * torch.randn(16, generator=g)
    p['ray_attention.layer_norm.bias'] = 0.1 * torch.randn(16, generator=g)
    p['pos_encoding'] = posenc_table(n_samples)
    return p


def _lin(p, name, x):
    return F.linear(x, p[name + '.weight'], p[name + '.bias'])


def _weighted_mean_var(x, w):
    """ref: ibrnet/mlp_network.py:144-149 (fused_mean_variance), reduction over the view axis (dim 2)."""
    mean = torch.sum(x * w, dim=2, keepdim=True)
    var = torch.sum(w * (x - mean) ** 2, dim=2, keepdim=True)
    return mean, var


def ray_attention(p, x, row_mask):
    """ref: ibrnet/mlp_network.py:69-119 + :23-43.  x [R,S,16]; row_mask [R,S,1] float (1 = keep).
    4 heads x d_k=d_v=4, temperature 2, masked QUERY rows -> -1e9 before softmax, post-LN eps 1e-6."""
    R, S, _ = x.shape
    q = F.linear(x, p['ray_attention.w_qs.weight']).view(R, S, 4, 4).transpose(1, 2)
    k = F.linear(x, p['ray_attention.w_ks.weight']).view(R, S, 4, 4).transpose(1, 2)
    v = F.linear(x, p['ray_attention.w_vs.weight']).view(R, S, 4, 4).transpose(1, 2)
    scores = torch.matmul(q / 2.0, k.transpose(2, 3))               # [R,4,S,S]
    scores = scores.masked_fill(row_mask[:, None] == 0, -1e9)       # mask [R,1,S,1] broadcasts over keys
    attn = F.softmax(scores, dim=-1)
    o = torch.matmul(attn, v).transpose(1, 2).reshape(R, S, 16)
    o = F.linear(o, p['ray_attention.fc.weight']) + x
    return F.layer_norm(o, (16,), p['ray_attention.layer_norm.weight'], p['ray_attention.layer_norm.bias'], eps=1e-6)


def ibrnet_forward(p, rgb_feat, ray_diff, mask, anti_alias_pooling=True, return_aux=False):
    """ref: ibrnet/mlp_network.py:222-274 (IBRNet.forward).
    rgb_feat [R,S,V,35], ray_diff [R,S,V,4], mask [R,S,V,1] -> [R,S,4] (rgb, sigma)."""
    V = rgb_feat.shape[2]
    dir_feat = F.elu(_lin(p, 'ray_dir_fc.2', F.elu(_lin(p, 'ray_dir_fc.0', ray_diff))))
    rgb_in = rgb_feat[..., :3]
    f = rgb_feat + dir_feat
    if anti_alias_pooling:
        dot = ray_diff[..., 3:4]
        e = torch.exp(torch.abs(p['s']) * (dot - 1))
        w = (e - torch.min(e, dim=2, keepdim=True)[0]) * mask
        w = w / (torch.sum(w, dim=2, keepdim=True) + 1e-8)
    else:
        w = mask / (torch.sum(mask, dim=2, keepdim=True) + 1e-8)

    mean, var = _weighted_mean_var(f, w)
    x = torch.cat([mean.expand(-1, -1, V, -1), var.expand(-1, -1, V, -1), f], dim=-1)
    x = F.elu(_lin(p, 'base_fc.2', F.elu(_lin(p, 'base_fc.0', x))))
    x_base = x

    xv = F.elu(_lin(p, 'vis_fc.2', F.elu(_lin(p, 'vis_fc.0', x * w))))
    x_res, vis = xv[..., :32], xv[..., 32:33]
    vis = torch.sigmoid(vis) * mask
    x = x + x_res
    vis = torch.sigmoid(_lin(p, 'vis_fc2.2', F.elu(_lin(p, 'vis_fc2.0', x * vis)))) * mask
    w2 = vis / (torch.sum(vis, dim=2, keepdim=True) + 1e-8)

    mean2, var2 = _weighted_mean_var(x, w2)
    g = torch.cat([mean2.squeeze(2), var2.squeeze(2), w2.mean(dim=2)], dim=-1)       # [R,S,65]
    g = F.elu(_lin(p, 'geometry_fc.2', F.elu(_lin(p, 'geometry_fc.0', g))))        # [R,S,16]
    n_valid = torch.sum(mask, dim=2)                                                # [R,S,1]
    g_pe = g + p['pos_encoding']
    g_att = ray_attention(p, g_pe, (n_valid > 1).float())
    sigma = F.relu(_lin(p, 'out_geometry_fc.2', F.elu(_lin(p, 'out_geometry_fc.0', g_att))))
    sigma = sigma.masked_fill(n_valid < 1, 0.0)

    y = torch.cat([x, vis, ray_diff], dim=-1)
    y = _lin(p, 'rgb_fc.4', F.elu(_lin(p, 'rgb_fc.2', F.elu(_lin(p, 'rgb_fc.0', y)))))
    y = y.masked_fill(mask == 0, -1e9)
    blend = F.softmax(y, dim=2)
    rgb = torch.sum(rgb_in * blend, dim=2)
    out = torch.cat([rgb, sigma], dim=-1)
    if return_aux:
        return out, {'weight': w, 'base': x_base, 'base_res': x, 'vis': vis, 'weight2': w2, 'globalfeat': g, 'attn_out': g_att,
                     'sigma': sigma, 'blend': blend}
    return out


# --------------------------------------------------------------------------------------------------
# a6  alpha compositing                                 ref: ibrnet/render_ray.py:123-170
# --------------------------------------------------------------------------------------------------


def raw2outputs(raw, z_vals, mask, white_bkgd=False):
    rgb = raw[:, :, :3]
    sigma = raw[:, :, 3]
    alpha = 1.0 - torch.exp(-sigma)                                  # intervals intentionally unused (:139)
    T = torch.cumprod(1.0 - alpha + 1e-10, dim=-1)[:, :-1]
    T = torch.cat([torch.ones_like(T[:, :1]), T], dim=-1)
    weights = alpha * T
    rgb_map = torch.sum(weights[:, :, None] * rgb, dim=1)
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - torch.sum(weights, dim=-1, keepdim=True))
    ray_mask = mask.float().sum(dim=1) > 8
    depth_map = torch.sum(weights * z_vals, dim=-1)
    return OrderedDict([('rgb', rgb_map), ('depth', depth_map), ('weights', weights),
                        ('mask', ray_mask), ('alpha', alpha), ('z_vals', z_vals)])


# --------------------------------------------------------------------------------------------------
# a7  hierarchical re-sampling                          ref: ibrnet/render_ray.py:24-70, 216-243
# --------------------------------------------------------------------------------------------------


def sample_pdf(bins, weights, N_samples, det=False):
    """bins [R,M+1], weights [R,M] -> [R,N_samples].  `above` is the count of cdf edges (first M of the M+1)
    that are <= u, exactly the reference's M-iteration compare loop (:48-50)."""
    M = weights.shape[1]
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], dim=-1)     # [R,M+1]
    if det:
        u = torch.linspace(0.0, 1.0, N_samples)[None].repeat(bins.shape[0], 1)
    else:
        u = torch.rand(bins.shape[0], N_samples)
    above = (u[:, :, None] >= cdf[:, None, :M]).long().sum(dim=-1)    # in [1, M]
    below = torch.clamp(above - 1, min=0)
    cdf_lo, cdf_hi = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_lo, bin_hi = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_lo) / denom
    return bin_lo + t * (bin_hi - bin_lo)


def fine_depths(z_vals, weights, N_importance, inv_uniform, det=True):
    """ref: ibrnet/render_ray.py:216-237.  Returns the sorted union [R, S+N_importance]."""
    w = weights.clone().detach()[:, 1:-1]
    if inv_uniform:
        inv_z = 1.0 / z_vals
        inv_mid = 0.5 * (inv_z[:, 1:] + inv_z[:, :-1])
        inv_new = sample_pdf(torch.flip(inv_mid, dims=[1]), torch.flip(w, dims=[1]), N_importance, det=det)
        z_new = 1.0 / inv_new
    else:
        mid = 0.5 * (z_vals[:, 1:] + z_vals[:, :-1])
        z_new = sample_pdf(mid, w, N_importance, det=det)
    z_all, _ = torch.sort(torch.cat([z_vals, z_new], dim=-1), dim=-1)
    return z_all


# --------------------------------------------------------------------------------------------------
# a9  render_rays                                       ref: ibrnet/render_ray.py:173-256
# --------------------------------------------------------------------------------------------------


def render_rays(ray_batch, params_coarse, params_fine, featmaps, N_samples, inv_uniform=False, N_importance=0,
                det=False, white_bkgd=False, src_ray_batch=None, anti_alias_pooling=True, z_fine=None):
    """ref: ibrnet/render_ray.py:173-256.  z_fine (checker hook): use these fine-level depths instead of re-sampling --
    the inverse-CDF re-sampling is a non-differentiable, discontinuous step (a u_k within rounding of a cdf edge lands in
    either bin), so a gradient check evaluates the float64 oracle at the depths the checked implementation drew."""
    src = ray_batch if src_ray_batch is None else src_ray_batch
    ret = {'outputs_coarse': None, 'outputs_fine': None}
    pts, z_vals = sample_along_camera_ray(ray_batch['ray_o'], ray_batch['ray_d'], ray_batch['depth_range'],
                                          N_samples, inv_uniform=inv_uniform, det=det)
    rgb_feat, ray_diff, mask = projector_compute(pts, ray_batch['camera'], src['src_rgbs'], src['src_cameras'],
                                                 featmaps[0])
    pixel_mask = mask[..., 0].sum(dim=2) > 1
    raw = ibrnet_forward(params_coarse, rgb_feat, ray_diff, mask, anti_alias_pooling)
    ret['outputs_coarse'] = raw2outputs(raw, z_vals, pixel_mask, white_bkgd)
    if N_importance > 0:
        assert params_fine is not None
        z_vals = fine_depths(z_vals, ret['outputs_coarse']['weights'], N_importance, inv_uniform, det) if z_fine is None else z_fine
        pts = z_vals[:, :, None] * ray_batch['ray_d'][:, None, :] + ray_batch['ray_o'][:, None, :]
        rgb_feat, ray_diff, mask = projector_compute(pts, ray_batch['camera'], src['src_rgbs'], src['src_cameras'],
                                                     featmaps[1])
        pixel_mask = mask[..., 0].sum(dim=2) > 1
        raw = ibrnet_forward(params_fine, rgb_feat, ray_diff, mask, anti_alias_pooling)
        ret['outputs_fine'] = raw2outputs(raw, z_vals, pixel_mask, white_bkgd)
    return ret


def render_rays_hybrid(ray_batch, params_coarse, params_fine, featmaps, featmaps_clean, N_samples, use_clean_color,
                       use_clean_density, inv_uniform=False, N_importance=0, det=False, white_bkgd=False,
                       src_ray_batch=None, anti_alias_pooling=True):
    """ref: ibrnet/render_ray.py:261-389 -- each level evaluated on attacked and on clean feature maps, colour and
    density picked per flag, sample mask of the attacked pass."""
    src = ray_batch if src_ray_batch is None else src_ray_batch

    def level(pts, z_vals, params, fm_adv, fm_clean):
        outs = []
        for fm in (fm_adv, fm_clean):
            rgb_feat, ray_diff, mask = projector_compute(pts, ray_batch['camera'], src['src_rgbs'], src['src_cameras'], fm)
            outs.append((ibrnet_forward(params, rgb_feat, ray_diff, mask, anti_alias_pooling), mask))
        (raw_adv, mask_adv), (raw_clean, _) = outs
        colour = (raw_clean if use_clean_color else raw_adv)[:, :, :3]
        sigma = (raw_clean if use_clean_density else raw_adv)[:, :, 3:4]
        return raw2outputs(torch.cat([colour, sigma], dim=2), z_vals, mask_adv[..., 0].sum(dim=2) > 1, white_bkgd)

    ret = {'outputs_coarse': None, 'outputs_fine': None}
    pts, z_vals = sample_along_camera_ray(ray_batch['ray_o'], ray_batch['ray_d'], ray_batch['depth_range'], N_samples,
                                          inv_uniform=inv_uniform, det=det)
    ret['outputs_coarse'] = level(pts, z_vals, params_coarse, featmaps[0], featmaps_clean[0])
    if N_importance > 0:
        z_vals = fine_depths(z_vals, ret['outputs_coarse']['weights'], N_importance, inv_uniform, det)
        pts = z_vals[:, :, None] * ray_batch['ray_d'][:, None, :] + ray_batch['ray_o'][:, None, :]
        ret['outputs_fine'] = level(pts, z_vals, params_fine, featmaps[1], featmaps_clean[1])
    return ret


# --------------------------------------------------------------------------------------------------
# a8  masked MSE                                        ref: utils.py:48-58, ibrnet/criterion.py:23-33
# --------------------------------------------------------------------------------------------------


def img2mse(x, y, mask=None):
    if mask is None:
        return torch.mean((x - y) * (x - y))
    return torch.sum((x - y) * (x - y) * mask.unsqueeze(-1)) / (torch.sum(mask) * x.shape[-1] + 1e-6)


def criterion(outputs, ray_batch):
    return img2mse(outputs['rgb'], ray_batch['rgb'], outputs['mask'].float())


def mse2psnr(x):
    """ref: utils.py:35."""
    return -10.0 * np.log(x + 1e-6) / np.log(10.0)


# --------------------------------------------------------------------------------------------------
# a1  ray generation                                    ref: ibrnet/sample_ray.py:98-116
# --------------------------------------------------------------------------------------------------


def rays_single_image(H, W, intrinsics, c2w, render_stride=1):
    """Pixel grid without half-pixel offset; rays_d = R * K^-1 * (u,v,1); rays_o = camera centre."""
    u, v = np.meshgrid(np.arange(W)[::render_stride], np.arange(H)[::render_stride])
    pix = np.stack([u.reshape(-1), v.reshape(-1), np.ones(u.size)], axis=0).astype(np.float32)
    pix = torch.from_numpy(pix)[None]
    rays_d = c2w[:, :3, :3].bmm(torch.inverse(intrinsics[:, :3, :3])).bmm(pix).transpose(1, 2).reshape(-1, 3)
    rays_o = c2w[:, :3, 3][:, None, :].expand(-1, rays_d.shape[0], -1).reshape(-1, 3)
    return rays_o, rays_d


# --------------------------------------------------------------------------------------------------
# a10 render_single_image                               ref: ibrnet/render_image.py:21-123
# --------------------------------------------------------------------------------------------------


def render_single_image(H, W, ray_batch, params_coarse, params_fine, featmaps, chunk_size, N_samples,
                        inv_uniform=False, N_importance=0, det=False, white_bkgd=False, render_stride=1,
                        src_ray_batch=None, anti_alias_pooling=True):
    whole = ('camera', 'depth_range', 'src_rgbs', 'src_cameras')
    acc = {'outputs_coarse': OrderedDict(), 'outputs_fine': OrderedDict()}
    n = ray_batch['ray_o'].shape[0]
    for i in range(0, n, chunk_size):
        chunk = {k: (v if k in whole or v is None else v[i:i + chunk_size]) for k, v in ray_batch.items()}
        ret = render_rays(chunk, params_coarse, params_fine, featmaps, N_samples, inv_uniform, N_importance, det,
                          white_bkgd, src_ray_batch, anti_alias_pooling)
        for level in acc:
            if ret[level] is None:
                acc[level] = None
                continue
            for k, v in ret[level].items():
                acc[level].setdefault(k, []).append(v.detach().cpu())
    hs = len(range(0, H, render_stride))
    ws = len(range(0, W, render_stride))
    for level in ('outputs_coarse', 'outputs_fine'):
        if acc[level] is None:
            continue
        for k in acc[level]:
            acc[level][k] = torch.cat(acc[level][k], dim=0).reshape(hs, ws, -1).squeeze()
        if level == 'outputs_coarse':   # coarse only (:113)
            acc[level]['rgb'][acc[level]['mask'] == 0] = 1.0
    return acc
