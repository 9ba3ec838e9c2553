"""Ray renderer: depth sampling, projection, IBRNet, compositing, hierarchical re-sampling -- the reference's
ibrnet/render_ray.py call surface (`sample_along_camera_ray`, `raw2outputs`, `render_rays`, same argument names and
return dicts) with every stage executed by a HIP kernel."""
from collections import OrderedDict

import torch

from .. import ops
from .projection import Projector


def sample_along_camera_ray(ray_o, ray_d, depth_range, N_samples, inv_uniform=False, det=False):
    """ref: ibrnet/render_ray.py:73-116.  -> pts [N_rays, N_samples, 3], z_vals [N_rays, N_samples]"""
    t_rand = None
    if not det:
        t_rand = torch.rand(ray_d.shape[0], N_samples, dtype=torch.float32, device=ray_d.device)
    return ops.sample_along_ray(ray_o, ray_d, depth_range, N_samples, inv_uniform, t_rand)


def sample_pdf(bins, weights, N_samples, det=False):
    """ref: ibrnet/render_ray.py:24-70.  bins [N_rays, M+1], weights [N_rays, M] -> [N_rays, N_samples].  Unlike the
    reference the caller's `weights` tensor is not modified in place (the 1e-5 is added inside the kernel)."""
    u = None if det else torch.rand(bins.shape[0], N_samples, dtype=torch.float32, device=bins.device)
    return ops.sample_pdf(bins.detach(), weights.detach(), N_samples, u)


class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z_vals, pixel_mask, white_bkgd):
        rgb, depth, weights, alpha, ray_mask = ops.composite_fwd(raw, z_vals, pixel_mask, white_bkgd)
        ctx.save_for_backward(raw, z_vals)
        ctx.white_bkgd = white_bkgd
        ctx.mark_non_differentiable(ray_mask)
        ctx.set_materialize_grads(False)          # unused outputs arrive as None instead of freshly zero-filled tensors
        return rgb, depth, weights, alpha, ray_mask

    @staticmethod
    def backward(ctx, d_rgb, d_depth, d_weights, d_alpha, _d_mask):
        raw, z_vals = ctx.saved_tensors
        if d_rgb is None:
            d_rgb = torch.zeros(raw.shape[0], 3, dtype=raw.dtype, device=raw.device)
        d_raw = ops.composite_bwd(raw, z_vals, ctx.white_bkgd, d_rgb, d_depth, d_weights, d_alpha)
        return d_raw, None, None, None


def raw2outputs(raw, z_vals, mask, white_bkgd=False, geo_noise=None):
    """ref: ibrnet/render_ray.py:123-170.  raw [N_rays, N_samples, 4], z_vals [N_rays, N_samples], mask (bool per sample:
    at least two valid observations -- or the per-view validity flags [N_rays, N_samples, V], from which the kernel forms it).
    geo_noise is a training-only option of the reference and is rejected here."""
    if geo_noise is not None and geo_noise > 0:
        raise NotImplementedError('geo_noise is a training-time option outside the attack path')
    rgb, depth, weights, alpha, ray_mask = _Composite.apply(raw, z_vals, mask, bool(white_bkgd))
    return OrderedDict([('rgb', rgb), ('depth', depth), ('weights', weights), ('mask', ray_mask), ('alpha', alpha),
                        ('z_vals', z_vals)])


def sample_fine_depths(z_vals, weights, N_importance, inv_uniform=False, det=False):
    """ref: ibrnet/render_ray.py:216-237 (sample_pdf on the detached inner weights, union with the coarse depths,
    sort) as one kernel.  -> [N_rays, N_samples + N_importance] ascending."""
    u = None
    if not det:
        u = torch.rand(z_vals.shape[0], N_importance, dtype=torch.float32, device=z_vals.device)
    return ops.sample_fine(z_vals.detach(), weights.detach(), N_importance, inv_uniform, u)


def camera_workspace(ray_batch, src_ray_batch=None):
    """ops.camera_setup for the cameras of a batch: a caller that renders many chunks of ONE view (render_single_image, a chunk loop)
    builds it once and hands it to every render_rays call as `cam_ws=`; render_rays builds it itself otherwise (one 5 us launch).
    The workspace is owned by the caller's loop -- nothing is remembered between calls, so an in-place edit of a camera tensor can
    never meet a stale projection."""
    src = ray_batch if src_ray_batch is None else src_ray_batch
    return ops.camera_setup(ray_batch['camera'].detach(), src['src_cameras'].detach())


def _level(pts, z_vals, ray_batch, src, net, featmap, projector, white_bkgd, geo_noise, cams):
    """cams: one-element list caching the camera workspace of this render_rays call (both levels see the same cameras)"""
    can = getattr(net, 'can_gather', None)
    ours = isinstance(projector, Projector)
    if ours and cams[0] is None:
        cams[0] = camera_workspace(ray_batch, src)
    if can is not None and ours and can(featmap, pts.shape[1], src['src_cameras'].shape[1]):
        # projection + bilinear gather run inside the network's row kernel (and their adjoint inside its backward)
        raw, mask = net.forward_gathered(pts, cams[0], src['src_rgbs'][0], featmap)
        views = mask                                    # [n_rays, n_samples, n_views]
    else:
        rgb_feat, ray_diff, mask = projector.compute(pts, ray_batch['camera'], src['src_rgbs'], src['src_cameras'],
                                                     featmaps=featmap, **({'cam_ws': cams[0]} if ours else {}))
        views = mask[..., 0]
        raw = net(rgb_feat, ray_diff, mask)
    # "at least 2 observations" (:210) is counted by the compositing kernel from the per-view flags
    return raw2outputs(raw, z_vals, views, white_bkgd=white_bkgd, geo_noise=geo_noise)


def render_rays(ray_batch, model, featmaps, projector, N_samples, inv_uniform=False, N_importance=0, det=False,
                white_bkgd=False, args=None, src_ray_batch=None, geo_noise=None, cam_ws=None):
    """
    :param ray_batch: {'ray_o': [N_rays, 3], 'ray_d': [N_rays, 3], 'camera', 'depth_range', 'src_rgbs', 'src_cameras'}
    :param cam_ws: optional `camera_workspace(ray_batch, src_ray_batch)` of a caller that loops over chunks of one view
    :param model: object with .net_coarse / .net_fine
    :param featmaps: (coarse [V,32,Hf,Wf], fine [V,32,Hf,Wf])
    :return: {'outputs_coarse': OrderedDict, 'outputs_fine': OrderedDict or None}   (ibrnet/render_ray.py:173-256)
    """
    src = ray_batch if src_ray_batch is None else src_ray_batch
    ret = {'outputs_coarse': None, 'outputs_fine': None}
    pts, z_vals = sample_along_camera_ray(ray_batch['ray_o'], ray_batch['ray_d'], ray_batch['depth_range'], N_samples,
                                          inv_uniform=inv_uniform, det=det)
    cams = [cam_ws]
    ret['outputs_coarse'] = _level(pts, z_vals, ray_batch, src, model.net_coarse, featmaps[0], projector, white_bkgd,
                                   geo_noise, cams)
    if N_importance > 0:
        assert model.net_fine is not None
        z_vals = sample_fine_depths(z_vals, ret['outputs_coarse']['weights'], N_importance, inv_uniform, det)
        pts = ops.points_from_depths(ray_batch['ray_o'], ray_batch['ray_d'], z_vals)
        ret['outputs_fine'] = _level(pts, z_vals, ray_batch, src, model.net_fine, featmaps[1], projector, white_bkgd,
                                     geo_noise, cams)
    return ret


def render_rays_hybrid(ray_batch, model, featmaps, projector, N_samples, inv_uniform=False, N_importance=0, det=False,
                       white_bkgd=False, args=None, src_ray_batch=None, featmaps_clean=None):
    """Clean-colour / clean-density ablation (ibrnet/render_ray.py:261-389): every level is evaluated twice, on the
    attacked and on the clean feature maps, and the composited `raw` takes colour and density from the source selected
    by args.use_clean_color / args.use_clean_density.  The sample mask comes from the attacked pass (identical geometry)."""
    src = ray_batch if src_ray_batch is None else src_ray_batch
    assert featmaps_clean is not None

    def level(pts, z_vals, net, fm_adv, fm_clean):
        rgb_feat, ray_diff, mask = projector.compute(pts, ray_batch['camera'], src['src_rgbs'], src['src_cameras'],
                                                     featmaps=fm_adv)
        pixel_mask = ops.pixel_mask(mask[..., 0])
        raw_adv = net(rgb_feat, ray_diff, mask)
        rgb_feat_c, ray_diff_c, mask_c = projector.compute(pts, ray_batch['camera'], src['src_rgbs'], src['src_cameras'],
                                                           featmaps=fm_clean)
        raw_clean = net(rgb_feat_c, ray_diff_c, mask_c)
        colour = raw_clean if args.use_clean_color else raw_adv
        density = raw_clean if args.use_clean_density else raw_adv
        raw = torch.cat([colour[:, :, :3], density[:, :, 3:4]], dim=2)       # channel select, no arithmetic
        return raw2outputs(raw, z_vals, pixel_mask, white_bkgd=white_bkgd)

    ret = {'outputs_coarse': None, 'outputs_fine': None}
    pts, z_vals = sample_along_camera_ray(ray_batch['ray_o'], ray_batch['ray_d'], ray_batch['depth_range'], N_samples,
                                          inv_uniform=inv_uniform, det=det)
    ret['outputs_coarse'] = level(pts, z_vals, model.net_coarse, featmaps[0], featmaps_clean[0])
    if N_importance > 0:
        assert model.net_fine is not None
        z_vals = sample_fine_depths(z_vals, ret['outputs_coarse']['weights'], N_importance, inv_uniform, det)
        pts = ops.points_from_depths(ray_batch['ray_o'], ray_batch['ray_d'], z_vals)
        ret['outputs_fine'] = level(pts, z_vals, model.net_fine, featmaps[1], featmaps_clean[1])
    return ret
