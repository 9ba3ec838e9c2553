"""GNT renderer (gnt/render_ray.py:196-279 call surface): sampling and projection are the kernels shared with the IBRNet
flavour, the per-ray network is GNT; there is no compositing stage -- the network outputs the pixel colour, and with
ret_alpha the attention-derived per-sample weights that give the depth map and drive the hierarchical resampling."""
import torch

from .. import ops
from ..ibrnet.render_ray import sample_along_camera_ray, sample_fine_depths


def sample_fine_pts(inv_uniform, N_importance, det, N_samples, ray_batch, weights, z_vals):
    """gnt/render_ray.py:164-193: inverse-CDF resampling on the (detached) inner weights, union with the coarse depths, sort --
    the same arithmetic as ibrnet/render_ray.py:216-243, one fused kernel here."""
    z_vals = sample_fine_depths(z_vals, weights, N_importance, inv_uniform, det)
    return ops.points_from_depths(ray_batch['ray_o'], ray_batch['ray_d'], z_vals), z_vals


def _split(out, z_vals, ret_alpha):
    if not ret_alpha:
        return {'rgb': out, 'weights': None, 'depth': None}
    rgb, weights = out[:, 0:3], out[:, 3:]
    return {'rgb': rgb, 'weights': weights, 'depth': torch.sum(weights * z_vals, dim=-1)}


def render_rays(ray_batch, model, featmaps, projector, N_samples, inv_uniform=False, N_importance=0, det=False,
                white_bkgd=False, ret_alpha=False, single_net=True, args=None, src_ray_batch=None, geo_noise=None):
    src = ray_batch if src_ray_batch is None else src_ray_batch
    ray_o, ray_d = ray_batch['ray_o'], ray_batch['ray_d']
    pts, z_vals = sample_along_camera_ray(ray_o, ray_d, ray_batch['depth_range'], N_samples, inv_uniform=inv_uniform, det=det)
    rgb_feat, ray_diff, mask = projector.compute(pts, ray_batch['camera'], src['src_rgbs'], src['src_cameras'],
                                                 featmaps=featmaps[0])
    ret = {'outputs_coarse': _split(model.net_coarse(rgb_feat, ray_diff, mask, pts, ray_d), z_vals, ret_alpha),
           'outputs_fine': None}
    if N_importance > 0:
        if ret['outputs_coarse']['weights'] is None:
            raise ValueError('N_importance > 0 needs the attention weights of the coarse pass: construct GNT with ret_alpha=True')
        pts, z_vals = sample_fine_pts(inv_uniform, N_importance, det, N_samples, ray_batch,
                                      ret['outputs_coarse']['weights'].clone().detach(), z_vals)
        rgb_feat, ray_diff, mask = projector.compute(pts, ray_batch['camera'], src['src_rgbs'], src['src_cameras'],
                                                     featmaps=featmaps[1])
        net = model.net_coarse if single_net else model.net_fine
        ret['outputs_fine'] = _split(net(rgb_feat, ray_diff, mask, pts, ray_d), z_vals, True)
    return ret


def render_rays_hybrid(ray_batch, model, featmaps, projector, N_samples, inv_uniform=False, N_importance=0, det=False,
                       white_bkgd=False, ret_alpha=False, single_net=True, args=None, src_ray_batch=None, geo_noise=None,
                       featmaps_clean=None):
    """gnt/render_ray.py:282-387 (clean-colour / clean-density ablation): the coarse network runs on the perturbed AND on the
    clean feature maps; colour comes from the clean pass with args.use_clean_color, the attention weights (which drive the
    fine resampling) from the clean pass with args.use_clean_density; depth always from the perturbed pass; the fine pass
    uses the perturbed maps."""
    src = ray_batch if src_ray_batch is None else src_ray_batch
    ray_o, ray_d = ray_batch['ray_o'], ray_batch['ray_d']
    pts, z_vals = sample_along_camera_ray(ray_o, ray_d, ray_batch['depth_range'], N_samples, inv_uniform=inv_uniform, det=det)

    def coarse(fm):
        rgb_feat, ray_diff, mask = projector.compute(pts, ray_batch['camera'], src['src_rgbs'], src['src_cameras'], featmaps=fm)
        return _split(model.net_coarse(rgb_feat, ray_diff, mask, pts, ray_d), z_vals, ret_alpha)

    adv, clean = coarse(featmaps[0]), coarse(featmaps_clean[0])
    ret = {'outputs_coarse': {'rgb': clean['rgb'] if args.use_clean_color else adv['rgb'],
                              'weights': clean['weights'] if args.use_clean_density else adv['weights'],
                              'depth': adv['depth']},
           'outputs_fine': None}
    if N_importance > 0:
        pts_f, z_f = sample_fine_pts(inv_uniform, N_importance, det, N_samples, ray_batch,
                                     ret['outputs_coarse']['weights'].clone().detach(), z_vals)
        rgb_feat, ray_diff, mask = projector.compute(pts_f, ray_batch['camera'], src['src_rgbs'], src['src_cameras'],
                                                     featmaps=featmaps[1])
        net = model.net_coarse if single_net else model.net_fine
        ret['outputs_fine'] = _split(net(rgb_feat, ray_diff, mask, pts_f, ray_d), z_f, True)
    return ret
