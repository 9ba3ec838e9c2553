"""GNT renderer (gnt/render_ray.py:196-279 call surface): sampling and projection are the kernels shared with the IBRNet
flavour, the per-ray network is GNT; there is no compositing stage -- the network outputs the pixel colour.
Built for the released configurations: N_importance = 0, single_net = True, ret_alpha = False."""
from ..ibrnet.render_ray import sample_along_camera_ray


def render_rays(ray_batch, model, featmaps, projector, N_samples, inv_uniform=False, N_importance=0, det=False,
                white_bkgd=False, ret_alpha=False, single_net=True, args=None, src_ray_batch=None, geo_noise=None):
    if N_importance > 0 or ret_alpha:
        raise NotImplementedError('GNT hierarchical sampling / ret_alpha are not built (configs/gnt/*.txt use N_importance = 0)')
    src = ray_batch if src_ray_batch is None else src_ray_batch
    ray_o, ray_d = ray_batch['ray_o'], ray_batch['ray_d']
    pts, z_vals = sample_along_camera_ray(ray_o, ray_d, ray_batch['depth_range'], N_samples, inv_uniform=inv_uniform, det=det)
    rgb_feat, ray_diff, mask = projector.compute(pts, ray_batch['camera'], src['src_rgbs'], src['src_cameras'],
                                                 featmaps=featmaps[0])
    rgb = model.net_coarse(rgb_feat, ray_diff, mask, pts, ray_d)
    return {'outputs_coarse': {'rgb': rgb, 'weights': None, 'depth': None}, 'outputs_fine': None}
