"""Full-image rendering for the GNT flavour (gnt/render_image.py:6-130 call surface): chunk loop over the GNT render_rays,
outputs that the renderer leaves at None (weights / depth without ret_alpha) stay None, chunk results leave for page-locked host
tensors on a second stream while the next chunk renders (ibrnet/render_image.HostCollector)."""
from collections import OrderedDict

import torch

from ..ibrnet.render_image import HostCollector
from .render_ray import render_rays, render_rays_hybrid

_WHOLE = ('camera', 'depth_range', 'src_rgbs', 'src_cameras')


def render_single_image(ray_sampler, ray_batch, model, projector, chunk_size, N_samples, inv_uniform=False, N_importance=0,
                        det=False, white_bkgd=False, render_stride=1, featmaps=None, ret_alpha=False, single_net=False,
                        args=None, src_ray_batch=None, featmaps_clean=None):
    hybrid = args is not None and (getattr(args, 'use_clean_color', False) or getattr(args, 'use_clean_density', False))
    if hybrid:
        assert featmaps_clean is not None
    n_rays = ray_batch['ray_o'].shape[0]
    out = HostCollector(n_rays, ray_batch['ray_o'].device)
    with torch.no_grad():
        for i in range(0, n_rays, chunk_size):
            chunk = OrderedDict((k, v if (k in _WHOLE or v is None) else v[i:i + chunk_size]) for k, v in ray_batch.items())
            kw = dict(projector=projector, N_samples=N_samples, inv_uniform=inv_uniform, N_importance=N_importance, det=det,
                      white_bkgd=white_bkgd, ret_alpha=ret_alpha, single_net=single_net, args=args, src_ray_batch=src_ray_batch)
            if hybrid:      # gnt/render_image.py:51-70
                ret = render_rays_hybrid(chunk, model, featmaps, featmaps_clean=featmaps_clean, **kw)
            else:
                ret = render_rays(chunk, model, featmaps, **kw)
            out.add(i, ret)
    all_ret = out.finish(len(range(0, ray_sampler.H, render_stride)), len(range(0, ray_sampler.W, render_stride)))
    return all_ret
