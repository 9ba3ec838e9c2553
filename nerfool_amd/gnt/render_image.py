"""Full-image rendering for the GNT flavour (gnt/render_image.py:6-130 call surface): chunk loop over the GNT render_rays,
outputs that the renderer leaves at None (weights / depth without ret_alpha) stay None, chunk results stay in HBM until the
single device-to-host copy at the end."""
from collections import OrderedDict

import torch

from .render_ray import render_rays, render_rays_hybrid

_WHOLE = ('camera', 'depth_range', 'src_rgbs', 'src_cameras')


def render_single_image(ray_sampler, ray_batch, model, projector, chunk_size, N_samples, inv_uniform=False, N_importance=0,
                        det=False, white_bkgd=False, render_stride=1, featmaps=None, ret_alpha=False, single_net=False,
                        args=None, src_ray_batch=None, featmaps_clean=None):
    hybrid = args is not None and (getattr(args, 'use_clean_color', False) or getattr(args, 'use_clean_density', False))
    if hybrid:
        assert featmaps_clean is not None
    parts = {'outputs_coarse': OrderedDict(), 'outputs_fine': OrderedDict()}
    n_rays = ray_batch['ray_o'].shape[0]
    with torch.no_grad():
        for i in range(0, n_rays, chunk_size):
            chunk = OrderedDict((k, v if (k in _WHOLE or v is None) else v[i:i + chunk_size]) for k, v in ray_batch.items())
            kw = dict(projector=projector, N_samples=N_samples, inv_uniform=inv_uniform, N_importance=N_importance, det=det,
                      white_bkgd=white_bkgd, ret_alpha=ret_alpha, single_net=single_net, args=args, src_ray_batch=src_ray_batch)
            if hybrid:      # gnt/render_image.py:51-70
                ret = render_rays_hybrid(chunk, model, featmaps, featmaps_clean=featmaps_clean, **kw)
            else:
                ret = render_rays(chunk, model, featmaps, **kw)
            for level in ('outputs_coarse', 'outputs_fine'):
                if ret[level] is None:
                    parts[level] = None
                    continue
                for k, v in ret[level].items():
                    parts[level].setdefault(k, [])
                    if v is not None:
                        parts[level][k].append(v)
    hs = len(range(0, ray_sampler.H, render_stride))
    ws = len(range(0, ray_sampler.W, render_stride))
    all_ret = OrderedDict([('outputs_coarse', OrderedDict()), ('outputs_fine', OrderedDict())])
    for level in ('outputs_coarse', 'outputs_fine'):
        if parts[level] is None:
            all_ret[level] = None
            continue
        for k, lst in parts[level].items():
            all_ret[level][k] = torch.cat(lst, dim=0).reshape(hs, ws, -1).squeeze().cpu() if lst else None
    return all_ret
