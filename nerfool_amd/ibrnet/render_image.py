"""Full-image rendering: chunk loop over render_rays (ibrnet/render_image.py:21-123 call surface and return schema).

Unlike the reference, chunk outputs stay in HBM and are moved to the host once per key at the end (the reference
synchronises with a `.cpu()` per key per chunk, render_image.py:98-102); the returned tensors are identical."""
from collections import OrderedDict

import torch

from .render_ray import render_rays, render_rays_hybrid

_WHOLE = ('camera', 'depth_range', 'src_rgbs', 'src_cameras')


def render_single_image(ray_sampler, ray_batch, model, projector, chunk_size, N_samples, inv_uniform=False,
                        N_importance=0, det=False, white_bkgd=False, render_stride=1, featmaps=None, args=None,
                        featmaps_clean=None, src_ray_batch=None):
    hybrid = args is not None and (getattr(args, 'use_clean_color', False) or getattr(args, 'use_clean_density', False))
    if hybrid:
        assert featmaps_clean is not None
    parts = {'outputs_coarse': OrderedDict(), 'outputs_fine': OrderedDict()}
    n_rays = ray_batch['ray_o'].shape[0]
    with torch.no_grad():
        for i in range(0, n_rays, chunk_size):
            chunk = OrderedDict()
            for k, v in ray_batch.items():
                chunk[k] = v if (k in _WHOLE or v is None) else v[i:i + chunk_size]
            if hybrid:
                ret = render_rays_hybrid(chunk, model, featmaps, projector=projector, N_samples=N_samples,
                                         inv_uniform=inv_uniform, N_importance=N_importance, det=det, white_bkgd=white_bkgd,
                                         args=args, src_ray_batch=src_ray_batch, featmaps_clean=featmaps_clean)
            else:
                ret = render_rays(chunk, model, featmaps, projector=projector, N_samples=N_samples,
                                  inv_uniform=inv_uniform, N_importance=N_importance, det=det, white_bkgd=white_bkgd,
                                  args=args, src_ray_batch=src_ray_batch)
            for level in ('outputs_coarse', 'outputs_fine'):
                if ret[level] is None:
                    parts[level] = None
                    continue
                for k, v in ret[level].items():
                    parts[level].setdefault(k, []).append(v)
    hs = len(range(0, ray_sampler.H, render_stride))
    ws = len(range(0, ray_sampler.W, render_stride))
    all_ret = OrderedDict([('outputs_coarse', OrderedDict()), ('outputs_fine', OrderedDict())])
    for level in ('outputs_coarse', 'outputs_fine'):
        if parts[level] is None:
            all_ret[level] = None
            continue
        for k, lst in parts[level].items():
            all_ret[level][k] = torch.cat(lst, dim=0).reshape(hs, ws, -1).squeeze().cpu()
    coarse = all_ret['outputs_coarse']
    coarse['rgb'][coarse['mask'] == 0] = 1.       # coarse level only (render_image.py:113)
    return all_ret
