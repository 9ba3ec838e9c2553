"""Full-image rendering: chunk loop over render_rays (ibrnet/render_image.py:21-123 call surface and return schema).

Unlike the reference (a synchronising `.cpu()` per key per chunk, render_image.py:98-102), a chunk's outputs leave for the
host on a second HIP stream while the next chunk renders: 2.3 KB per ray at 64 + 64 samples, 1.8 GB for a 756 x 1008 image --
a third of the image time when moved at the end through pageable memory.  The host tensors are page-locked (torch's caching
host allocator: pinned once, reused by later calls); the returned values are identical."""
from collections import OrderedDict

import torch

from .render_ray import render_rays, render_rays_hybrid

_WHOLE = ('camera', 'depth_range', 'src_rgbs', 'src_cameras')


class HostCollector:
    """chunk outputs -> page-locked host tensors [n_rays, ...] per level and key, copied on a second stream as they appear"""

    def __init__(self, n_rays, device):
        self.n_rays, self.device = n_rays, device
        self.on_gpu = device.type == 'cuda'
        self.host = {'outputs_coarse': OrderedDict(), 'outputs_fine': OrderedDict()}
        self.copy_stream = torch.cuda.Stream(device) if self.on_gpu else None
        self.in_flight = []          # chunk outputs stay referenced until their copies are done

    def add(self, i, ret):
        """outputs of the chunk that starts at ray i (a level may be None, and so may a key of the GNT flavour)"""
        if self.on_gpu:
            self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        for level in ('outputs_coarse', 'outputs_fine'):
            if ret[level] is None:
                self.host[level] = None
                continue
            for k, v in ret[level].items():
                if v is None:
                    self.host[level].setdefault(k, None)
                    continue
                if self.host[level].get(k) is None:
                    self.host[level][k] = torch.empty((self.n_rays,) + tuple(v.shape[1:]), dtype=v.dtype, pin_memory=self.on_gpu)
                dst = self.host[level][k][i:i + v.shape[0]]
                if self.on_gpu:
                    with torch.cuda.stream(self.copy_stream):
                        dst.copy_(v, non_blocking=True)
                    self.in_flight.append(v)
                else:
                    dst.copy_(v)

    def finish(self, hs, ws):
        """[hs, ws, ...] views of the host tensors, the reference's return schema"""
        if self.on_gpu:
            self.copy_stream.synchronize()
        self.in_flight = []
        all_ret = OrderedDict([('outputs_coarse', OrderedDict()), ('outputs_fine', OrderedDict())])
        for level in ('outputs_coarse', 'outputs_fine'):
            if self.host[level] is None:
                all_ret[level] = None
                continue
            for k, t in self.host[level].items():
                all_ret[level][k] = None if t is None else t.reshape(hs, ws, -1).squeeze()
        return all_ret


def render_single_image(ray_sampler, ray_batch, model, projector, chunk_size, N_samples, inv_uniform=False,
                        N_importance=0, det=False, white_bkgd=False, render_stride=1, featmaps=None, args=None,
                        featmaps_clean=None, src_ray_batch=None):
    hybrid = args is not None and (getattr(args, 'use_clean_color', False) or getattr(args, 'use_clean_density', False))
    if hybrid:
        assert featmaps_clean is not None
    n_rays = ray_batch['ray_o'].shape[0]
    out = HostCollector(n_rays, ray_batch['ray_o'].device)
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
            out.add(i, ret)
    all_ret = out.finish(len(range(0, ray_sampler.H, render_stride)), len(range(0, ray_sampler.W, render_stride)))
    coarse = all_ret['outputs_coarse']
    coarse['rgb'][coarse['mask'] == 0] = 1.       # coarse level only (render_image.py:113)
    return all_ret
