"""Full-image rendering for the GNT flavour (gnt/render_image.py:6-130 call surface): chunk loop over the GNT render_rays,
outputs that the renderer leaves at None (weights / depth without ret_alpha) stay None, chunk results leave for page-locked host
tensors on a second stream while the next chunk renders, or -- with `shard` -- are rendered by all ranks and assembled by one
collective (ibrnet/render_image.run_chunks)."""
from ..ibrnet.render_image import run_chunks
from .render_ray import render_rays, render_rays_hybrid


def render_single_image(ray_sampler, ray_batch, model, projector, chunk_size, N_samples, inv_uniform=False, N_importance=0,
                        det=False, white_bkgd=False, render_stride=1, featmaps=None, ret_alpha=False, single_net=False,
                        args=None, src_ray_batch=None, featmaps_clean=None, shard=None):
    hybrid = args is not None and (getattr(args, 'use_clean_color', False) or getattr(args, 'use_clean_density', False))
    if hybrid:
        assert featmaps_clean is not None
    kw = dict(projector=projector, N_samples=N_samples, inv_uniform=inv_uniform, N_importance=N_importance, det=det,
              white_bkgd=white_bkgd, ret_alpha=ret_alpha, single_net=single_net, args=args, src_ray_batch=src_ray_batch)
    if hybrid:      # gnt/render_image.py:51-70
        render_chunk = lambda chunk: render_rays_hybrid(chunk, model, featmaps, featmaps_clean=featmaps_clean, **kw)
    else:
        render_chunk = lambda chunk: render_rays(chunk, model, featmaps, **kw)
    return run_chunks(ray_batch, chunk_size, render_chunk, len(range(0, ray_sampler.H, render_stride)),
                      len(range(0, ray_sampler.W, render_stride)), shard)
