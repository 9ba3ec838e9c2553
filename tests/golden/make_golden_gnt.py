#!/usr/bin/env python
"""Golden vectors of the GNT flavour (reference gnt/ package imported from /root/reference, eval mode).  Build container only.
    python tests/golden/make_golden_gnt.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _refimport  # noqa: E402

_refimport.install('gnt')

from gnt.projection import Projector  # noqa: E402
from gnt.render_ray import render_rays, render_rays_hybrid  # noqa: E402
from gnt.transformer_network import GNT  # noqa: E402
import gnt.sample_ray as ref_sample_ray  # noqa: E402

from nerfool_amd.synthetic import make_scene, smooth_featmaps  # noqa: E402
from oracle.gnt_ref import random_gnt_params  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy().copy() if isinstance(t, torch.Tensor) else np.array(t)


def params_checksum(params):
    """order-independent fingerprint of a state dict: per tensor the float64 sum and sum of squares, summed over the sorted keys"""
    s1 = sum(float(v.double().sum()) for _, v in sorted(params.items()))
    s2 = sum(float((v.double() ** 2).sum()) for _, v in sorted(params.items()))
    return np.array([s1, s2], dtype=np.float64)


def case(name, H, W, V, R, S, depth, seed, tilt=0.0, store_weights=True):
    """store_weights=False: the network parameters are regenerated from their seed by the tests (oracle.gnt_ref.random_gnt_params,
    the call below) -- the fixture keeps the seed and a fingerprint instead of megabytes of weights (depth 8: 3.5 MB)."""
    torch.manual_seed(seed)
    data = make_scene(H, W, V, seed=seed, tilt=tilt)
    Hf, Wf = max(6, H // 4), max(8, W // 4)
    fm = smooth_featmaps(V, 32, Hf, Wf, seed=seed).requires_grad_(True)
    params = random_gnt_params(depth, seed=60 + seed)
    net = GNT(SimpleNamespace(netwidth=64, trans_depth=depth), in_feat_ch=32, posenc_dim=63, viewenc_dim=63, ret_alpha=False)
    net.load_state_dict(params, strict=True)
    net.eval()
    model = SimpleNamespace(net_coarse=net, net_fine=None)
    ref_sample_ray.rng.seed(234)
    sampler = ref_sample_ray.RaySamplerSingleImage(data, 'cpu')
    batch = sampler.random_sample(R, sample_mode='uniform', center_ratio=0.8)
    ret = render_rays(batch, model, (fm, fm), Projector(device='cpu'), S, inv_uniform=True, N_importance=0, det=True,
                      ret_alpha=False, single_net=True)
    rgb = ret['outputs_coarse']['rgb']
    loss = torch.mean((rgb - batch['rgb']) ** 2)
    grad, = torch.autograd.grad(loss, fm)
    out = {'cfg': np.array([H, W, V, R, S, depth, Hf, Wf], dtype=np.int64)}
    for k in ('rgb', 'camera', 'src_rgbs', 'src_cameras', 'depth_range'):
        out['in/' + k] = npy(data[k])
    out['in/featmap'] = npy(fm)
    out['in/ray_o'] = npy(batch['ray_o'])
    out['in/ray_d'] = npy(batch['ray_d'])
    out['in/gt_rgb'] = npy(batch['rgb'])
    if store_weights:
        for k, v in params.items():
            out['net/' + k] = npy(v)
    else:
        out['net_seed'] = np.array([depth, 60 + seed], dtype=np.int64)
        out['net_checksum'] = params_checksum(params)
    out['rgb'] = npy(rgb)
    out['loss'] = npy(loss)
    out['grad/featmap'] = npy(grad)
    # network-level capture for the kernel tests
    from gnt.render_ray import sample_along_camera_ray
    pts, z = sample_along_camera_ray(batch['ray_o'], batch['ray_d'], batch['depth_range'], S, inv_uniform=True, det=True)
    rgb_feat, ray_diff, mask = Projector(device='cpu').compute(pts, batch['camera'], batch['src_rgbs'], batch['src_cameras'],
                                                              featmaps=fm)
    out['net_in/rgb_feat'] = npy(rgb_feat)
    out['net_in/ray_diff'] = npy(ray_diff)
    out['net_in/mask'] = npy(mask)
    out['net_in/pts'] = npy(pts)
    print('%-24s loss %.6f  rgb range [%.3f, %.3f]  valid %.3f' % (name, float(loss), float(rgb.min()), float(rgb.max()),
                                                                  float(mask.mean())))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('   %.1f KB' % (os.path.getsize(os.path.join(HERE, name + '.npz')) / 1024.))


def alpha_case(name, H, W, V, R, S, N_imp, depth, seed, tilt=0.0):
    """ret_alpha = True (attention-derived weights / depth) and hierarchical sampling with a single network
    (gnt/render_ray.py:249-277): colour, weights, depth of both passes and the gradient of the summed loss."""
    torch.manual_seed(seed)
    data = make_scene(H, W, V, seed=seed, tilt=tilt)
    Hf, Wf = max(6, H // 4), max(8, W // 4)
    fm = smooth_featmaps(V, 32, Hf, Wf, seed=seed).requires_grad_(True)
    params = random_gnt_params(depth, seed=60 + seed)
    net = GNT(SimpleNamespace(netwidth=64, trans_depth=depth), in_feat_ch=32, posenc_dim=63, viewenc_dim=63, ret_alpha=True)
    net.load_state_dict(params, strict=True)
    net.eval()
    model = SimpleNamespace(net_coarse=net, net_fine=None)
    ref_sample_ray.rng.seed(234)
    sampler = ref_sample_ray.RaySamplerSingleImage(data, 'cpu')
    batch = sampler.random_sample(R, sample_mode='uniform', center_ratio=0.8)
    ret = render_rays(batch, model, (fm, fm), Projector(device='cpu'), S, inv_uniform=True, N_importance=N_imp, det=True,
                      ret_alpha=True, single_net=True)
    loss = sum(torch.mean((ret[k]['rgb'] - batch['rgb']) ** 2) for k in ('outputs_coarse', 'outputs_fine'))
    grad, = torch.autograd.grad(loss, fm)
    out = {'cfg': np.array([H, W, V, R, S, depth, Hf, Wf, N_imp], dtype=np.int64)}
    for k in ('rgb', 'camera', 'src_rgbs', 'src_cameras', 'depth_range'):
        out['in/' + k] = npy(data[k])
    out['in/featmap'] = npy(fm)
    out['in/ray_o'] = npy(batch['ray_o'])
    out['in/ray_d'] = npy(batch['ray_d'])
    out['in/gt_rgb'] = npy(batch['rgb'])
    for k, v in params.items():
        out['net/' + k] = npy(v)
    for lvl in ('outputs_coarse', 'outputs_fine'):
        for k in ('rgb', 'weights', 'depth'):
            out['%s/%s' % (lvl, k)] = npy(ret[lvl][k])
    out['loss'] = npy(loss)
    out['grad/featmap'] = npy(grad)
    # clean-colour / clean-density ablation (gnt/render_ray.py:282-387): the same network on perturbed and on clean feature maps
    fm_clean = smooth_featmaps(V, 32, Hf, Wf, seed=seed + 50)
    out['in/featmap_clean'] = npy(fm_clean)
    with torch.no_grad():
        for tag in ('clean_color', 'clean_density'):
            a = SimpleNamespace(use_clean_color=tag == 'clean_color', use_clean_density=tag == 'clean_density')
            h = render_rays_hybrid(batch, model, (fm, fm), Projector(device='cpu'), S, inv_uniform=True, N_importance=N_imp, det=True,
                                   ret_alpha=True, single_net=True, args=a, featmaps_clean=(fm_clean, fm_clean))
            for lvl in ('outputs_coarse', 'outputs_fine'):
                for k in ('rgb', 'weights', 'depth'):
                    out['hybrid/%s/%s/%s' % (tag, lvl, k)] = npy(h[lvl][k])
    print('%-24s loss %.6f  weights sum %.4f' % (name, float(loss), float(ret['outputs_coarse']['weights'].sum(-1).mean())))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('   %.1f KB' % (os.path.getsize(os.path.join(HERE, name + '.npz')) / 1024.))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'c4':
        # BASELINE config 4's network shape: trans_depth 8 (configs/gnt/gnt_full.txt:26), 10 source views, 64 samples per ray -- the shape
        # the matrix-core GNT kernels run at in the benchmark (gnt/transformer_network.py:270-309)
        case('gnt_c4_d8_v10', 32, 48, 10, 8, 64, 8, seed=4, tilt=0.3, store_weights=False)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'alpha':
        alpha_case('gnt_alpha_d2_v3', 32, 48, 3, 10, 32, 32, 2, seed=3, tilt=0.3)
        sys.exit(0)
    case('gnt_tiny_d2_v4', 32, 48, 4, 12, 8, 2, seed=0, tilt=0.4)
    case('gnt_tiny_d3_v5', 32, 48, 5, 10, 12, 3, seed=1, tilt=0.3)
