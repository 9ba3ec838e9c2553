#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REFERENCE (imported from /root/reference
with stubbed third-party modules, see _refimport.py) on seeded synthetic inputs.

Runs only in the build container (the GPU box has no /root/reference).  Usage:
    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

What is stored is data only: inputs, fixture weights and the reference's outputs per stage.
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

_refimport.install('ibrnet')

import eval_adv as EA  # noqa: E402  (reference eval/ibrnet/eval_adv.py)
from ibrnet.criterion import Criterion  # noqa: E402
from ibrnet.feature_network import ResUNet  # noqa: E402
from ibrnet.mlp_network import IBRNet  # noqa: E402
from ibrnet.projection import Projector  # noqa: E402
from ibrnet.render_image import render_single_image  # noqa: E402
from ibrnet.render_ray import render_rays, render_rays_hybrid, sample_along_camera_ray, sample_pdf  # noqa: E402
import ibrnet.sample_ray as ref_sample_ray  # noqa: E402

from nerfool_amd.synthetic import make_scene, smooth_featmaps  # noqa: E402
from oracle.feature_net_ref import random_resunet_state  # noqa: E402
from oracle.ibrnet_ref import random_ibrnet_params  # noqa: E402

EA.criterion = Criterion()
torch.set_num_threads(8)


def npy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy().copy()     # copy: optimizers / in-place ops mutate the storage later
    return np.array(t)


def ref_net(params, n_samples, aa=1):
    net = IBRNet(SimpleNamespace(anti_alias_pooling=aa), in_feat_ch=32, n_samples=n_samples)
    sd = {k: v.clone() for k, v in params.items()}
    if not aa:
        sd.pop('s')
    net.load_state_dict(sd, strict=True)
    net.eval()
    return net


def pack_params(prefix, params, out):
    for k, v in params.items():
        out['%s/%s' % (prefix, k)] = npy(v)


def reset_pixel_rng():
    ref_sample_ray.rng.seed(234)


def stage_case(name, H, W, V, R, S, N_imp, inv_uniform, white_bkgd, seed, aa=1, tilt=0.0, push_forward=0.0,
               Hf=None, Wf=None, store_stages=True, depth_range=None):
    torch.manual_seed(seed)
    data = make_scene(H, W, V, seed=seed, tilt=tilt, push_forward=push_forward,
                      **({} if depth_range is None else {'depth_range': depth_range}))
    Hf = Hf or max(6, H // 4)
    Wf = Wf or max(8, W // 4)
    fm_c = smooth_featmaps(V, 32, Hf, Wf, seed=seed).requires_grad_(True)
    fm_f = smooth_featmaps(V, 32, Hf, Wf, seed=seed + 1).requires_grad_(True)
    pc = random_ibrnet_params(S, seed=10 + seed)
    pf = random_ibrnet_params(S + N_imp, seed=20 + seed) if N_imp > 0 else None
    model = SimpleNamespace(net_coarse=ref_net(pc, S, aa), net_fine=ref_net(pf, S + N_imp, aa) if pf else None)
    projector = Projector(device='cpu')

    reset_pixel_rng()
    sampler = ref_sample_ray.RaySamplerSingleImage(data, 'cpu')
    batch = sampler.random_sample(R, sample_mode='uniform', center_ratio=0.8)

    out = {}
    for k in ('rgb', 'camera', 'src_rgbs', 'src_cameras', 'depth_range'):
        out['in/' + k] = npy(data[k])
    out['in/featmap_coarse'] = npy(fm_c)
    out['in/featmap_fine'] = npy(fm_f)
    out['in/selected_inds'] = npy(batch['selected_inds']).astype(np.int64)
    out['in/ray_o'] = npy(batch['ray_o'])
    out['in/ray_d'] = npy(batch['ray_d'])
    out['in/gt_rgb'] = npy(batch['rgb'])
    out['cfg'] = np.array([H, W, V, R, S, N_imp, int(inv_uniform), int(white_bkgd), aa, Hf, Wf], dtype=np.int64)
    pack_params('coarse', pc, out)
    if pf:
        pack_params('fine', pf, out)

    # ---- stage captures on the coarse level (reference functions called one by one)
    pts, z = sample_along_camera_ray(batch['ray_o'], batch['ray_d'], batch['depth_range'], S,
                                     inv_uniform=inv_uniform, det=True)
    cams = batch['src_cameras'].squeeze(0)
    pix, front = projector.compute_projections(pts, cams)
    rgb_feat, ray_diff, mask = projector.compute(pts, batch['camera'], batch['src_rgbs'], batch['src_cameras'],
                                                 featmaps=fm_c)
    hooks = {}
    net = model.net_coarse
    hs = [net.base_fc.register_forward_hook(lambda m, i, o: hooks.__setitem__('base', o.detach().clone())),
          net.geometry_fc.register_forward_hook(lambda m, i, o: hooks.__setitem__('globalfeat', o.detach().clone())),
          net.ray_attention.register_forward_hook(lambda m, i, o: hooks.__setitem__('attn_out', o[0].detach().clone())),
          net.vis_fc2.register_forward_hook(lambda m, i, o: hooks.__setitem__('vis2_raw', o.detach().clone()))]
    raw_c = net(rgb_feat, ray_diff, mask)
    for h in hs:
        h.remove()
    if store_stages:
        out['coarse/pts'] = npy(pts)
        out['coarse/z'] = npy(z)
        out['coarse/pix'] = npy(pix)
        out['coarse/front'] = npy(front)
        out['coarse/rgb_feat'] = npy(rgb_feat)
        out['coarse/ray_diff'] = npy(ray_diff)
        out['coarse/mask'] = npy(mask)
        out['coarse/raw'] = npy(raw_c)
        for k, v in hooks.items():
            out['coarse/aux_' + k] = npy(v)

    # ---- the reference's own end-to-end call
    ret = render_rays(batch, model, (fm_c, fm_f), projector, S, inv_uniform=inv_uniform, N_importance=N_imp,
                      det=True, white_bkgd=white_bkgd)
    loss, _ = EA.criterion(ret['outputs_coarse'], batch)
    if ret['outputs_fine'] is not None:
        lf, _ = EA.criterion(ret['outputs_fine'], batch)
        loss = loss + lf
    grads = torch.autograd.grad(loss, [fm_c, fm_f] if N_imp > 0 else [fm_c], allow_unused=True)
    for level in ('outputs_coarse', 'outputs_fine'):
        if ret[level] is None:
            continue
        for k, v in ret[level].items():
            out['%s/%s' % (level, k)] = npy(v)
    out['loss'] = npy(loss)
    out['grad/featmap_coarse'] = npy(grads[0])
    if N_imp > 0:
        out['grad/featmap_fine'] = npy(grads[1])

    m = npy(mask)[..., 0]
    print('%-28s loss %.6f  valid-view frac %.3f  samples with <2 views %.3f  ray-mask coarse %d/%d  behind-cam %.3f'
          % (name, float(loss), m.mean(), (m.sum(-1) < 2).mean(), int(npy(ret['outputs_coarse']['mask']).sum()), R,
             1.0 - npy(front).mean()))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


def attack_case(name, H, W, V, R, S, N_imp, seed, n_adam=10, n_sign=3):
    """Full delta-path: reference ResUNet + render_rays + EA.optimize_adv_perturb + torch.optim.Adam / sign-PGD."""
    torch.manual_seed(seed)
    data = make_scene(H, W, V, seed=seed, tilt=0.3)
    cnn_sd = random_resunet_state(seed + 100)
    feature_net = ResUNet(coarse_out_ch=32, fine_out_ch=32, coarse_only=False)
    feature_net.load_state_dict(cnn_sd, strict=True)
    feature_net.eval()
    pc = random_ibrnet_params(S, seed=30 + seed)
    pf = random_ibrnet_params(S + N_imp, seed=40 + seed)
    model = SimpleNamespace(net_coarse=ref_net(pc, S), net_fine=ref_net(pf, S + N_imp), feature_net=feature_net)
    projector = Projector(device='cpu')
    args = SimpleNamespace(gt_depth_path=None, use_patch_sampling=False, N_rand=R, sample_mode='uniform',
                           center_ratio=0.8, use_pseudo_gt=False, N_samples=S, inv_uniform=True, N_importance=N_imp,
                           det=True, white_bkgd=False, density_loss=0, depth_var_loss=0, depth_diff_loss=0,
                           depth_consistency_loss=0, depth_smooth_loss=0, camera_consistency_loss=0,
                           perturb_camera=False)
    sampler = ref_sample_ray.RaySamplerSingleImage(data, 'cpu')
    src_ray_batch = sampler.get_all()
    eps = torch.tensor(8 / 255.)
    torch.manual_seed(seed + 5)
    delta0 = EA.init_adv_perturb(args, src_ray_batch, eps, 1, 0).detach().clone()

    out = {}
    for k in ('rgb', 'camera', 'src_rgbs', 'src_cameras', 'depth_range'):
        out['in/' + k] = npy(data[k])
    out['in/delta0'] = npy(delta0)
    out['cfg'] = np.array([H, W, V, R, S, N_imp, seed + 100, n_adam, n_sign], dtype=np.int64)
    pack_params('coarse', pc, out)
    pack_params('fine', pf, out)

    with torch.no_grad():
        fc, ff = feature_net((src_ray_batch['src_rgbs'] + delta0).squeeze(0).permute(0, 3, 1, 2))
    out['cnn/coarse_sum'] = npy(fc.double().sum())
    out['cnn/fine_sum'] = npy(ff.double().sum())
    out['cnn/coarse_sample'] = npy(fc.reshape(-1)[::37][:1000])
    out['cnn/fine_sample'] = npy(ff.reshape(-1)[::41][:1000])
    out['cnn/shape'] = np.array(fc.shape, dtype=np.int64)

    # Adam-ascent, README settings (adam_lr 1e-3, gamma 1 -> use gamma .5/step 4 here to exercise the schedule)
    reset_pixel_rng()
    delta = delta0.clone().requires_grad_(True)
    opt = torch.optim.Adam([delta], lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=4, gamma=0.5)
    inds, losses = [], []
    for it in range(n_adam):
        # optimize_adv_perturb re-draws pixels from the module-global rng; record them through a shim
        loss, _ = EA.optimize_adv_perturb(args, delta, model, projector, src_ray_batch, data, return_loss=True)
        opt.zero_grad()
        loss.backward()
        if it < 3:
            out['adam/grad_iter%d' % it] = npy(delta.grad)
        delta.grad.data *= -1
        opt.step()
        sched.step()
        delta.data = EA.clamp(delta.data, -eps, eps)
        delta.data = EA.clamp(delta.data, 0 - src_ray_batch['src_rgbs'], 1 - src_ray_batch['src_rgbs'])
        losses.append(float(loss))
        if (it + 1) in (1, 2, 3, n_adam):
            out['adam/delta_%d' % (it + 1)] = npy(delta.data)
        if it + 1 == 3:
            st = opt.state[delta]
            out['adam/exp_avg_3'] = npy(st['exp_avg'])
            out['adam/exp_avg_sq_3'] = npy(st['exp_avg_sq'])
    out['adam/losses'] = np.array(losses, dtype=np.float64)
    # the pixel picks, reproduced from the same stream
    rs = np.random.RandomState(234)
    out['adam/selected_inds'] = np.stack([rs.choice(H * W, size=(R,), replace=False) for _ in range(n_adam)])

    # sign-PGD (adv_lr 2/255)
    reset_pixel_rng()
    alpha = torch.tensor(2 / 255.)
    delta = delta0.clone().requires_grad_(True)
    losses = []
    for it in range(n_sign):
        loss, _ = EA.optimize_adv_perturb(args, delta, model, projector, src_ray_batch, data, return_loss=True)
        loss.backward()
        grad = delta.grad.detach()
        if it == 0:
            out['sign/grad_iter0'] = npy(grad)
        delta.data = delta.data + alpha * torch.sign(grad)
        delta.grad.zero_()
        delta.data = EA.clamp(delta.data, -eps, eps)
        delta.data = EA.clamp(delta.data, 0 - src_ray_batch['src_rgbs'], 1 - src_ray_batch['src_rgbs'])
        losses.append(float(loss))
        if it == 0:
            out['sign/delta_1'] = npy(delta.data)
    out['sign/delta_%d' % n_sign] = npy(delta.data)
    out['sign/losses'] = np.array(losses, dtype=np.float64)
    print('%-28s adam losses %s  sign losses %s' % (name, np.round(out['adam/losses'], 5), np.round(losses, 5)))

    # render_single_image with the final Adam delta (coarse mask==0 -> 1 rule included)
    with torch.no_grad():
        d_fin = torch.from_numpy(out['adam/delta_%d' % n_adam])
        featmaps = feature_net((src_ray_batch['src_rgbs'] + d_fin).squeeze(0).permute(0, 3, 1, 2))
        ray_batch = sampler.get_all()
        rargs = SimpleNamespace(use_clean_color=False, use_clean_density=False)
        ret = render_single_image(ray_sampler=sampler, ray_batch=ray_batch, model=model, projector=projector,
                                  chunk_size=1000, det=True, N_samples=S, inv_uniform=True, N_importance=N_imp,
                                  white_bkgd=False, featmaps=featmaps, args=rargs, src_ray_batch=src_ray_batch)
    for level in ('outputs_coarse', 'outputs_fine'):
        for k in ('rgb', 'depth', 'mask'):
            out['image/%s/%s' % (level, k)] = npy(ret[level][k])
    gt = data['rgb'][0]
    mse = float(torch.mean((ret['outputs_fine']['rgb'] - gt) ** 2))
    out['image/psnr_fine'] = np.array(-10. * np.log(mse + 1e-6) / np.log(10.))
    out['image/ray_o'] = npy(ray_batch['ray_o'][::97])
    out['image/ray_d'] = npy(ray_batch['ray_d'][::97])
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


def hybrid_case(name, base):
    """render_rays_hybrid (clean colour / clean density) of the reference on the inputs of stage case `base`; the clean
    feature maps are smooth_featmaps(seed 50 / 51) and are regenerated, not stored.  Also a stand-alone sample_pdf."""
    z = np.load(os.path.join(HERE, base + '.npz'))
    H, W, V, R, S, N_imp, inv_u, white, aa, Hf, Wf = [int(x) for x in z['cfg']]
    t = lambda k: torch.from_numpy(z[k])
    pc = {k[7:]: t(k) for k in z.files if k.startswith('coarse/') and ('.' in k or k in ('coarse/s', 'coarse/pos_encoding'))}
    pf = {k[5:]: t(k) for k in z.files if k.startswith('fine/') and ('.' in k or k in ('fine/s', 'fine/pos_encoding'))}
    model = SimpleNamespace(net_coarse=ref_net(pc, S, aa), net_fine=ref_net(pf, S + N_imp, aa))
    batch = {'ray_o': t('in/ray_o'), 'ray_d': t('in/ray_d'), 'rgb': t('in/gt_rgb'), 'camera': t('in/camera'),
             'depth_range': t('in/depth_range'), 'src_rgbs': t('in/src_rgbs'), 'src_cameras': t('in/src_cameras')}
    fm = (t('in/featmap_coarse'), t('in/featmap_fine'))
    fm_clean = (smooth_featmaps(V, 32, Hf, Wf, seed=50), smooth_featmaps(V, 32, Hf, Wf, seed=51))
    out = {'cfg': z['cfg'], 'base': np.array(base)}
    for tag, cc, cd in (('clean_color', True, False), ('clean_density', False, True)):
        args = SimpleNamespace(use_clean_color=cc, use_clean_density=cd)
        with torch.no_grad():
            ret = render_rays_hybrid(batch, model, fm, Projector(device='cpu'), S, inv_uniform=bool(inv_u),
                                     N_importance=N_imp, det=True, white_bkgd=bool(white), args=args,
                                     featmaps_clean=fm_clean)
        for level in ('outputs_coarse', 'outputs_fine'):
            for k in ('rgb', 'depth', 'weights', 'z_vals'):
                out['%s/%s/%s' % (tag, level, k)] = npy(ret[level][k])
    gen = torch.Generator().manual_seed(9)
    bins = torch.sort(torch.rand(12, 21, generator=gen) * 4 + 2, dim=1)[0]
    w = torch.rand(12, 20, generator=gen) ** 3
    w[3] = 0                                     # an all-zero row: uniform pdf from the +1e-5
    out['pdf/bins'] = npy(bins)
    out['pdf/weights'] = npy(w)
    out['pdf/samples_17'] = npy(sample_pdf(bins.clone(), w.clone(), 17, det=True))
    out['pdf/samples_64'] = npy(sample_pdf(bins.clone(), w.clone(), 64, det=True))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('%-28s written' % name)


if __name__ == '__main__':
    stage_case('ibrnet_tiny_invu', 40, 56, 4, 24, 16, 16, True, False, seed=0, tilt=0.45)
    stage_case('ibrnet_tiny_lin_white', 40, 56, 3, 24, 12, 10, False, True, seed=1, tilt=0.25, push_forward=2.5)
    stage_case('ibrnet_tiny_noaa_v5', 32, 48, 5, 16, 16, 0, True, False, seed=2, aa=0, tilt=0.4)
    stage_case('ibrnet_medium', 96, 128, 4, 256, 64, 64, True, False, seed=3, tilt=0.35, store_stages=False)
    attack_case('attack_tiny', 48, 64, 4, 64, 8, 8, seed=4)
    hybrid_case('hybrid_and_pdf', 'ibrnet_tiny_invu')
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print('%-32s %8.1f KB' % (f, os.path.getsize(os.path.join(HERE, f)) / 1024.))
