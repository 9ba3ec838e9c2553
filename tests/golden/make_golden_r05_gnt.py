#!/usr/bin/env python
"""Round-5 golden vector: the OUTCOME of a whole view-specific GNT attack of the REFERENCE (gnt/ package imported from /root/reference, eval
mode as the view-specific loop runs it: eval/gnt/eval_adv.py:959 `switch_to_eval`, :967-1054 the loop, :282-339 the loss = unmasked MSE of
the single-network render, :1119 render_single_image) -- run in float32, float64 and float32 with another summation order, as
make_golden_r05.py does for the IBRNet flavour.  tests/golden/attack100_g1.npz; inputs regenerated from seeds (tests/fixtures.py).

    python tests/golden/make_golden_r05_gnt.py
Data only; build container only."""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _refimport  # noqa: E402

_refimport.install('gnt')

from gnt.feature_network import ResUNet  # noqa: E402
from gnt.projection import Projector  # noqa: E402
from gnt.render_image import render_single_image  # noqa: E402
from gnt.render_ray import render_rays  # noqa: E402
from gnt.transformer_network import GNT  # noqa: E402
import gnt.sample_ray as ref_sample_ray  # noqa: E402

from fixtures import ATTACK100, attack100_gnt_inputs, attack_outcome_stats  # noqa: E402


def clamp(X, lo, hi):      # eval/gnt/eval_adv.py:49-50
    return torch.max(torch.min(X, hi), lo)


def reference_attack(dtype, c, inputs, mkldnn=True, log=None):
    data, cnn_sd, params, delta0 = inputs
    S, R, depth = c['S'], c['N_rand'], c['depth']
    old = torch.get_default_dtype()
    cast = lambda t: t.to(dtype) if torch.is_tensor(t) and t.is_floating_point() else t
    try:
        sampler = ref_sample_ray.RaySamplerSingleImage(data, 'cpu')       # fp32 rays: inputs of every run
        torch.set_default_dtype(dtype)
        feature_net = ResUNet(coarse_out_ch=32, fine_out_ch=32, single_net=True)
        feature_net.load_state_dict(cnn_sd, strict=True)
        feature_net = feature_net.to(dtype).eval()
        net = GNT(SimpleNamespace(netwidth=64, trans_depth=depth), in_feat_ch=32, posenc_dim=63, viewenc_dim=63, ret_alpha=False)
        net.load_state_dict(params, strict=True)
        net = net.to(dtype).eval()
        model = SimpleNamespace(net_coarse=net, net_fine=None, feature_net=feature_net)
        projector = Projector(device='cpu')
        src_ray_batch = {k: cast(v) for k, v in sampler.get_all().items()}
        src = src_ray_batch['src_rgbs']
        eps = torch.tensor(c['epsilon'] / 255., dtype=dtype)
        rays_o, rays_d, rgb_all = sampler.rays_o.to(dtype), sampler.rays_d.to(dtype), cast(sampler.rgb)
        rs = np.random.RandomState(234)
        delta = delta0.to(dtype).clone().requires_grad_(True)
        opt = torch.optim.Adam([delta], lr=c['adam_lr'])
        sched = torch.optim.lr_scheduler.StepLR(opt, step_size=c['lr_step_size'], gamma=c['lr_gamma'])
        losses, pick_sum = [], 0
        t0 = time.time()
        with torch.backends.mkldnn.flags(enabled=mkldnn):
            for it in range(c['adv_iters']):
                picks = rs.choice(sampler.H * sampler.W, size=(R,), replace=False)
                pick_sum += int(picks.astype(np.int64).sum()) * (it + 1)
                sel = torch.from_numpy(picks.astype(np.int64))
                batch = {'ray_o': rays_o[sel], 'ray_d': rays_d[sel], 'rgb': rgb_all[sel], 'camera': cast(sampler.camera),
                         'depth_range': cast(sampler.depth_range), 'src_rgbs': src, 'src_cameras': cast(sampler.src_cameras),
                         'selected_inds': sel}
                featmaps = feature_net((src + delta).squeeze(0).permute(0, 3, 1, 2))
                ret = render_rays(ray_batch=batch, model=model, projector=projector, featmaps=featmaps, N_samples=S, inv_uniform=True,
                                  N_importance=0, det=True, white_bkgd=False, ret_alpha=False, args=None, src_ray_batch=src_ray_batch)
                loss = torch.mean((ret['outputs_coarse']['rgb'] - batch['rgb']) ** 2)      # gnt/criterion.py:14-20 without a mask
                opt.zero_grad()
                loss.backward()
                delta.grad.data *= -1
                opt.step()
                sched.step()
                delta.data = clamp(delta.data, -eps, eps)
                delta.data = clamp(delta.data, 0 - src, 1 - src)
                losses.append(float(loss))
                if log and (it % 20 == 0 or it + 1 == c['adv_iters']):
                    print('  %s iter %3d loss %.7f  (%.1f s)' % (log, it, losses[-1], time.time() - t0), flush=True)
            images = {}
            with torch.no_grad():
                ray_batch = {k: cast(v) for k, v in sampler.get_all().items()}
                for tag, d in (('adv', delta.data), ('clean', torch.zeros_like(delta.data))):
                    featmaps = feature_net((src + d).squeeze(0).permute(0, 3, 1, 2))
                    ret = render_single_image(ray_sampler=sampler, ray_batch=ray_batch, model=model, projector=projector,
                                              chunk_size=c['chunk_size'], det=True, N_samples=S, inv_uniform=True, N_importance=0,
                                              white_bkgd=False, featmaps=featmaps, ret_alpha=False, single_net=True, args=None,
                                              src_ray_batch=src_ray_batch)
                    images[tag] = ret['outputs_coarse']['rgb'].double().numpy()
        gt = data['rgb'][0].double().numpy()
        psnr = {k: float(-10. * np.log10(np.mean((v - gt) ** 2))) for k, v in images.items()}
        return dict(losses=np.array(losses), delta=delta.data.double().numpy().copy(), image=images['adv'], psnr=psnr['adv'],
                    psnr_clean=psnr['clean'], pick_sum=pick_sum)
    finally:
        torch.set_default_dtype(old)


if __name__ == '__main__':
    torch.set_num_threads(8)
    tag = 'g1'
    c = ATTACK100[tag]
    inputs = attack100_gnt_inputs(c)
    eps = c['epsilon'] / 255.
    t0 = time.time()
    r32 = reference_attack(torch.float32, c, inputs, log='g1 ref32')
    r64 = reference_attack(torch.float64, c, inputs, log='g1 ref64')
    alt = reference_attack(torch.float32, c, inputs, mkldnn=False, log='g1 alt32')
    assert r32['pick_sum'] == r64['pick_sum'] == alt['pick_sum']
    out = {'cfg_tag': np.array(tag), 'pick_checksum': np.array(r32['pick_sum'], dtype=np.int64)}
    for name, r in (('ref32', r32), ('ref64', r64), ('alt32', alt)):
        out[name + '/losses'] = r['losses']
        out[name + '/psnr'] = np.array(r['psnr'])
        out[name + '/psnr_clean'] = np.array(r['psnr_clean'])
        out[name + '/frac_at_eps'] = np.array(float((np.abs(r['delta']) >= eps * (1 - 1e-5)).mean()))
    for name, r in (('ref32', r32), ('ref64', r64)):
        out[name + '/delta'] = r['delta'].reshape(-1)[::c['delta_stride']].astype(np.float32)
        out[name + '/image'] = r['image'].astype(np.float32)
    for name, a, b in (('ref32_vs_ref64', r32, r64), ('alt32_vs_ref64', alt, r64), ('ref32_vs_alt32', r32, alt)):
        s = attack_outcome_stats(a, b, eps)
        for k, v in s.items():
            out['floor/%s/%s' % (name, k)] = np.array(v)
        print('%s floor %-15s %s' % (tag, name, '  '.join('%s %.3e' % kv for kv in sorted(s.items()))), flush=True)
    print('%s: PSNR clean %.3f dB | attacked ref32 %.3f ref64 %.3f alt32 %.3f | at +-eps: %.4f %.4f %.4f | losses first/last %.5f -> %.5f'
          % (tag, r64['psnr_clean'], r32['psnr'], r64['psnr'], alt['psnr'], out['ref32/frac_at_eps'], out['ref64/frac_at_eps'],
             out['alt32/frac_at_eps'], r64['losses'][0], r64['losses'][-1]))
    path = os.path.join(HERE, 'attack100_%s.npz' % tag)
    np.savez_compressed(path, **out)
    print('%s %.1f KB  (%.0f s)' % (path, os.path.getsize(path) / 1024., time.time() - t0), flush=True)
