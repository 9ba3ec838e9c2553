"""Diagnostic (GPU): gradient of the attack loss w.r.t. delta on the golden attack case, matrix-core vs generic IBRNet
forward, against the reference's gradient; also compares the re-sampled fine depths of the two paths."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import parity_cases as pc
from nerfool_amd import eval_adv as EA
from nerfool_amd.ibrnet import mlp_network
from nerfool_amd.ibrnet.projection import Projector
from nerfool_amd.ibrnet.render_ray import render_rays

dev = 'cuda'
g, args, model, data, sampler, dims = pc._attack_setup(dev)
src_ray_batch = sampler.get_all()
picks = g.np('adam/selected_inds')
ref = g.np('adam/grad_iter0')
res = {}
for path in ('auto', 'generic', 'auto'):
    mlp_network.KERNEL_PATH = path
    a = EA.PGDAttack(args, model, Projector(dev), src_ray_batch, delta=g.t('in/delta0', dev).clone().requires_grad_(True))
    grad = a.gradient(data, select_inds=picks[0]).cpu().numpy()
    err = np.linalg.norm(grad - ref) / np.linalg.norm(ref)
    bad = (np.abs(grad - ref) > 2e-3 * np.abs(ref).max() + 1e-2 * np.abs(ref)).mean()
    print('%-8s loss %.7f (ref %.7f)  rel-L2 grad err %.3e  frac off %.4f' % (path, float(a.last_loss), g.np('adam/losses')[0], err, bad))
    with torch.no_grad():
        fm = model.feature_net((src_ray_batch['src_rgbs'] + g.t('in/delta0', dev)).squeeze(0).permute(0, 3, 1, 2))
        ret = render_rays(sampler.select(picks[0]), model, fm, Projector(dev), args.N_samples, inv_uniform=True,
                          N_importance=args.N_importance, det=True, src_ray_batch=src_ray_batch)
    res[path] = (grad, ret['outputs_fine']['z_vals'].cpu().numpy(), ret['outputs_coarse']['weights'].cpu().numpy())
za, zg = res['auto'][1], res['generic'][1]
print('fine z differing > 1e-4 between paths:', int((np.abs(za - zg) > 1e-4).sum()), 'of', za.size,
      ' rays affected:', int((np.abs(za - zg) > 1e-4).any(1).sum()))
print('coarse weights max diff between paths: %.3e' % np.abs(res['auto'][2] - res['generic'][2]).max())
print('grad rel-L2 between paths: %.3e' % (np.linalg.norm(res['auto'][0] - res['generic'][0]) / np.linalg.norm(ref)))
