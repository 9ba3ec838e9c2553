"""Diagnostic (GPU): per-stage error budget of d loss / d delta against the reference's float64 ground truth
(tests/golden/attack_grad64.npz, made by tests/golden/make_golden_grad64.py from the imported reference).

For the tiny (48x64, 8+8 samples) and the medium (96x128, 64+64 samples) case it prints, as relative L2 distances to
the float64 result, next to the reference's own fp32 floor:
  whole step      grad(delta0) through CNN fwd -> render -> loss -> render bwd -> CNN bwd
  CNN forward     feature maps at delta0
  render bwd      d loss / d feature maps with the float64 feature maps as the (leaf) input
  CNN backward    the CNN's vector-Jacobian product at delta0 with the float64 d feature maps as upstream gradient
and repeats the rows with one stage swapped for another implementation (plain nn.Module graph on MIOpen, MIOpen 3x3
convolutions inside the fused executor, shape-generic IBRNet kernels, deterministic scatter).

    python tools/diag_grad_budget.py [tiny|medium|both] [--json out.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch

import parity_cases as pc
from fixtures import Golden
from nerfool_amd import eval_adv as EA
from nerfool_amd.ibrnet import feature_network, mlp_network
from nerfool_amd.ibrnet.projection import Projector
from nerfool_amd.ibrnet.render_ray import render_rays

dev = 'cuda'


def rel(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def stages(case, label):
    g64, args, model, data, sampler, delta0, picks = pc.grad64_setup(case, dev)
    tag = case + '/'
    src = sampler.get_all()
    out = {}
    atk = EA.PGDAttack(args, model, Projector(dev), src, delta=delta0.clone().requires_grad_(True))
    grad = atk.gradient(data, select_inds=picks, lookahead=False)
    out['whole step'] = rel(grad, g64.np(tag + 'grad64'))
    out['vs reference fp32'] = rel(grad, g64.np(tag + 'grad32'))
    out['loss rel err'] = abs(float(atk.last_loss) - float(g64.np(tag + 'loss64'))) / float(g64.np(tag + 'loss64'))
    # CNN forward
    d = delta0.clone().requires_grad_(True)
    x = (src['src_rgbs'] + d).squeeze(0).permute(0, 3, 1, 2)
    fc, ff = model.feature_net(x)
    fm = torch.cat([fc, ff], 1)
    out['CNN forward'] = rel(fm, g64.np(tag + 'fm64'))
    # CNN backward: VJP with the float64 upstream gradient
    up = g64.t(tag + 'dfm64', dev)
    gd, = torch.autograd.grad([fc, ff], d, [up[:, :32], up[:, 32:]])
    out['CNN backward'] = rel(gd, g64.np(tag + 'grad64'))
    # render backward at the float64 feature maps
    fm64 = g64.t(tag + 'fm64', dev)
    f0 = fm64[:, :32].contiguous(memory_format=torch.channels_last).requires_grad_(True)
    f1 = fm64[:, 32:].contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rb = sampler.select(picks)
    ret = render_rays(rb, model, (f0, f1), Projector(dev), args.N_samples, inv_uniform=True, N_importance=args.N_importance,
                      det=True, src_ray_batch=src)
    loss = EA.criterion(ret['outputs_coarse'], rb)[0] + EA.criterion(ret['outputs_fine'], rb)[0]
    g0, g1 = torch.autograd.grad(loss, [f0, f1])
    out['render bwd'] = rel(torch.cat([g0, g1], 1), g64.np(tag + 'dfm64'))
    floors = {k: float(g64.np(tag + 'floor/' + k)) for k in ('fm', 'dfm', 'grad')}
    print('%-7s %-34s whole %.2e (ref32 floor %.2e; vs ref32 %.2e) | CNN fwd %.2e (%.2e) | render bwd %.2e (%.2e) | '
          'CNN bwd %.2e | loss %.1e' % (case, label, out['whole step'], floors['grad'], out['vs reference fp32'],
                                        out['CNN forward'], floors['fm'], out['render bwd'], floors['dfm'],
                                        out['CNN backward'], out['loss rel err']), flush=True)
    out['floors'] = floors
    return out


def main():
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('--') else 'both'
    cases = ('tiny', 'medium') if which == 'both' else (which,)
    table = {}
    variants = [('product path', {})]
    variants += [('nn.Module graph (ATen + MIOpen)', {'cnn': 'torch'}),
                 ('fused CNN, Winograd 3x3 forced', {'conv3x3': 'wino'}),
                 ('generic IBRNet kernels', {'ibr': 'generic'})]
    for case in cases:
        for label, v in variants:
            saved = (feature_network.CNN_PATH, feature_network.CONV3X3, mlp_network.KERNEL_PATH)
            feature_network.CNN_PATH = v.get('cnn', saved[0])
            feature_network.CONV3X3 = v.get('conv3x3', saved[1])
            mlp_network.KERNEL_PATH = v.get('ibr', saved[2])
            try:
                table['%s | %s' % (case, label)] = stages(case, label)
            finally:
                feature_network.CNN_PATH, feature_network.CONV3X3, mlp_network.KERNEL_PATH = saved
    if '--json' in sys.argv:
        with open(sys.argv[sys.argv.index('--json') + 1], 'w') as f:
            json.dump(table, f, indent=1)


if __name__ == '__main__':
    main()
