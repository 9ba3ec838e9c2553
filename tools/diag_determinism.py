"""Diagnostic (GPU): run-to-run bitwise reproducibility of the forward path, stage by stage (feature CNN, full render)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import parity_cases as pc
from nerfool_amd.ibrnet import feature_network as fn
from nerfool_amd.ibrnet.projection import Projector
from nerfool_amd.ibrnet.render_ray import render_rays

dev = 'cuda'
g, args, model, data, sampler, dims = pc._attack_setup(dev)
src = sampler.get_all()
x = (src['src_rgbs'] + g.t('in/delta0', dev)).squeeze(0).permute(0, 3, 1, 2)
outs = []
with torch.no_grad():
    for it in range(4):
        fn.TRACE_RELU = tr = []
        c, f = model.feature_net(x)
        fn.TRACE_RELU = None
        outs.append((c.clone(), f.clone(), [t.clone() for t in tr]))
for it in range(1, 4):
    dc = float((outs[it][0] - outs[0][0]).abs().max())
    first = next((i for i, (a, b) in enumerate(zip(outs[it][2], outs[0][2])) if not torch.equal(a, b)), None)
    print('CNN forward pass %d vs 0: max |diff| %.3e, first differing ReLU layer: %s' % (it, dc, first))
# the two Winograd workgroup widths must agree bit for bit (the per-layer choice between them is timed)
res = {}
with torch.no_grad():
    for mode in ('wino', 'wino32'):
        fn.CONV3X3 = mode
        res[mode] = model.feature_net(x)[0].clone()
    fn.CONV3X3 = 'auto'
print('wino vs wino32 feature maps: max |diff| %.3e' % float((res['wino'] - res['wino32']).abs().max()))
print('conv choices:', {str(k): v for k, v in fn._CONV_CHOICE.items()})
rb = sampler.get_all()
chunk = {k: (v[:1000] if k in ('ray_o', 'ray_d', 'rgb') else v) for k, v in rb.items()}
with torch.no_grad():
    fm = (outs[0][0], outs[0][1])
    rs = [render_rays(chunk, model, fm, Projector(dev), args.N_samples, inv_uniform=True, N_importance=args.N_importance, det=True,
                      src_ray_batch=src) for _ in range(3)]
for it in (1, 2):
    for level in ('outputs_coarse', 'outputs_fine'):
        for k in ('rgb', 'weights', 'z_vals'):
            d = float((rs[it][level][k] - rs[0][level][k]).abs().max())
            if d:
                print('render pass %d %s %s max |diff| %.3e' % (it, level, k, d))
print('render compared')
