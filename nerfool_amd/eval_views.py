"""Clean / adversarial evaluation loop over target views: the evaluation half of eval/ibrnet/eval.py:65-160 and
eval/ibrnet/eval_adv.py:861-905 -- render a full image through `render_single_image`, clip, PSNR and SSIM -- without
TensorFlow / LPIPS / file output.  The metrics restate the TF ops the reference calls (`tf.image.psnr`, `tf.image.ssim`:
11x11 Gaussian window, sigma 1.5, K1 0.01, K2 0.03, VALID filtering, mean over pixels then channels) in torch."""
import math

import torch
import torch.nn.functional as F

from .ibrnet.render_image import render_single_image
from .ibrnet.sample_ray import RaySamplerSingleImage


def psnr(pred, gt, max_val=1.0):
    """tf.image.psnr on [H,W,C] images: 20 log10(max) - 10 log10(mean squared error)."""
    mse = torch.mean((pred.double() - gt.double()) ** 2)
    return float(20.0 * math.log10(max_val) - 10.0 * torch.log10(mse))


def _gauss_window(size, sigma, dtype, device):
    x = torch.arange(size, dtype=dtype, device=device) - (size - 1) / 2.0
    g = torch.exp(-(x ** 2) / (2.0 * sigma ** 2))
    g = g / g.sum()
    return g[:, None] * g[None, :]


def ssim(pred, gt, max_val=1.0, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    """tf.image.ssim on [H,W,C] images (single scale)."""
    x = pred.double().permute(2, 0, 1)[None]
    y = gt.double().permute(2, 0, 1)[None]
    C = x.shape[1]
    win = _gauss_window(filter_size, filter_sigma, x.dtype, x.device)[None, None].repeat(C, 1, 1, 1)
    conv = lambda t: F.conv2d(t, win, groups=C)
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    mean0, mean1 = conv(x), conv(y)
    num0, den0 = mean0 * mean1 * 2.0, mean0 ** 2 + mean1 ** 2
    luminance = (num0 + c1) / (den0 + c1)
    num1, den1 = conv(x * y) * 2.0, conv(x ** 2 + y ** 2)
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)
    return float((luminance * cs).mean(dim=(2, 3)).mean())


def evaluate_view(args, model, projector, data, delta=None, device=None):
    """One target view: full-image render (perturbed source images when `delta` [1,V,H,W,3] is given; the clean feature maps
    are also computed when the clean-colour / clean-density ablation flags are set, eval_adv.py:868-871) -> metrics dict."""
    device = device if device is not None else next(model.feature_net.parameters()).device
    model.switch_to_eval()
    with torch.no_grad():
        sampler = RaySamplerSingleImage(data, device)
        ray_batch = sampler.get_all()
        src = ray_batch['src_rgbs']
        featmaps_clean = None
        if delta is not None:
            featmaps = model.feature_net((src + delta).squeeze(0).permute(0, 3, 1, 2))
            if getattr(args, 'use_clean_color', False) or getattr(args, 'use_clean_density', False):
                featmaps_clean = model.feature_net(src.squeeze(0).permute(0, 3, 1, 2))
        else:
            featmaps = model.feature_net(src.squeeze(0).permute(0, 3, 1, 2))
        ret = render_single_image(ray_sampler=sampler, ray_batch=ray_batch, model=model, projector=projector,
                                  chunk_size=args.chunk_size, det=True, N_samples=args.N_samples,
                                  inv_uniform=args.inv_uniform, N_importance=args.N_importance, white_bkgd=args.white_bkgd,
                                  featmaps=featmaps, args=args, featmaps_clean=featmaps_clean)
    gt = data['rgb'][0].cpu()
    out = {}
    for level in ('coarse', 'fine'):
        o = ret['outputs_' + level]
        if o is None:
            continue
        pred = o['rgb'].detach().cpu().clamp(0.0, 1.0)
        out[level + '_psnr'] = psnr(pred, gt)
        out[level + '_ssim'] = ssim(pred, gt)
    out['ret'] = ret
    return out


def evaluate_views(args, model, projector, loader, delta_for=None):
    """The loop of eval.py:65-160: running means of PSNR / SSIM over the views of `loader`; `delta_for(data)` (optional) returns
    the perturbation to apply to that view's source images."""
    sums, n = {}, 0
    per_view = []
    for data in loader:
        m = evaluate_view(args, model, projector, data, None if delta_for is None else delta_for(data))
        m.pop('ret')
        per_view.append(m)
        for k, v in m.items():
            sums[k] = sums.get(k, 0.0) + v
        n += 1
    return {'mean': {k: v / max(n, 1) for k, v in sums.items()}, 'per_view': per_view}
