"""Clean / adversarial evaluation and frame-rendering loops over target views: the evaluation half of
eval/ibrnet/eval.py:65-160, eval/ibrnet/eval_adv.py:861-905 and eval/gnt/eval.py:41-236 (`evaluate_view(s)`: render a full
image through `render_single_image` of the model's flavour, clip, PSNR and SSIM) and the frame loop of
eval/ibrnet/render_llff_video.py:156-223 / eval/gnt/render.py:41-98 (`render_frames`: a list of camera batches -> cropped
uint8 frames, depth and accumulation maps) -- without TensorFlow / LPIPS / image or video file output.  The metrics follow the
flavour's own definitions: the IBRNet scripts call the TF ops (`tf.image.psnr`, `tf.image.ssim`: 11x11 Gaussian window, sigma
1.5, K1 0.01, K2 0.03, VALID filtering, mean over pixels then channels -- eval/ibrnet/eval.py:130-141), the GNT scripts their
own torch functions (eval/gnt/utils.py:29,55-71 `img2psnr` = -10 log10(mse + 1e-6) on the fp32 mean; :211-277 `ssim`: the same
window zero-padded to SAME size, one mean over the whole map, fp32).  `evaluate_view` picks by the model's flavour."""
import math

import torch
import torch.nn.functional as F

from .ibrnet.render_image import render_single_image
from .ibrnet.sample_ray import RaySamplerSingleImage


def psnr(pred, gt, max_val=1.0, tiny=0.0):
    """tf.image.psnr on [H,W,C] images: 20 log10(max) - 10 log10(mean squared error).  tiny=1e-6: the GNT flavour's
    `img2psnr` (eval/gnt/utils.py:29 `mse2psnr(x) = -10 log(x + TINY_NUMBER) / log 10` on the fp32 mean of :61)."""
    if tiny:
        mse = float(torch.mean((pred - gt) * (pred - gt)))
        return float(20.0 * math.log10(max_val) - 10.0 * math.log(mse + tiny) / math.log(10.0))
    mse = torch.mean((pred.double() - gt.double()) ** 2)
    return float(20.0 * math.log10(max_val) - 10.0 * torch.log10(mse))


def _gauss_window(size, sigma, dtype, device):
    x = torch.arange(size, dtype=dtype, device=device) - (size - 1) / 2.0
    g = torch.exp(-(x ** 2) / (2.0 * sigma ** 2))
    g = g / g.sum()
    return g[:, None] * g[None, :]


def ssim(pred, gt, max_val=1.0, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03, padding='valid'):
    """padding='valid': tf.image.ssim on [H,W,C] images (single scale, float64).  padding='same': the GNT flavour's `ssim`
    (eval/gnt/utils.py:211-235,269-277: windows zero-padded by filter_size // 2, variances as E[x^2] - mu^2, one mean over
    all channels and pixels, evaluated in the images' own precision like the reference)."""
    if padding == 'same':
        x = pred.permute(2, 0, 1)[None]
        y = gt.permute(2, 0, 1)[None].to(x.dtype)
        C = x.shape[1]
        win = _gauss_window(filter_size, filter_sigma, torch.float32, x.device).to(x.dtype)[None, None].repeat(C, 1, 1, 1)
        conv = lambda t: F.conv2d(t, win, padding=filter_size // 2, groups=C)
        c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
        mu1, mu2 = conv(x), conv(y)
        s11, s22, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
        return float((((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))).mean())
    assert padding == 'valid'
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


def _is_gnt(model):
    from .gnt.transformer_network import GNT
    return isinstance(getattr(model, 'net_coarse', None), GNT)


def render_view(args, model, projector, data, delta=None, device=None, shard=None):
    """Full-image render of one target view with the model's own flavour of render_single_image (perturbed source images when
    `delta` [1,V,H,W,3] is given; the clean feature maps are also computed when the clean-colour / clean-density ablation
    flags are set, eval_adv.py:868-871).  -> (ret dict of CPU tensors, sampler).  shard: optional `eval_adv.RayShard` -- the
    image's chunks are rendered by all ranks (every rank computes the feature maps) and `ret` is None except on the gathering
    rank."""
    device = device if device is not None else next(model.feature_net.parameters()).device
    model.switch_to_eval()
    stride = getattr(args, 'render_stride', 1)
    with torch.no_grad():
        sampler = RaySamplerSingleImage(data, device, render_stride=stride)
        ray_batch = sampler.get_all()
        src = ray_batch['src_rgbs']
        featmaps_clean = None
        if delta is not None:
            featmaps = model.feature_net((src + delta).squeeze(0).permute(0, 3, 1, 2))
            if getattr(args, 'use_clean_color', False) or getattr(args, 'use_clean_density', False):
                featmaps_clean = model.feature_net(src.squeeze(0).permute(0, 3, 1, 2))
        else:
            featmaps = model.feature_net(src.squeeze(0).permute(0, 3, 1, 2))
        kw = dict(ray_sampler=sampler, ray_batch=ray_batch, model=model, projector=projector, chunk_size=args.chunk_size, det=True,
                  N_samples=args.N_samples, inv_uniform=args.inv_uniform, N_importance=args.N_importance,
                  white_bkgd=args.white_bkgd, render_stride=stride, featmaps=featmaps, args=args, featmaps_clean=featmaps_clean,
                  shard=shard)
        if _is_gnt(model):      # eval/gnt/eval.py:153-176 (log_view)
            from .gnt.render_image import render_single_image as gnt_render_single_image
            ret = gnt_render_single_image(ret_alpha=getattr(args, 'ret_alpha', False), single_net=getattr(args, 'single_net', True), **kw)
        else:
            ret = render_single_image(**kw)
    return ret, sampler


def evaluate_view(args, model, projector, data, delta=None, device=None, shard=None):
    """One target view -> metrics dict ('coarse_psnr', 'coarse_ssim', and the fine pair when there is a fine level; the GNT
    flavour reports the level its log_view scores: fine if present, else coarse -- eval/gnt/eval.py:223-231), each flavour
    with its own metric definitions (module docstring).  With `shard`: None on the ranks that do not receive the image."""
    ret, sampler = render_view(args, model, projector, data, delta, device, shard)
    if ret is None:
        return None
    gnt = _is_gnt(model)
    stride = getattr(args, 'render_stride', 1)
    gt = data['rgb'][0].cpu()[::stride, ::stride]
    out = {}
    for level in ('coarse', 'fine'):
        o = ret['outputs_' + level]
        if o is None:
            continue
        pred = o['rgb'].detach().cpu().clamp(0.0, 1.0)
        out[level + '_psnr'] = psnr(pred, gt, tiny=1e-6 if gnt else 0.0)
        out[level + '_ssim'] = ssim(pred, gt, padding='same' if gnt else 'valid')
    out['ret'] = ret
    return out


def render_frames(args, model, projector, frames, delta=None, crop_ratio=0.075, device=None):
    """The frame loop of render_llff_video.py:156-223: every element of `frames` is a loader-style batch of one target camera
    (no ground-truth image needed) with its source views.  Per frame: the 8-bit colour image of the finest level cropped by
    `crop_ratio` at every border (what the reference appends to the video), and per level the uncropped 8-bit colour, the
    depth map and the accumulation map (sum of the weights; None for GNT without ret_alpha)."""
    out = []
    for data in frames:
        ret, sampler = render_view(args, model, projector, data, delta, device)
        frame = {}
        for level in ('coarse', 'fine'):
            o = ret['outputs_' + level]
            if o is None:
                continue
            rgb8 = (255 * o['rgb'].detach().cpu().numpy().clip(0.0, 1.0)).astype('uint8')
            w = o.get('weights')
            frame[level] = {'rgb8': rgb8, 'depth': o.get('depth'), 'acc': None if w is None else w.sum(dim=-1)}
        best = frame['fine' if 'fine' in frame else 'coarse']['rgb8']
        h, w_ = best.shape[:2]
        ch, cw = int(h * crop_ratio), int(w_ * crop_ratio)
        frame['video_frame'] = best[ch:h - ch, cw:w_ - cw, :]
        out.append(frame)
    return out


def evaluate_views(args, model, projector, loader, delta_for=None, shard=None):
    """The loop of eval.py:65-160: running means of PSNR / SSIM over the views of `loader`; `delta_for(data)` (optional) returns
    the perturbation to apply to that view's source images.  shard: every view is rendered by all ranks (ray-range sharding); the
    metrics exist on the rank that receives the images, the others return None."""
    sums, n = {}, 0
    per_view = []
    for data in loader:
        m = evaluate_view(args, model, projector, data, None if delta_for is None else delta_for(data), shard=shard)
        if m is None:
            continue
        m.pop('ret')
        per_view.append(m)
        for k, v in m.items():
            sums[k] = sums.get(k, 0.0) + v
        n += 1
    if shard is not None and shard.world > 1 and shard.gather_render_to is not None and shard.rank != shard.gather_render_to:
        return None
    return {'mean': {k: v / max(n, 1) for k, v in sums.items()}, 'per_view': per_view}
