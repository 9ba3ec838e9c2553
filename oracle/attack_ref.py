"""Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): the PGD / Adam-ascent perturbation loop on the
source images, PyTorch-CPU fp32.

ref: eval/ibrnet/eval_adv.py:28-29 (clamp), :248-254 (init_adv_perturb), :258-310,512-519
(optimize_adv_perturb, rgb-loss path), :783-843 (view-specific loop), :634-740 (universal loop).
"""
import numpy as np
import torch

from . import feature_net_ref as fnet
from . import ibrnet_ref as ib


def clamp(X, lower_limit, upper_limit):
    """ref: eval/ibrnet/eval_adv.py:28-29 -- max(min(X, hi), lo) with tensor or scalar bounds."""
    lo = lower_limit if torch.is_tensor(lower_limit) else torch.tensor(lower_limit, dtype=X.dtype)
    hi = upper_limit if torch.is_tensor(upper_limit) else torch.tensor(upper_limit, dtype=X.dtype)
    return torch.max(torch.min(X, hi), lo)


def init_adv_perturb(src_rgbs, epsilon, upper_limit=1.0, lower_limit=0.0, generator=None):
    """ref: eval/ibrnet/eval_adv.py:248-254: delta ~ U(-eps, eps), then projected so that src+delta in [0,1]."""
    delta = torch.zeros_like(src_rgbs)
    delta.uniform_(-float(epsilon), float(epsilon), generator=generator)
    delta = clamp(delta, lower_limit - src_rgbs, upper_limit - src_rgbs)
    return delta.requires_grad_(True)


def attack_loss(delta, cnn_state, params_coarse, params_fine, src_ray_batch, train_ray_batch, cfg, cnn_trace=None):
    """ref: eval/ibrnet/eval_adv.py:292-310 -- features from PERTURBED images, colours from CLEAN images.
    cfg['use_pseudo_gt'] (ref :271-290): the target colours are the fine-level render from the CLEAN source images (no grad).
    cnn_trace: optional feature_net_ref.ReluTrace (evaluate the CNN on a given ReLU activation pattern / record its own)."""
    if cfg.get('use_pseudo_gt', False):
        with torch.no_grad():
            clean = fnet.resunet_forward(cnn_state, src_ray_batch['src_rgbs'].squeeze(0).permute(0, 3, 1, 2))
            ret_gt = ib.render_rays(train_ray_batch, params_coarse, params_fine, clean, cfg['N_samples'],
                                    inv_uniform=cfg['inv_uniform'], N_importance=cfg['N_importance'], det=True,
                                    white_bkgd=cfg.get('white_bkgd', False), src_ray_batch=src_ray_batch,
                                    anti_alias_pooling=cfg.get('anti_alias_pooling', True))
        train_ray_batch = dict(train_ray_batch, rgb=ret_gt['outputs_fine']['rgb'], depth=ret_gt['outputs_fine']['depth'])
    imgs = (src_ray_batch['src_rgbs'] + delta).squeeze(0).permute(0, 3, 1, 2)
    featmaps = fnet.resunet_forward(cnn_state, imgs, trace=cnn_trace)
    ret = ib.render_rays(train_ray_batch, params_coarse, params_fine, featmaps, cfg['N_samples'],
                         inv_uniform=cfg['inv_uniform'], N_importance=cfg['N_importance'], det=True,
                         white_bkgd=cfg.get('white_bkgd', False), src_ray_batch=src_ray_batch,
                         anti_alias_pooling=cfg.get('anti_alias_pooling', True))
    loss = ib.criterion(ret['outputs_coarse'], train_ray_batch)
    if ret['outputs_fine'] is not None:
        loss = loss + ib.criterion(ret['outputs_fine'], train_ray_batch)
    return loss, ret


class AdamAscent:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) applied to g = -dL/d(delta), with StepLR(step, gamma),
    written out explicitly (ref: eval/ibrnet/eval_adv.py:789-790, 805-819)."""

    def __init__(self, shape, lr, step_size=100, gamma=0.5):
        self.m = torch.zeros(shape)
        self.v = torch.zeros(shape)
        self.t = 0
        self.lr0, self.step_size, self.gamma = lr, step_size, gamma

    def lr(self):
        return self.lr0 * self.gamma ** (self.t // self.step_size)

    def step(self, delta, grad):
        """Same operation order as torch.optim.Adam's single-tensor path (lerp_, mul_/addcmul_, addcdiv_)."""
        g = -grad
        lr = self.lr()
        self.t += 1
        self.m = self.m + (g - self.m) * (1 - 0.9)
        self.v = self.v * 0.999 + (1 - 0.999) * (g * g)
        bc1 = 1.0 - 0.9 ** self.t
        bc2 = 1.0 - 0.999 ** self.t
        denom = self.v.sqrt() / (bc2 ** 0.5) + 1e-8
        return delta + (-(lr / bc1)) * (self.m / denom)


def project(delta, src_rgbs, epsilon, upper_limit=1.0, lower_limit=0.0):
    """ref: eval/ibrnet/eval_adv.py:838-839 -- eps-ball, then the [0,1] image box."""
    delta = clamp(delta, -epsilon, epsilon)
    return clamp(delta, lower_limit - src_rgbs, upper_limit - src_rgbs)


def pgd_attack(delta0, cnn_state, params_coarse, params_fine, src_ray_batch, ray_batches, cfg, n_iters,
               use_adam=True, adam_lr=1e-3, lr_step_size=100, lr_gamma=0.5, adv_lr=2.0, epsilon=8.0,
               record=()):
    """View-specific loop (ref :796-843): `ray_batches` is a callable it -> train_ray_batch (already-sampled rays).
    Returns final delta, list of losses, and snapshots {it: delta} for iterations in `record`."""
    eps = epsilon / 255.0
    alpha = adv_lr / 255.0
    delta = delta0.detach().clone()
    opt = AdamAscent(delta.shape, adam_lr, lr_step_size, lr_gamma) if use_adam else None
    losses, snaps = [], {}
    src = src_ray_batch['src_rgbs']
    for it in range(n_iters):
        d = delta.clone().requires_grad_(True)
        loss, _ = attack_loss(d, cnn_state, params_coarse, params_fine, src_ray_batch, ray_batches(it), cfg)
        grad, = torch.autograd.grad(loss, d)
        losses.append(float(loss.detach()))
        if use_adam:
            delta = opt.step(delta, grad)
        else:
            delta = delta + alpha * torch.sign(grad)
        delta = project(delta, src, eps)
        if (it + 1) in record:
            snaps[it + 1] = delta.clone()
    return delta, losses, snaps, opt


def pick_pixels(rng, n_pixels, n_rand):
    """ref: ibrnet/sample_ray.py:146-148 -- rng.choice(H*W, N_rand, replace=False) on RandomState(234)."""
    return rng.choice(n_pixels, size=(n_rand,), replace=False)


def new_pixel_rng():
    """ref: ibrnet/sample_ray.py:20."""
    return np.random.RandomState(234)


def float64_gradient(delta, cnn_state, params_coarse, params_fine, data, picks, cfg, relu_masks=None):
    """Ground truth for gradient-parity checks: d loss / d delta of attack_loss evaluated in float64 from the fp32 inputs
    (`data` = loader-style batch dict of the target view, `picks` = flat pixel indices of the drawn rays; the rays are the
    fp32 rays of ibrnet_ref.rays_single_image, an input of the step like the images).  Pinned against the reference's own
    float64 evaluation by tests/test_oracle_golden.py (tests/golden/attack_grad64.npz).
    relu_masks: evaluate the CNN on this ReLU activation pattern (feature_net_ref.ReluTrace) instead of its own.
    -> (loss float, grad float64 tensor, ReluTrace with the float64 ReLU arguments)"""
    f64 = lambda t: t.detach().cpu().double() if torch.is_tensor(t) and t.is_floating_point() else t
    cam = data['camera'].detach().cpu().float()
    H, W = int(cam[0, 0]), int(cam[0, 1])
    ro, rd = ib.rays_single_image(H, W, cam[:, 2:18].reshape(-1, 4, 4), cam[:, 18:34].reshape(-1, 4, 4))
    idx = torch.as_tensor(np.asarray(picks, dtype=np.int64))
    src = {'src_rgbs': f64(data['src_rgbs']), 'src_cameras': f64(data['src_cameras'])}
    batch = {'ray_o': f64(ro[idx]), 'ray_d': f64(rd[idx]), 'rgb': f64(data['rgb'].detach().cpu().reshape(-1, 3)[idx]),
             'camera': f64(cam), 'depth_range': f64(data['depth_range']), 'src_rgbs': src['src_rgbs'],
             'src_cameras': src['src_cameras']}
    trace = fnet.ReluTrace(relu_masks)
    d = f64(delta).clone().requires_grad_(True)
    loss, _ = attack_loss(d, {k: f64(v) for k, v in cnn_state.items()}, {k: f64(v) for k, v in params_coarse.items()},
                          {k: f64(v) for k, v in params_fine.items()}, src, batch, cfg, cnn_trace=trace)
    grad, = torch.autograd.grad(loss, d)
    return float(loss.detach()), grad, trace


def relu_pattern_flips(trace64, masks):
    """Units whose ReLU decision in `masks` differs from the float64 evaluation recorded in `trace64`:
    -> (number of flipped units, number of units, largest |argument| of a flipped unit relative to its plane's rms).
    A flip is legitimate rounding behaviour only where the float64 argument is within fp32 noise of zero."""
    n_flip, n_all, worst = 0, 0, 0.0
    for pre, m in zip(trace64.pre, masks):
        m = m.to(pre.device)
        diff = (pre > 0) != m
        n_all += pre.numel()
        k = int(diff.sum())
        if k:
            rms = pre.pow(2).mean(dim=(2, 3), keepdim=True).sqrt().expand_as(pre)
            worst = max(worst, float((pre.abs() / rms)[diff].max()))
            n_flip += k
    return n_flip, n_all, worst


def unseen_camera_stream(args, render_poses, camera, n_steps, seed):
    """The target cameras the reference's universal loop visits under --use_unseen_views (eval/ibrnet/eval_adv.py:652-691):
    per step, from numpy's global generator in this order -- three distinct render-pose indices (softmax of the poses' forward
    z over args.temp as probabilities with args.sample_based_on_depth :655-661), then the interpolation parameters (decoupled:
    two uniforms for rotation, two for translation :668-672; depth-based: two Beta(beta, beta) draws scaled by
    interp_upbound_rot :676; else two uniforms in [0, interp_upbound] :678) -- then interp3 (eval/ibrnet/geo_interp.py:44-45;
    pinned by tests/golden/metrics_r03.npz) replaces elements 18..33 of the [1,34] camera.  -> list of [1,34] tensors."""
    import numpy as np
    from scipy.spatial.transform import Rotation

    def slerp(p0, p1, t):
        omega = np.arccos(np.dot(p0 / np.linalg.norm(p0), p1 / np.linalg.norm(p1)))
        return np.sin((1.0 - t) * omega) / np.sin(omega) * p0 + np.sin(t * omega) / np.sin(omega) * p1

    def interp(a, b, s):
        s_rot, s_trans = (s[0], s[1]) if type(s) == list else (s, s)
        m = np.eye(4)
        m[:3, 3] = (1 - s_trans) * a[:-1, -1] + s_trans * b[:-1, -1]
        m[:3, :3] = Rotation.from_quat(slerp(Rotation.from_matrix(a[:3, :3]).as_quat(), Rotation.from_matrix(b[:3, :3]).as_quat(),
                                             s_rot)).as_matrix()
        return m

    rs = np.random.RandomState(seed)
    poses = [np.asarray(p, dtype=np.float64) for p in render_poses]
    out = []
    for _ in range(n_steps):
        if getattr(args, 'sample_based_on_depth', False):
            z = np.array([p[2, 2] for p in poses])
            ids = rs.choice(len(poses), size=3, p=np.exp(z / args.temp) / np.sum(np.exp(z / args.temp)), replace=False)
        else:
            ids = rs.choice(len(poses), size=3, replace=False)
        if getattr(args, 'decouple_interp_range', False):
            s12_rot, s3_rot = rs.uniform(0, args.interp_upbound_rot, size=2)
            s12_trans, s3_trans = rs.uniform(0, args.interp_upbound_trans, size=2)
            s12, s3 = [s12_rot, s12_trans], [s3_rot, s3_trans]
        elif getattr(args, 'sample_based_on_depth', False):
            s12, s3 = rs.beta(args.beta, args.beta, size=2) * args.interp_upbound_rot
        else:
            s12, s3 = rs.uniform(0, args.interp_upbound, size=2)
        pose = torch.from_numpy(interp(interp(poses[ids[0]], poses[ids[1]], s12), poses[ids[2]], s3))
        out.append(torch.cat([camera[:, :18], pose.flatten().unsqueeze(0).to(camera)], dim=1))
    return out
