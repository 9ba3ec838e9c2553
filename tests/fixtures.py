"""Loader for the committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py)."""
import os
from collections import OrderedDict

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
STAGE_CASES = ['ibrnet_tiny_invu', 'ibrnet_tiny_lin_white', 'ibrnet_tiny_noaa_v5', 'ibrnet_tiny_v10', 'ibrnet_medium']
# end-to-end captures only (no per-stage tensors): the medium case and the BASELINE-config-5-shaped one (V = 8, 128 + 128 samples)
END_TO_END_ONLY = ['ibrnet_medium', 'ibrnet_c5_v8']


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + '.npz'))

    def __contains__(self, k):
        return k in self.z.files

    def np(self, k):
        return self.z[k]

    def t(self, k, device='cpu'):
        a = self.z[k]
        return torch.from_numpy(np.array(a, order='C')).to(device)      # np.array keeps 0-d shapes

    def params(self, prefix, device='cpu'):
        if prefix + '_seed' in self.z.files:
            # parameters regenerated from their seed (the generator script made the same call in front of the reference) and
            # checked against the fingerprint stored with the capture
            from oracle.gnt_ref import random_gnt_params
            depth, seed = [int(x) for x in self.z[prefix + '_seed']]
            p = random_gnt_params(depth, seed=seed)
            s1 = sum(float(v.double().sum()) for _, v in sorted(p.items()))
            s2 = sum(float((v.double() ** 2).sum()) for _, v in sorted(p.items()))
            want = self.z[prefix + '_checksum']
            assert abs(s1 - want[0]) <= 1e-9 * max(1.0, abs(want[0])) and abs(s2 - want[1]) <= 1e-9 * abs(want[1]), \
                'regenerated parameters do not match the fingerprint of the capture (%r vs %r)' % ((s1, s2), tuple(want))
            return OrderedDict((k, v.to(device)) for k, v in p.items())
        out = OrderedDict()
        for k in self.z.files:
            name = k[len(prefix) + 1:]
            if k.startswith(prefix + '/') and ('.' in name or name in ('s', 'pos_encoding')):
                out[name] = self.t(k, device)
        return out or None

    def stage_cfg(self):
        H, W, V, R, S, N_imp, inv_u, white, aa, Hf, Wf = [int(x) for x in self.z['cfg']]
        return dict(H=H, W=W, V=V, R=R, S=S, N_importance=N_imp, inv_uniform=bool(inv_u), white_bkgd=bool(white),
                    anti_alias_pooling=bool(aa), Hf=Hf, Wf=Wf)

    def ray_batch(self, device='cpu'):
        return {'ray_o': self.t('in/ray_o', device), 'ray_d': self.t('in/ray_d', device),
                'rgb': self.t('in/gt_rgb', device), 'camera': self.t('in/camera', device),
                'depth_range': self.t('in/depth_range', device), 'src_rgbs': self.t('in/src_rgbs', device),
                'src_cameras': self.t('in/src_cameras', device)}


def assert_close(a, b, rtol=1e-4, atol=1e-5, name='', frac_ok=0.0):
    """|a-b| <= atol + rtol*|b| elementwise; `frac_ok` tolerates that fraction of outliers (discrete flips)."""
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, '%s: shape %s vs %s' % (name, a.shape, b.shape)
    bad = np.abs(a - b) > atol + rtol * np.abs(b)
    if bad.mean() > frac_ok:
        i = np.unravel_index(np.argmax(np.abs(a - b) - rtol * np.abs(b)), a.shape) if a.ndim else ()
        raise AssertionError('%s: %d/%d elements off (max abs err %.3e at %s: %r vs %r)'
                             % (name, bad.sum(), bad.size, np.abs(a - b).max(), i, a[i], b[i]))


# the medium case of tests/golden/attack_grad64.npz: every input is regenerated from seeds (only picks + outputs are stored)
GRAD64_MEDIUM = dict(H=96, W=128, V=4, R=256, S=64, N_imp=64, seed=6)


def grad64_medium_inputs(c=GRAD64_MEDIUM):
    """(data, ResUNet state, coarse params, fine params, delta0, pixel picks) of the medium float64 gradient case; the same
    calls are made by tests/golden/make_golden_grad64.py in front of the reference."""
    from nerfool_amd.synthetic import make_scene
    from oracle.feature_net_ref import random_resunet_state
    from oracle.ibrnet_ref import random_ibrnet_params
    data = make_scene(c['H'], c['W'], c['V'], seed=c['seed'], tilt=0.3)
    cnn_sd = random_resunet_state(c['seed'] + 100)
    pc = random_ibrnet_params(c['S'], seed=30 + c['seed'])
    pf = random_ibrnet_params(c['S'] + c['N_imp'], seed=40 + c['seed'])
    gen = torch.Generator().manual_seed(c['seed'] + 5)
    eps = 8 / 255.
    delta0 = torch.zeros_like(data['src_rgbs']).uniform_(-eps, eps, generator=gen)
    delta0 = torch.max(torch.min(delta0, 1 - data['src_rgbs']), 0 - data['src_rgbs'])
    picks = np.random.RandomState(234).choice(c['H'] * c['W'], size=(c['R'],), replace=False)
    return data, cnn_sd, pc, pf, delta0, picks


def second_target_view(data, shift=(0.12, -0.05, 0.03), seed=77):
    """A second TARGET view for the universal-attack fixture: same source views (the perturbation lives on them), the target
    camera moved by `shift` (scene units, camera-to-world translation) and another smooth target image.  Deterministic; used
    by tests/golden/make_golden_r02.py in front of the reference and by the tests."""
    from nerfool_amd.synthetic import _smooth_image
    cam = data['camera'].clone()
    c2w = cam[0, 18:34].reshape(4, 4).clone()
    c2w[:3, 3] += torch.tensor(shift, dtype=cam.dtype)
    cam[0, 18:34] = c2w.reshape(-1)
    H, W = int(cam[0, 0]), int(cam[0, 1])
    out = dict(data)
    out['camera'] = cam
    out['rgb'] = _smooth_image(torch.Generator().manual_seed(seed), H, W)[None]
    out['rgb_path'] = ['golden_view_b']
    return out


# whole-attack outcome cases (tests/golden/attack100_<tag>.npz, made by tests/golden/make_golden_r05.py): every input is regenerated
# from seeds; the reference's settings are its defaults (config.py:119-169: adv_iters 100, epsilon 8, lr_step_size 100, lr_gamma 0.5)
# with README.md:64's --use_adam --adam_lr 1e-3.  c1 = BASELINE config 1's shape (V 4, 16 + 16 = 32 samples per ray, 100 iterations),
# c2 = config 2's sampling (64 + 64 samples, N_rand 512) on a half-size LLFF frame so that the reference's float64 run fits an hour.
ATTACK100 = {
    'c1': dict(H=96, W=128, V=4, S=16, N_imp=16, N_rand=512, adv_iters=100, epsilon=8, adam_lr=1e-3, lr_step_size=100,
               lr_gamma=0.5, seed=8, chunk_size=4096, delta_stride=1),
    # the two other loops of the reference at config 1's shape: sign-PGD (eval_adv.py:822-828, adv_lr 2 / 255: the perturbation saturates
    # at +-eps within a few steps) and the UNIVERSAL loop over two target views (eval_adv.py:609-740: adv_iters + 1 steps) = config 3's loop
    's1': dict(H=96, W=128, V=4, S=16, N_imp=16, N_rand=512, adv_iters=100, epsilon=8, adam_lr=1e-3, lr_step_size=100,
               lr_gamma=0.5, seed=8, chunk_size=4096, delta_stride=1, mode='sign', adv_lr=2),
    'u1': dict(H=96, W=128, V=4, S=16, N_imp=16, N_rand=512, adv_iters=100, epsilon=8, adam_lr=1e-3, lr_step_size=100,
               lr_gamma=0.5, seed=8, chunk_size=4096, delta_stride=1, mode='universal'),
    # the GNT flavour (config 4's network family at fixture size: depth 2, single network, 32 samples = the matrix-core kernels' smallest shape, unmasked MSE), view-specific loop
    'g1': dict(H=64, W=96, V=4, S=32, N_imp=0, N_rand=256, depth=2, adv_iters=100, epsilon=8, adam_lr=1e-3, lr_step_size=100,
               lr_gamma=0.5, seed=12, chunk_size=2048, delta_stride=1, mode='adam', flavour='gnt'),
    'c2': dict(H=378, W=504, V=4, S=64, N_imp=64, N_rand=512, adv_iters=100, epsilon=8, adam_lr=1e-3, lr_step_size=100,
               lr_gamma=0.5, seed=9, chunk_size=4096, delta_stride=10, image_dtype='float16'),
    # round 6: BASELINE config 2 at its REAL frame size (756 x 1008, 64 + 64 samples, N_rand 512), 100 of its 1000 iterations, run FIVE times by
    # the reference (tests/golden/make_golden_r06.py: float32, float64, and three other summation orders of float32 -- oneDNN off, 3 threads,
    # 5 threads with oneDNN off): the floor is a maximum over ten pairwise distances.  The attacked image is rendered through the
    # reference's own render_stride argument (every 4th pixel: 189 x 252 rays; a full-frame CPU render is ~40 min per run).
    'c2full': dict(H=756, W=1008, V=4, S=64, N_imp=64, N_rand=512, adv_iters=100, epsilon=8, adam_lr=1e-3, lr_step_size=100,
                   lr_gamma=0.5, seed=10, chunk_size=4096, delta_stride=40, image_dtype='float16', render_stride=4, skip_clean_render=True),
}


def attack100_inputs(c):
    """(data, ResUNet state, coarse params, fine params, delta0) of a whole-attack case; the same call is made by
    tests/golden/make_golden_r05.py in front of the reference."""
    from nerfool_amd.synthetic import make_scene
    from oracle.feature_net_ref import random_resunet_state
    from oracle.ibrnet_ref import random_ibrnet_params
    data = make_scene(c['H'], c['W'], c['V'], seed=c['seed'], tilt=0.3)
    cnn_sd = random_resunet_state(c['seed'] + 100)
    pc = random_ibrnet_params(c['S'], seed=30 + c['seed'])
    pf = random_ibrnet_params(c['S'] + c['N_imp'], seed=40 + c['seed'])
    gen = torch.Generator().manual_seed(c['seed'] + 5)
    eps = c['epsilon'] / 255.
    delta0 = torch.zeros_like(data['src_rgbs']).uniform_(-eps, eps, generator=gen)
    delta0 = torch.max(torch.min(delta0, 1 - data['src_rgbs']), 0 - data['src_rgbs'])
    return data, cnn_sd, pc, pf, delta0


def attack_outcome_stats(a, b, eps):
    """Distances between the outcomes of two runs of one attack (dicts with losses [T], delta (flat or strided sample), image
    [H,W,3], psnr): the quantities the whole-attack parity test bounds by multiples of the reference's own fp32-vs-float64 values."""
    la, lb = np.asarray(a['losses'], dtype=np.float64), np.asarray(b['losses'], dtype=np.float64)
    da, db = np.asarray(a['delta'], dtype=np.float64).reshape(-1), np.asarray(b['delta'], dtype=np.float64).reshape(-1)
    ia, ib = np.asarray(a['image'], dtype=np.float64), np.asarray(b['image'], dtype=np.float64)
    at = lambda d: float((np.abs(d) >= eps * (1 - 1e-5)).mean())
    return dict(loss_rel_max=float(np.max(np.abs(la - lb) / lb)),
                loss_rel_mean=float(np.mean(np.abs(la - lb) / lb)),
                loss_last10_rel=float(abs(la[-10:].mean() - lb[-10:].mean()) / lb[-10:].mean()),
                delta_mean_abs_over_eps=float(np.mean(np.abs(da - db)) / eps),
                delta_sign_disagree=float(np.mean(np.sign(da) != np.sign(db))),
                frac_at_eps_diff=abs(at(da) - at(db)),
                image_rms=float(np.sqrt(np.mean((ia - ib) ** 2))),
                psnr_diff=abs(float(a['psnr']) - float(b['psnr'])))


def attack100_gnt_inputs(c):
    """(data, single-network ResUNet state, GNT parameters, delta0) of the GNT whole-attack case; the same call is made by
    tests/golden/make_golden_r05_gnt.py in front of the reference."""
    from nerfool_amd.synthetic import make_scene
    from oracle.feature_net_ref import random_resunet_state
    from oracle.gnt_ref import random_gnt_params
    data = make_scene(c['H'], c['W'], c['V'], seed=c['seed'], tilt=0.3)
    cnn_sd = random_resunet_state(c['seed'] + 100, 32, 0)
    params = random_gnt_params(c['depth'], seed=60 + c['seed'])
    gen = torch.Generator().manual_seed(c['seed'] + 5)
    eps = c['epsilon'] / 255.
    delta0 = torch.zeros_like(data['src_rgbs']).uniform_(-eps, eps, generator=gen)
    delta0 = torch.max(torch.min(delta0, 1 - data['src_rgbs']), 0 - data['src_rgbs'])
    return data, cnn_sd, params, delta0


# training-mode GNT on the MATRIX-CORE kernels (tests/golden/gnt_train_mfma_d2.npz, make_golden_gnt_train.py mfma): a network-level input at
# the smallest shape those kernels take (32 samples per ray), regenerated from seeds on both sides
GNT_TRAIN_MFMA = dict(R=6, S=32, V=3, depth=2, seed=31)


def gnt_train_mfma_inputs(c=GNT_TRAIN_MFMA):
    """(parameters, rgb_feat [R,S,V,35], ray_diff [R,S,V,4], mask [R,S,V,1], pts [R,S,3], ray_d [R,3]) -- the same call is made by
    tests/golden/make_golden_gnt_train.py in front of the reference"""
    from oracle.gnt_ref import random_gnt_params
    R, S, V = c['R'], c['S'], c['V']
    g = torch.Generator().manual_seed(c['seed'])
    params = random_gnt_params(c['depth'], seed=70 + c['seed'])
    rgb_feat = torch.randn(R, S, V, 35, generator=g) * 0.6
    rgb_feat[..., :3] = torch.rand(R, S, V, 3, generator=g)
    d = torch.randn(R, S, V, 3, generator=g)
    ray_diff = torch.cat([d / d.norm(dim=-1, keepdim=True), torch.rand(R, S, V, 1, generator=g) * 2 - 1], dim=-1)
    mask = (torch.rand(R, S, V, 1, generator=g) > 0.15).float()
    mask[0, :4] = 0.0                       # samples no view sees
    mask[1, :, 0] = 0.0                     # a view that sees nothing of a ray
    ray_d = torch.randn(R, 3, generator=g)
    ray_o = torch.randn(R, 3, generator=g) * 0.2
    z = torch.linspace(2.0, 6.0, S)
    pts = ray_o[:, None, :] + z[None, :, None] * (ray_d / ray_d.norm(dim=-1, keepdim=True))[:, None, :]
    return params, rgb_feat, ray_diff, mask, pts, ray_d
