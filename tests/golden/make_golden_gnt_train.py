#!/usr/bin/env python
"""Golden vectors of the GNT network in TRAINING mode (reference gnt/ package imported from /root/reference).  The reference's universal
GNT loop runs before `model.switch_to_eval()` (eval/gnt/eval_adv.py:739-878 vs :959): all 8 nn.Dropout(0.1) sites of every layer are live
(gnt/transformer_network.py:45-48, :72/:85-88, :136/:162-166).  torch's CPU generator cannot be reproduced on a GPU, so parity is pinned twice:

  exact/*   the reference network in train() mode with every nn.Dropout instance REPLACED by a module that multiplies by the mask of the
            counter-based generator `oracle.gnt_ref.keep_mask(seed, site, shape, p)` -- the function the kernels compile in (nf_gnt.h:
            gnt_keep).  Inputs = the network-level capture of gnt_tiny_d2_v4.npz / gnt_alpha_d2_v3.npz; stored: rgb (and the returned
            attention with ret_alpha) and d sum(w * out) / d rgb_feat for two seeds.
  stat/*    the reference network in train() mode with ITS OWN nn.Dropout on torch's generator: mean and standard deviation of the output
            over 400 draws -- what the distribution of the counter-based masks must reproduce.

    python tests/golden/make_golden_gnt_train.py        # writes tests/golden/gnt_train_d2.npz
    python tests/golden/make_golden_gnt_train.py mfma   # writes tests/golden/gnt_train_mfma_d2.npz (32 samples per ray: the matrix-core kernels)
Data only; build container only."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _refimport  # noqa: E402

_refimport.install('gnt')

from gnt.transformer_network import GNT  # noqa: E402

from oracle.gnt_ref import keep_mask  # noqa: E402

P_DROP = 0.1


class MaskDropout(nn.Module):
    """stands in for one nn.Dropout instance of the reference; every module of the reference calls its `dp` twice per forward (attention:
    probabilities, then output; feed-forward: hidden, then output), hence the call counter"""

    def __init__(self, state, site0):
        super().__init__()
        self.state, self.site0, self.calls = state, site0, 0

    def forward(self, x):
        site = self.site0 + (self.calls & 1)
        self.calls += 1
        if not self.training:
            return x
        return x * keep_mask(self.state['seed'], site, tuple(x.shape), P_DROP).to(x.dtype)


def inject(net, state):
    for i, (ct, st) in enumerate(zip(net.view_crosstrans, net.view_selftrans)):
        ct.attn.dp = MaskDropout(state, 8 * i + 0)
        ct.ff.dp = MaskDropout(state, 8 * i + 2)
        st.attn.dp = MaskDropout(state, 8 * i + 4)
        st.ff.dp = MaskDropout(state, 8 * i + 6)


def npy(t):
    return t.detach().cpu().numpy().copy()


def run(tag, base, ret_alpha, out):
    if base == 'seeded':  # round 6: the matrix-core kernels' smallest shape (32 samples per ray), every input regenerated from seeds
        sys.path.insert(0, os.path.dirname(HERE))
        from fixtures import GNT_TRAIN_MFMA, gnt_train_mfma_inputs
        params, rgb_feat, ray_diff, mask, pts, ray_d = gnt_train_mfma_inputs()
        depth = GNT_TRAIN_MFMA['depth']
    else:
        z = np.load(os.path.join(HERE, base + '.npz'))
        depth = int(z['cfg'][5])
        t = lambda k: torch.from_numpy(z[k])
        params = {k[4:]: t(k) for k in z.files if k.startswith('net/')}
        if ret_alpha:         # the alpha fixture has no network-level capture: take the geometry of the tiny one, its own weights
            zin = np.load(os.path.join(HERE, 'gnt_tiny_d2_v4.npz'))
        else:
            zin = z
        tin = lambda k: torch.from_numpy(zin[k])
        rgb_feat, ray_diff, mask, pts = tin('net_in/rgb_feat'), tin('net_in/ray_diff'), tin('net_in/mask'), tin('net_in/pts')
        ray_d = tin('in/ray_d')
    R, S = rgb_feat.shape[:2]
    net = GNT(SimpleNamespace(netwidth=64, trans_depth=depth), in_feat_ch=32, posenc_dim=63, viewenc_dim=63, ret_alpha=ret_alpha)
    net.load_state_dict(params, strict=True)
    out[tag + '/base'] = np.array(base)
    out[tag + '/geometry'] = np.array('seeded' if base == 'seeded' else ('gnt_tiny_d2_v4' if ret_alpha else base))
    gen = torch.Generator().manual_seed(17)
    w = torch.randn(R, 3 + (S if ret_alpha else 0), generator=gen)
    out[tag + '/w'] = npy(w)
    # ---- statistics with the reference's own Dropout on torch's generator
    net.train()
    torch.manual_seed(1234)
    with torch.no_grad():
        draws = torch.stack([net(rgb_feat, ray_diff, mask, pts, ray_d) for _ in range(400)])
    out[tag + '/stat/mean'] = npy(draws.mean(0))
    out[tag + '/stat/std'] = npy(draws.std(0))
    out[tag + '/stat/n'] = np.array(400)
    net.eval()
    with torch.no_grad():
        out[tag + '/eval'] = npy(net(rgb_feat, ray_diff, mask, pts, ray_d))
    # ---- exact: injected counter-based masks
    state = {'seed': 0}
    inject(net, state)
    net.train()
    for seed in (5, 90210):
        state['seed'] = seed
        for m in net.modules():
            if isinstance(m, MaskDropout):
                m.calls = 0
        x = rgb_feat.clone().requires_grad_(True)
        y = net(x, ray_diff, mask, pts, ray_d)
        g, = torch.autograd.grad((y * w).sum(), x)
        out['%s/exact/%d/out' % (tag, seed)] = npy(y)
        out['%s/exact/%d/d_rgb_feat' % (tag, seed)] = npy(g)
        print('%s seed %d: out range [%.3f, %.3f], |out - eval| max %.3e, stat std mean %.3e' % (
            tag, seed, float(y.min()), float(y.max()), float((y - torch.from_numpy(out[tag + '/eval'])).abs().max()),
            float(out[tag + '/stat/std'].mean())))
    out[tag + '/seeds'] = np.array([5, 90210], dtype=np.int64)


if __name__ == '__main__':
    torch.set_num_threads(4)
    out = {'p': np.array(P_DROP)}
    if sys.argv[1:] == ['mfma']:          # python tests/golden/make_golden_gnt_train.py mfma -> gnt_train_mfma_d2.npz
        run('plain', 'seeded', False, out)
        run('alpha', 'seeded', True, out)
        path = os.path.join(HERE, 'gnt_train_mfma_d2.npz')
    else:
        run('plain', 'gnt_tiny_d2_v4', False, out)
        run('alpha', 'gnt_alpha_d2_v3', True, out)
        path = os.path.join(HERE, 'gnt_train_d2.npz')
    np.savez_compressed(path, **out)
    print('%s %.1f KB' % (path, os.path.getsize(path) / 1024.))
