"""Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): GNT per-ray network and renderer, PyTorch-CPU fp32.  Eval mode by default
(all Dropout layers are identities: the view-specific attack runs after `model.switch_to_eval()`); with `dropout=(seed, p)`
the eight Dropout sites of every layer are live -- the reference's UNIVERSAL GNT loop runs before `switch_to_eval`
(eval/gnt/eval_adv.py:739-878 vs :959) -- with masks from a counter-based generator (`keep_mask`) instead of torch's: the same function
is compiled into the kernels and injected into the reference's own modules by tests/golden/make_golden_gnt_train.py, so all three
evaluate identical masks.

ref: gnt/transformer_network.py:6-37 (Embedder), :55-89 (Attention2D), :93-113 (Transformer2D), :121-171 (Attention),
:175-202 (Transformer), :205-309 (GNT); gnt/render_ray.py:196-279 (render_rays, N_importance = 0, ret_alpha = False);
gnt/criterion.py:14-20 + eval/gnt/utils.py (unmasked MSE).  State-dict keys are the reference module paths.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import ibrnet_ref as ib


def posenc(x, n_freqs=10):
    """Embedder(include_input, log_sampling, max_freq_log2 = 9, num_freqs = 10): [x, sin(2^k x), cos(2^k x)]_k  (3 -> 63)."""
    freqs = 2.0 ** torch.linspace(0.0, float(n_freqs - 1), steps=n_freqs)
    out = [x]
    for f in freqs:
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


# ---- counter-based dropout masks (round 5) ----------------------------------------------------------------------------------------------
# keep(seed, site, idx): bit-exact restatement of nf_gnt.h:gnt_keep (32-bit integer hash, murmur3 finaliser rounds).  site = 8 * layer +
# {0 view-attention probabilities [R,S,V,64], 1 view-attention output [R,S,64], 2 view feed-forward hidden [R,S,256], 3 view feed-forward
# output [R,S,64], 4 ray-attention probabilities [R,4,S,S], 5 ray-attention output, 6 ray feed-forward hidden, 7 ray feed-forward output};
# idx = flat index of the element in the tensor as the reference hands it to nn.Dropout (C order of the shapes above).
def _mix32(x):
    import numpy as np
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85ebca6b)) & np.uint64(0xffffffff)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xc2b2ae35)) & np.uint64(0xffffffff)
    x ^= x >> np.uint64(16)
    return x


def keep_mask(seed, site, shape, p):
    """float32 tensor of `shape`: 1 / (1 - p) where the element is kept, 0 where it is dropped"""
    import numpy as np
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    m32 = np.uint64(0xffffffff)
    a = _mix32(np.array([(int(seed) ^ ((int(site) * 0x9e3779b9) & 0xffffffff)) & 0xffffffff], dtype=np.uint64))
    a = _mix32(a ^ (idx & m32))
    a = _mix32((a + (((idx >> np.uint64(32)) * np.uint64(0x7f4a7c15)) & m32) + np.uint64(0x165667b1)) & m32)
    thr = np.uint64(int(float(p) * 16777216.0))
    keep = (a >> np.uint64(8)) >= thr
    return torch.from_numpy((keep.astype(np.float32) / np.float32(1.0 - float(p))).reshape(shape))


def _drop(x, dropout, site):
    if dropout is None:
        return x
    seed, p = dropout
    return x * keep_mask(seed, site, tuple(x.shape), p).to(x.dtype)


def _lin(p, name, x, bias=True):
    return F.linear(x, p[name + '.weight'], p[name + '.bias'] if bias else None)


def _ln(p, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), p[name + '.weight'], p[name + '.bias'], eps=eps)


def view_transformer(p, pre, q, X, ray_diff, mask, dropout=None, site0=0):
    """Transformer2D: q [R,S,C], X [R,S,V,C], ray_diff [R,S,V,4], mask [R,S,V,1].  dropout = (seed, p): sites site0 .. site0 + 3."""
    a = pre + '.attn'
    x = _ln(p, pre + '.attn_norm', q, 1e-6)
    Q = _lin(p, a + '.q_fc', x, bias=False)
    K = _lin(p, a + '.k_fc', X, bias=False)
    Vv = _lin(p, a + '.v_fc', K, bias=False)                     # v = v_fc(k_fc(k))  (:76-77)
    pos = _lin(p, a + '.pos_fc.2', F.relu(_lin(p, a + '.pos_fc.0', ray_diff)))
    att = K - Q[:, :, None, :] + pos
    att = _lin(p, a + '.attn_fc.2', F.relu(_lin(p, a + '.attn_fc.0', att)))
    att = att.masked_fill(mask == 0, -1e9)
    att = torch.softmax(att, dim=-2)                             # over views, per channel
    att = _drop(att, dropout, site0 + 0)                         # :85
    x = _drop(_lin(p, a + '.out_fc', ((Vv + pos) * att).sum(dim=2)), dropout, site0 + 1)       # :88
    x = x + q
    y = _ln(p, pre + '.ff_norm', x, 1e-6)
    y = _drop(F.relu(_lin(p, pre + '.ff.fc1', y)), dropout, site0 + 2)                         # :47
    y = _drop(_lin(p, pre + '.ff.fc2', y), dropout, site0 + 3)                                 # :48
    return y + x


def ray_transformer(p, pre, q, n_heads=4, ret_attn=False, dropout=None, site0=4):
    """Transformer (attn_mode 'qk'): pre-LN multi-head self-attention over the S samples of a ray, no mask.
    ret_attn: also the attention row of sample 0 averaged over the heads (transformer_network.py:196-200)."""
    a = pre + '.attn'
    R, S, C = q.shape
    x = _ln(p, pre + '.attn_norm', q, 1e-6)
    split = lambda t: t.view(R, S, n_heads, C // n_heads).permute(0, 2, 1, 3)
    Q, K, Vv = split(_lin(p, a + '.q_fc', x, False)), split(_lin(p, a + '.k_fc', x, False)), split(_lin(p, a + '.v_fc', x, False))
    att = torch.softmax(torch.matmul(Q, K.transpose(-2, -1)) / (C // n_heads) ** 0.5, dim=-1)
    att = _drop(att, dropout, site0 + 0)                         # :162 (the attention returned with ret_attn is the dropped one)
    out = torch.matmul(att, Vv).permute(0, 2, 1, 3).reshape(R, S, C)
    x = _drop(_lin(p, a + '.out_fc', out), dropout, site0 + 1) + q                             # :166
    y = _ln(p, pre + '.ff_norm', x, 1e-6)
    y = _drop(F.relu(_lin(p, pre + '.ff.fc1', y)), dropout, site0 + 2)
    y = _drop(_lin(p, pre + '.ff.fc2', y), dropout, site0 + 3)
    if ret_attn:
        return y + x, att.mean(dim=1)[:, 0]
    return y + x


def gnt_forward(p, rgb_feat, ray_diff, mask, pts, ray_d, trans_depth, ret_alpha=False, dropout=None):
    """ref: gnt/transformer_network.py:270-309 -> rgb [R,3], or [R,3+S] with ret_alpha (attention of the last ray transformer).
    dropout = (seed, p): training-mode forward with the counter-based masks (module header)."""
    viewdirs = ray_d / torch.norm(ray_d, dim=-1, keepdim=True)
    view_emb = posenc(viewdirs.reshape(-1, 3).float())                        # [R,63]
    pts_emb = posenc(pts.reshape(-1, 3).float()).reshape(list(pts.shape[:-1]) + [63])
    view_emb = view_emb[:, None].expand(pts_emb.shape)
    X = _lin(p, 'rgbfeat_fc.2', F.relu(_lin(p, 'rgbfeat_fc.0', rgb_feat)))
    q = X.max(dim=2)[0]
    for i in range(trans_depth):
        q = view_transformer(p, 'view_crosstrans.%d' % i, q, X, ray_diff, mask, dropout, 8 * i)
        if i % 2 == 0:
            q = torch.cat((q, pts_emb, view_emb), dim=-1)
            q = _lin(p, 'q_fcs.%d.2' % i, F.relu(_lin(p, 'q_fcs.%d.0' % i, q)))
        q = ray_transformer(p, 'view_selftrans.%d' % i, q, ret_attn=ret_alpha, dropout=dropout, site0=8 * i + 4)
        if ret_alpha:
            q, attn = q
    h = _ln(p, 'norm', q, 1e-5)
    out = _lin(p, 'rgb_fc', h.mean(dim=1))
    return torch.cat([out, attn], dim=1) if ret_alpha else out


def random_gnt_params(trans_depth, seed, width=64):
    """Fixture weights: scaled normal matrices, small biases, LayerNorm gains ~ 1 -- keys in reference module order."""
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()

    def lin(name, fin, fout, bias=True, gain=1.0):
        p[name + '.weight'] = torch.randn(fout, fin, generator=g) * gain * (1.0 / fin) ** 0.5
        if bias:
            p[name + '.bias'] = torch.randn(fout, generator=g) * 0.05

    def ln(name):
        p[name + '.weight'] = 1.0 + 0.1 * torch.randn(width, generator=g)
        p[name + '.bias'] = 0.05 * torch.randn(width, generator=g)

    lin('rgbfeat_fc.0', 35, width)
    lin('rgbfeat_fc.2', width, width)
    for i in range(trans_depth):
        pre = 'view_selftrans.%d' % i
        ln(pre + '.attn_norm'); ln(pre + '.ff_norm')
        lin(pre + '.ff.fc1', width, 4 * width); lin(pre + '.ff.fc2', 4 * width, width)
        for n in ('q_fc', 'k_fc', 'v_fc'):
            lin(pre + '.attn.' + n, width, width, bias=False, gain=1.5)
        lin(pre + '.attn.out_fc', width, width)
    for i in range(trans_depth):
        pre = 'view_crosstrans.%d' % i
        ln(pre + '.attn_norm'); ln(pre + '.ff_norm')
        lin(pre + '.ff.fc1', width, 4 * width); lin(pre + '.ff.fc2', 4 * width, width)
        for n in ('q_fc', 'k_fc', 'v_fc'):
            lin(pre + '.attn.' + n, width, width, bias=False)
        lin(pre + '.attn.pos_fc.0', 4, width // 8); lin(pre + '.attn.pos_fc.2', width // 8, width)
        lin(pre + '.attn.attn_fc.0', width, width // 8, gain=1.5); lin(pre + '.attn.attn_fc.2', width // 8, width, gain=1.5)
        lin(pre + '.attn.out_fc', width, width)
    for i in range(0, trans_depth, 2):
        lin('q_fcs.%d.0' % i, width + 126, width); lin('q_fcs.%d.2' % i, width, width)
    ln('norm')
    lin('rgb_fc', width, 3)
    return p


def render_rays(ray_batch, params, featmaps, N_samples, trans_depth, inv_uniform=False, det=False, src_ray_batch=None,
                N_importance=0, ret_alpha=False, dropout=None):
    """ref: gnt/render_ray.py:196-279 (single_net): {'rgb', 'weights', 'depth'} per level; the fine pass resamples on the
    detached attention weights of the coarse pass (sample_fine_pts :164-193 == the IBRNet fine-sample assembly).
    dropout = (seed of the first network call, p): training mode; the fine pass is the NEXT call and takes seed + 1."""
    calls = [0]
    src = ray_batch if src_ray_batch is None else src_ray_batch
    pts, z_vals = ib.sample_along_camera_ray(ray_batch['ray_o'], ray_batch['ray_d'], ray_batch['depth_range'], N_samples,
                                             inv_uniform=inv_uniform, det=det)

    def level(pts, z_vals, fm, with_alpha):
        rgb_feat, ray_diff, mask = ib.projector_compute(pts, ray_batch['camera'], src['src_rgbs'], src['src_cameras'], fm)
        dp = None if dropout is None else ((dropout[0] + calls[0]) & 0xffffffff, dropout[1])
        calls[0] += 1
        out = gnt_forward(params, rgb_feat, ray_diff, mask, pts, ray_batch['ray_d'], trans_depth, ret_alpha=with_alpha, dropout=dp)
        if not with_alpha:
            return {'rgb': out, 'weights': None, 'depth': None}
        return {'rgb': out[:, :3], 'weights': out[:, 3:], 'depth': torch.sum(out[:, 3:] * z_vals, dim=-1)}

    ret = {'outputs_coarse': level(pts, z_vals, featmaps[0], ret_alpha), 'outputs_fine': None}
    if N_importance > 0:
        z_vals = ib.fine_depths(z_vals, ret['outputs_coarse']['weights'].clone().detach(), N_importance, inv_uniform, det)
        pts = z_vals[..., None] * ray_batch['ray_d'][:, None, :] + ray_batch['ray_o'][:, None, :]
        ret['outputs_fine'] = level(pts, z_vals, featmaps[1], True)
    return ret


def criterion(outputs, ray_batch):
    """gnt/criterion.py:14-20: no 'mask' key in the GNT outputs => plain mean squared error."""
    return ib.img2mse(outputs['rgb'], ray_batch['rgb'], None)
