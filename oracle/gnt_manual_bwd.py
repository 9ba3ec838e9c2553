"""Oracle (TEST INFRASTRUCTURE): hand-derived backward of GNT.forward w.r.t. `rgb_feat` (no autograd) -- blueprint of the
HIP GNT backward kernels; checked against autograd of oracle.gnt_ref in tests/test_manual_backward.py.
ref of the forward: gnt/transformer_network.py:270-309 (eval mode, ret_alpha = False)."""
import torch
import torch.nn.functional as F

from .gnt_ref import posenc


def _ln_fwd(x, w, b, eps):
    mu = x.mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((x - mu) ** 2).mean(-1, keepdim=True) + eps)
    xh = (x - mu) * rstd
    return xh * w + b, (xh, rstd)


def _ln_bwd(dy, w, saved):
    xh, rstd = saved
    dxh = dy * w
    return rstd * (dxh - dxh.mean(-1, keepdim=True) - xh * (dxh * xh).mean(-1, keepdim=True))


def forward_saved(p, rgb_feat, ray_diff, mask, pts, ray_d, depth):
    sv = {'layers': []}
    W = lambda n: p[n + '.weight']
    Bv = lambda n: p[n + '.bias']
    R, S, V, _ = rgb_feat.shape
    r = F.relu(F.linear(rgb_feat, W('rgbfeat_fc.0'), Bv('rgbfeat_fc.0')))
    X = F.linear(r, W('rgbfeat_fc.2'), Bv('rgbfeat_fc.2'))
    q, amax = X.max(dim=2)
    sv.update(r=r, X=X, amax=amax)
    viewdirs = ray_d / torch.norm(ray_d, dim=-1, keepdim=True)
    pe = torch.cat([posenc(pts.reshape(-1, 3)).reshape(R, S, 63), posenc(viewdirs)[:, None].expand(R, S, 63)], -1)
    for i in range(depth):
        L = {}
        a = 'view_crosstrans.%d' % i
        x, L['ln1'] = _ln_fwd(q, W(a + '.attn_norm'), Bv(a + '.attn_norm'), 1e-6)
        Q = F.linear(x, W(a + '.attn.q_fc'))
        K = F.linear(X, W(a + '.attn.k_fc'))
        Vv = F.linear(K, W(a + '.attn.v_fc'))
        pos = F.linear(F.relu(F.linear(ray_diff, W(a + '.attn.pos_fc.0'), Bv(a + '.attn.pos_fc.0'))), W(a + '.attn.pos_fc.2'),
                       Bv(a + '.attn.pos_fc.2'))
        h = F.relu(F.linear(K - Q[:, :, None] + pos, W(a + '.attn.attn_fc.0'), Bv(a + '.attn.attn_fc.0')))
        logit = F.linear(h, W(a + '.attn.attn_fc.2'), Bv(a + '.attn.attn_fc.2')).masked_fill(mask == 0, -1e9)
        prob = torch.softmax(logit, dim=2)
        u = ((Vv + pos) * prob).sum(2)
        q1 = q + F.linear(u, W(a + '.attn.out_fc'), Bv(a + '.attn.out_fc'))
        y, L['ln2'] = _ln_fwd(q1, W(a + '.ff_norm'), Bv(a + '.ff_norm'), 1e-6)
        f = F.relu(F.linear(y, W(a + '.ff.fc1'), Bv(a + '.ff.fc1')))
        q = q1 + F.linear(f, W(a + '.ff.fc2'), Bv(a + '.ff.fc2'))
        L.update(vp=Vv + pos, h=h, prob=prob, f=f)
        if i % 2 == 0:
            g = F.relu(F.linear(torch.cat([q, pe], -1), W('q_fcs.%d.0' % i), Bv('q_fcs.%d.0' % i)))
            q = F.linear(g, W('q_fcs.%d.2' % i), Bv('q_fcs.%d.2' % i))
            L['g'] = g
        b = 'view_selftrans.%d' % i
        x, L['rln1'] = _ln_fwd(q, W(b + '.attn_norm'), Bv(b + '.attn_norm'), 1e-6)
        split = lambda t: t.view(R, S, 4, 16).permute(0, 2, 1, 3)
        Qh, Kh, Vh = split(F.linear(x, W(b + '.attn.q_fc'))), split(F.linear(x, W(b + '.attn.k_fc'))), split(F.linear(x, W(b + '.attn.v_fc')))
        A = torch.softmax(Qh @ Kh.transpose(-2, -1) / 4.0, -1)
        out = (A @ Vh).permute(0, 2, 1, 3).reshape(R, S, 64)
        q1 = q + F.linear(out, W(b + '.attn.out_fc'), Bv(b + '.attn.out_fc'))
        y, L['rln2'] = _ln_fwd(q1, W(b + '.ff_norm'), Bv(b + '.ff_norm'), 1e-6)
        f2 = F.relu(F.linear(y, W(b + '.ff.fc1'), Bv(b + '.ff.fc1')))
        q = q1 + F.linear(f2, W(b + '.ff.fc2'), Bv(b + '.ff.fc2'))
        L.update(Qh=Qh, Kh=Kh, Vh=Vh, A=A, f2=f2)
        sv['layers'].append(L)
    hfin, sv['lnf'] = _ln_fwd(q, W('norm'), Bv('norm'), 1e-5)
    rgb = F.linear(hfin.mean(1), W('rgb_fc'), Bv('rgb_fc'))
    return rgb, sv


def backward_rgb_feat(p, sv, mask, d_rgb, depth):
    W = lambda n: p[n + '.weight']
    X = sv['X']
    R, S, V, C = X.shape
    d_X = torch.zeros_like(X)
    d_q = _ln_bwd(((d_rgb @ W('rgb_fc')) / S)[:, None, :].expand(R, S, C), W('norm'), sv['lnf'])
    for i in reversed(range(depth)):
        L = sv['layers'][i]
        b = 'view_selftrans.%d' % i
        # ---- ray transformer
        d_q1 = d_q + _ln_bwd(((d_q @ W(b + '.ff.fc2')) * (L['f2'] > 0).float()) @ W(b + '.ff.fc1'), W(b + '.ff_norm'), L['rln2'])
        d_out = (d_q1 @ W(b + '.attn.out_fc')).view(R, S, 4, 16).permute(0, 2, 1, 3)
        A, Qh, Kh, Vh = L['A'], L['Qh'], L['Kh'], L['Vh']
        d_V = A.transpose(-2, -1) @ d_out
        d_A = d_out @ Vh.transpose(-2, -1)
        d_S = A * (d_A - (A * d_A).sum(-1, keepdim=True))
        d_Q = d_S @ Kh / 4.0
        d_K = d_S.transpose(-2, -1) @ Qh / 4.0
        flat = lambda t: t.permute(0, 2, 1, 3).reshape(R, S, C)
        d_x = flat(d_Q) @ W(b + '.attn.q_fc') + flat(d_K) @ W(b + '.attn.k_fc') + flat(d_V) @ W(b + '.attn.v_fc')
        d_q = d_q1 + _ln_bwd(d_x, W(b + '.attn_norm'), L['rln1'])
        # ---- positional MLP on even layers
        if i % 2 == 0:
            d_g = (d_q @ W('q_fcs.%d.2' % i)) * (L['g'] > 0).float()
            d_q = (d_g @ W('q_fcs.%d.0' % i))[..., :C]
        # ---- view transformer
        a = 'view_crosstrans.%d' % i
        d_q1 = d_q + _ln_bwd(((d_q @ W(a + '.ff.fc2')) * (L['f'] > 0).float()) @ W(a + '.ff.fc1'), W(a + '.ff_norm'), L['ln2'])
        d_u = d_q1 @ W(a + '.attn.out_fc')
        prob = L['prob']
        d_prob = L['vp'] * d_u[:, :, None]
        d_Vv = prob * d_u[:, :, None]
        d_logit = prob * (d_prob - (prob * d_prob).sum(2, keepdim=True)) * (mask != 0).float()
        d_a = ((d_logit @ W(a + '.attn.attn_fc.2')) * (L['h'] > 0).float()) @ W(a + '.attn.attn_fc.0')
        d_K = d_a + d_Vv @ W(a + '.attn.v_fc')
        d_X = d_X + d_K @ W(a + '.attn.k_fc')
        d_x = (-d_a.sum(2)) @ W(a + '.attn.q_fc')
        d_q = d_q1 + _ln_bwd(d_x, W(a + '.attn_norm'), L['ln1'])
    d_X.scatter_add_(2, sv['amax'][:, :, None, :], d_q[:, :, None, :])           # q0 = max over views
    d_r = (d_X @ W('rgbfeat_fc.2')) * (sv['r'] > 0).float()
    return d_r @ W('rgbfeat_fc.0')
