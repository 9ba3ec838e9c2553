"""Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): hand-derived backward of IBRNet.forward w.r.t.
`rgb_feat`, written without autograd.  It is the blueprint the HIP backward kernels follow (same saved
quantities, same order of operations) and is itself checked against autograd of oracle.ibrnet_ref in
tests/test_manual_backward.py.

ref for the forward being differentiated: ibrnet/mlp_network.py:222-274.
Only d/d(rgb_feat) is produced: ray_diff, mask, the first pooling weight and all network weights are constants of
the attack.
"""
import torch
import torch.nn.functional as F


def _elu_grad_from_out(y):
    """ELU'(pre) expressed through the OUTPUT y = ELU(pre): 1 if y > 0 else y + 1."""
    return torch.where(y > 0, torch.ones_like(y), y + 1.0)


def forward_saved(p, rgb_feat, ray_diff, mask, anti_alias_pooling=True):
    """Forward pass that keeps every activation the backward needs (names match the HIP kernels)."""
    sv = {}
    W = lambda n: p[n + '.weight']
    B = lambda n: p[n + '.bias']
    lin = lambda n, x: F.linear(x, W(n), B(n))
    V = rgb_feat.shape[2]
    d1 = F.elu(lin('ray_dir_fc.0', ray_diff))
    dirf = F.elu(lin('ray_dir_fc.2', d1))
    f = rgb_feat + dirf
    if anti_alias_pooling:
        e = torch.exp(torch.abs(p['s']) * (ray_diff[..., 3:4] - 1))
        w = (e - e.min(dim=2, keepdim=True)[0]) * mask
        w = w / (w.sum(dim=2, keepdim=True) + 1e-8)
    else:
        w = mask / (mask.sum(dim=2, keepdim=True) + 1e-8)
    mean = (f * w).sum(2, keepdim=True)
    var = (w * (f - mean) ** 2).sum(2, keepdim=True)
    x_in = torch.cat([mean.expand(-1, -1, V, -1), var.expand(-1, -1, V, -1), f], -1)
    h1 = F.elu(lin('base_fc.0', x_in))
    h = F.elu(lin('base_fc.2', h1))
    v1 = F.elu(lin('vis_fc.0', h * w))
    xv = F.elu(lin('vis_fc.2', v1))
    sig1 = torch.sigmoid(xv[..., 32:33])
    vis1 = sig1 * mask
    x2 = h + xv[..., :32]
    u = F.elu(lin('vis_fc2.0', x2 * vis1))
    sig2 = torch.sigmoid(lin('vis_fc2.2', u))
    vis2 = sig2 * mask
    vsum = vis2.sum(2, keepdim=True) + 1e-8
    w2 = vis2 / vsum
    mean2 = (x2 * w2).sum(2, keepdim=True)
    var2 = (w2 * (x2 - mean2) ** 2).sum(2, keepdim=True)
    g_in = torch.cat([mean2.squeeze(2), var2.squeeze(2), w2.mean(2)], -1)
    g1 = F.elu(lin('geometry_fc.0', g_in))
    g = F.elu(lin('geometry_fc.2', g1))
    n_valid = mask.sum(2)
    gpe = g + p['pos_encoding']
    R, S, _ = gpe.shape
    q = F.linear(gpe, p['ray_attention.w_qs.weight']).view(R, S, 4, 4).transpose(1, 2)
    k = F.linear(gpe, p['ray_attention.w_ks.weight']).view(R, S, 4, 4).transpose(1, 2)
    v = F.linear(gpe, p['ray_attention.w_vs.weight']).view(R, S, 4, 4).transpose(1, 2)
    row_on = (n_valid > 1).float()                                    # [R,S,1]
    scores = torch.matmul(q / 2.0, k.transpose(2, 3)).masked_fill(row_on[:, None] == 0, -1e9)
    attn = F.softmax(scores, -1)                                      # [R,4,S,S]
    o = torch.matmul(attn, v).transpose(1, 2).reshape(R, S, 16)
    pre = F.linear(o, p['ray_attention.fc.weight']) + gpe
    mu = pre.mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((pre - mu) ** 2).mean(-1, keepdim=True) + 1e-6)
    xhat = (pre - mu) * rstd
    gat = xhat * p['ray_attention.layer_norm.weight'] + p['ray_attention.layer_norm.bias']
    og1 = F.elu(lin('out_geometry_fc.0', gat))
    sig_pre = lin('out_geometry_fc.2', og1)
    sigma = F.relu(sig_pre).masked_fill(n_valid < 1, 0.0)
    y_in = torch.cat([x2, vis2, ray_diff], -1)
    r1 = F.elu(lin('rgb_fc.0', y_in))
    r2 = F.elu(lin('rgb_fc.2', r1))
    y = lin('rgb_fc.4', r2).masked_fill(mask == 0, -1e9)
    beta = F.softmax(y, 2)
    rgb_in = rgb_feat[..., :3]
    rgb = (rgb_in * beta).sum(2)
    sv.update(dict(f=f, w=w, mean=mean, var=var, h1=h1, h=h, v1=v1, xv=xv, sig1=sig1, vis1=vis1, x2=x2, u=u,
                   sig2=sig2, vis2=vis2, vsum=vsum, w2=w2, mean2=mean2, var2=var2, g1=g1, g=g, n_valid=n_valid,
                   q=q, k=k, v=v, row_on=row_on, attn=attn, o=o, rstd=rstd, xhat=xhat, gat=gat, og1=og1,
                   sig_pre=sig_pre, r1=r1, r2=r2, beta=beta, rgb_in=rgb_in, mask=mask))
    return torch.cat([rgb, sigma], -1), sv


def backward_rgb_feat(p, sv, d_raw):
    """d_raw [R,S,4] -> d rgb_feat [R,S,V,35]."""
    W = lambda n: p[n + '.weight']
    mask = sv['mask']
    V = mask.shape[2]
    d_rgb = d_raw[..., :3]                                            # [R,S,3]
    d_sigma = d_raw[..., 3:4]                                         # [R,S,1]

    # ---- colour branch: blend softmax over views, rgb_fc (1 <- 8 <- 16 <- 37)
    d_beta = (sv['rgb_in'] * d_rgb[:, :, None, :]).sum(-1, keepdim=True)          # [R,S,V,1]
    beta = sv['beta']
    d_y = beta * (d_beta - (beta * d_beta).sum(2, keepdim=True))
    d_y = d_y * (mask != 0).float()                                               # masked_fill blocks the gradient
    d_r2 = (d_y @ W('rgb_fc.4')) * _elu_grad_from_out(sv['r2'])
    d_r1 = (d_r2 @ W('rgb_fc.2')) * _elu_grad_from_out(sv['r1'])
    d_yin = d_r1 @ W('rgb_fc.0')                                                  # [R,S,V,37]
    d_x2 = d_yin[..., :32]
    d_vis2 = d_yin[..., 32:33]

    # ---- density branch: out_geometry_fc, LayerNorm, ray attention, geometry_fc
    live = (sv['n_valid'] >= 1).float() * (sv['sig_pre'] > 0).float()
    d_sig_pre = d_sigma * live
    d_og1 = (d_sig_pre @ W('out_geometry_fc.2')) * _elu_grad_from_out(sv['og1'])
    d_gat = d_og1 @ W('out_geometry_fc.0')                                        # [R,S,16]
    d_xhat = d_gat * p['ray_attention.layer_norm.weight']
    xhat, rstd = sv['xhat'], sv['rstd']
    d_pre = rstd * (d_xhat - d_xhat.mean(-1, keepdim=True) - xhat * (d_xhat * xhat).mean(-1, keepdim=True))
    d_gpe = d_pre.clone()                                                         # residual
    d_o = d_pre @ p['ray_attention.fc.weight']                                    # [R,S,16]
    R, S, _ = d_o.shape
    d_o_h = d_o.view(R, S, 4, 4).transpose(1, 2)                                  # [R,4,S,4]
    attn, q, k, v = sv['attn'], sv['q'], sv['k'], sv['v']
    d_v = attn.transpose(2, 3) @ d_o_h                                            # [R,4,S,4]
    d_attn = d_o_h @ v.transpose(2, 3)                                            # [R,4,S,S]
    d_scores = attn * (d_attn - (attn * d_attn).sum(-1, keepdim=True))
    d_scores = d_scores * sv['row_on'][:, None]                                   # masked query rows: no grad
    d_q = (d_scores @ k) / 2.0
    d_k = d_scores.transpose(2, 3) @ (q / 2.0)
    flat = lambda t: t.transpose(1, 2).reshape(R, S, 16)
    d_gpe = d_gpe + flat(d_q) @ p['ray_attention.w_qs.weight'] + flat(d_k) @ p['ray_attention.w_ks.weight'] \
        + flat(d_v) @ p['ray_attention.w_vs.weight']
    d_g = d_gpe * _elu_grad_from_out(sv['g'])
    d_g1 = (d_g @ W('geometry_fc.2')) * _elu_grad_from_out(sv['g1'])
    d_gin = d_g1 @ W('geometry_fc.0')                                             # [R,S,65]
    d_mean2 = d_gin[..., None, :32]
    d_var2 = d_gin[..., None, 32:64]
    d_wmean = d_gin[..., None, 64:65]

    # ---- second pooling (weights w2 depend on the features through vis2)
    x2, w2, mean2 = sv['x2'], sv['w2'], sv['mean2']
    dev2 = x2 - mean2
    d_mean2_tot = d_mean2 + d_var2 * (-2.0 * (w2 * dev2).sum(2, keepdim=True))
    d_x2 = d_x2 + w2 * (d_mean2_tot + 2.0 * dev2 * d_var2)
    d_w2 = (x2 * d_mean2_tot).sum(-1, keepdim=True) + (dev2 ** 2 * d_var2).sum(-1, keepdim=True) + d_wmean / V
    d_vis2 = d_vis2 + (d_w2 - (d_w2 * w2).sum(2, keepdim=True)) / sv['vsum']

    # ---- vis_fc2 (1 <- 32 <- 32) on x2 * vis1
    d_z2 = d_vis2 * mask * sv['sig2'] * (1.0 - sv['sig2'])
    d_u = (d_z2 @ W('vis_fc2.2')) * _elu_grad_from_out(sv['u'])
    d_xvis = d_u @ W('vis_fc2.0')                                                 # grad of (x2 * vis1)
    d_x2 = d_x2 + d_xvis * sv['vis1']
    d_vis1 = (d_xvis * x2).sum(-1, keepdim=True)

    # ---- vis_fc (33 <- 32 <- 32) on h * w ; x2 = h + xv[:32]
    d_xv = torch.cat([d_x2, d_vis1 * mask * sv['sig1'] * (1.0 - sv['sig1'])], -1) * _elu_grad_from_out(sv['xv'])
    d_v1 = (d_xv @ W('vis_fc.2')) * _elu_grad_from_out(sv['v1'])
    d_h = d_x2 + (d_v1 @ W('vis_fc.0')) * sv['w']

    # ---- base_fc (32 <- 64 <- 105) and the first pooling (weights w are constants)
    d_h1 = ((d_h * _elu_grad_from_out(sv['h'])) @ W('base_fc.2')) * _elu_grad_from_out(sv['h1'])
    d_xin = d_h1 @ W('base_fc.0')                                                 # [R,S,V,105]
    d_mean = d_xin[..., :35].sum(2, keepdim=True)
    d_var = d_xin[..., 35:70].sum(2, keepdim=True)
    f, w, mean = sv['f'], sv['w'], sv['mean']
    dev = f - mean
    d_mean_tot = d_mean + d_var * (-2.0 * (w * dev).sum(2, keepdim=True))
    d_f = d_xin[..., 70:] + w * (d_mean_tot + 2.0 * dev * d_var)
    d_rgb_feat = d_f.clone()
    d_rgb_feat[..., :3] = d_rgb_feat[..., :3] + sv['beta'] * d_rgb[:, :, None, :]
    return d_rgb_feat
