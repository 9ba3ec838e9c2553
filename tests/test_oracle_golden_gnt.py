"""CPU: the GNT oracle (oracle/gnt_ref.py) replayed on the golden vectors captured from the reference's gnt/ package."""
import numpy as np
import pytest
import torch

from fixtures import Golden, assert_close
from oracle import gnt_ref as gr

GNT_CASES = ['gnt_tiny_d2_v4', 'gnt_tiny_d3_v5', 'gnt_c4_d8_v10']      # the last: BASELINE config 4's network shape


def gnt_batch(g, device='cpu'):
    return {'ray_o': g.t('in/ray_o', device), 'ray_d': g.t('in/ray_d', device), 'rgb': g.t('in/gt_rgb', device),
            'camera': g.t('in/camera', device), 'depth_range': g.t('in/depth_range', device),
            'src_rgbs': g.t('in/src_rgbs', device), 'src_cameras': g.t('in/src_cameras', device)}


@pytest.mark.parametrize('case', GNT_CASES)
def test_gnt_network_and_renderer_match_reference(case):
    g = Golden(case)
    H, W, V, R, S, depth, Hf, Wf = [int(x) for x in g.np('cfg')]
    p = g.params('net')
    rgb = gr.gnt_forward(p, g.t('net_in/rgb_feat'), g.t('net_in/ray_diff'), g.t('net_in/mask'), g.t('net_in/pts'),
                         g.t('in/ray_d'), depth)
    assert_close(rgb, g.np('rgb'), 1e-4, 1e-5, 'GNT rgb (network only)')
    fm = g.t('in/featmap').requires_grad_(True)
    rb = gnt_batch(g)
    ret = gr.render_rays(rb, p, (fm, fm), S, depth, inv_uniform=True, det=True)
    assert ret['outputs_fine'] is None and ret['outputs_coarse']['weights'] is None
    assert_close(ret['outputs_coarse']['rgb'], g.np('rgb'), 1e-4, 2e-5, 'GNT rgb (render_rays)')
    loss = gr.criterion(ret['outputs_coarse'], rb)
    assert_close(loss, g.np('loss'), 1e-5, 1e-7, 'loss')
    grad, = torch.autograd.grad(loss, fm)
    ref = g.np('grad/featmap')
    assert_close(grad, ref, 1e-3, 1e-4 * float(np.abs(ref).max()), 'd loss / d featmap')


def test_gnt_ret_alpha_and_hierarchical_sampling_match_reference():
    """ret_alpha = True + N_importance > 0 with a single network (gnt/render_ray.py:249-277, transformer_network.py:196-200)."""
    g = Golden('gnt_alpha_d2_v3')
    H, W, V, R, S, depth, Hf, Wf, N_imp = [int(x) for x in g.np('cfg')]
    fm = g.t('in/featmap').requires_grad_(True)
    rb = gnt_batch(g)
    ret = gr.render_rays(rb, g.params('net'), (fm, fm), S, depth, inv_uniform=True, det=True, N_importance=N_imp, ret_alpha=True)
    for lvl in ('outputs_coarse', 'outputs_fine'):
        assert_close(ret[lvl]['rgb'], g.np(lvl + '/rgb'), 1e-4, 2e-5, lvl + ' rgb')
        assert_close(ret[lvl]['weights'], g.np(lvl + '/weights'), 1e-4, 1e-6, lvl + ' weights')
        assert_close(ret[lvl]['depth'], g.np(lvl + '/depth'), 1e-4, 1e-5, lvl + ' depth')
    loss = gr.criterion(ret['outputs_coarse'], rb) + gr.criterion(ret['outputs_fine'], rb)
    assert_close(loss, g.np('loss'), 1e-5, 1e-7, 'loss')
    grad, = torch.autograd.grad(loss, fm)
    ref = g.np('grad/featmap')
    assert_close(grad, ref, 1e-3, 1e-4 * float(np.abs(ref).max()), 'd loss / d featmap')


def gnt_train_inputs(g, tag, device='cpu'):
    """(params, rgb_feat, ray_diff, mask, pts, ray_d, depth, upstream weights) of one case of gnt_train_d2.npz: the network weights of
    its base fixture on the network-level capture of its geometry fixture"""
    if str(g.np(tag + '/base')) == 'seeded':        # gnt_train_mfma_d2.npz: every input regenerated from seeds (tests/fixtures.py)
        from fixtures import GNT_TRAIN_MFMA, gnt_train_mfma_inputs
        p, rgb_feat, ray_diff, mask, pts, ray_d = gnt_train_mfma_inputs()
        to = lambda t: t.to(device)
        from collections import OrderedDict
        return (OrderedDict((k, to(v)) for k, v in p.items()), to(rgb_feat), to(ray_diff), to(mask), to(pts), to(ray_d), GNT_TRAIN_MFMA['depth'],
                g.t(tag + '/w', device))
    base, geo = Golden(str(g.np(tag + '/base'))), Golden(str(g.np(tag + '/geometry')))
    depth = int(base.np('cfg')[5])
    return (base.params('net', device), geo.t('net_in/rgb_feat', device), geo.t('net_in/ray_diff', device), geo.t('net_in/mask', device),
            geo.t('net_in/pts', device), geo.t('in/ray_d', device), depth, g.t(tag + '/w', device))


@pytest.mark.parametrize('fixture', ['gnt_train_d2', 'gnt_train_mfma_d2'])
@pytest.mark.parametrize('tag', ['plain', 'alpha'])
def test_gnt_training_mode_dropout_matches_reference(tag, fixture):
    """The reference's universal GNT loop runs with Dropout(0.1) live (eval/gnt/eval_adv.py:739-878 before switch_to_eval at :959;
    gnt/transformer_network.py:45-48, 72/85-88, 136/162-166).  (1) EXACT: the reference network in train() mode with its nn.Dropout
    instances replaced by the counter-based masks (tests/golden/make_golden_gnt_train.py) against the oracle with the same (seed, site,
    index) masks -- output and d out / d rgb_feat.  (2) STATISTICAL: over 400 seeds the oracle's outputs have the mean and the spread of
    400 draws of the reference with its OWN torch-generator Dropout (4.5 standard errors per element; spread within 25 %)."""
    g = Golden(fixture)          # (gnt_train_mfma_d2: 32 samples per ray, the shape the matrix-core kernels are pinned on)
    p, rgb_feat, ray_diff, mask, pts, ray_d, depth, w = gnt_train_inputs(g, tag)
    alpha = tag == 'alpha'
    pd = float(g.np('p'))
    for seed in [int(x) for x in g.np(tag + '/seeds')]:
        x = rgb_feat.clone().requires_grad_(True)
        y = gr.gnt_forward(p, x, ray_diff, mask, pts, ray_d, depth, ret_alpha=alpha, dropout=(seed, pd))
        assert_close(y, g.np('%s/exact/%d/out' % (tag, seed)), 1e-4, 2e-5, 'train-mode output, seed %d' % seed)
        grad, = torch.autograd.grad((y * w).sum(), x)
        ref = g.np('%s/exact/%d/d_rgb_feat' % (tag, seed))
        assert_close(grad, ref, 1e-3, 1e-4 * float(np.abs(ref).max()), 'train-mode d out / d rgb_feat, seed %d' % seed)
    with torch.no_grad():
        assert_close(gr.gnt_forward(p, rgb_feat, ray_diff, mask, pts, ray_d, depth, ret_alpha=alpha), g.np(tag + '/eval'), 1e-4, 2e-5, 'eval output')
        n_draws = 400 if fixture == "gnt_train_d2" else 60          # (the 32-sample case costs 16x the attention work per draw)
        draws = torch.stack([gr.gnt_forward(p, rgb_feat, ray_diff, mask, pts, ray_d, depth, ret_alpha=alpha, dropout=(1000 + s, pd))
                             for s in range(n_draws)])
    mean, std, n = g.np(tag + '/stat/mean'), g.np(tag + '/stat/std'), int(g.np(tag + '/stat/n'))
    se = np.sqrt(std ** 2 / n + draws.std(0).numpy() ** 2 / n_draws) + 1e-6
    zmax = float(np.abs((draws.mean(0).numpy() - mean) / se).max())
    ratio = float(draws.std(0).numpy().mean() / std.mean())
    print('[gnt train mode] %s: max |z| of the mean over %d elements %.2f; spread ratio %.3f' % (tag, mean.size, zmax, ratio))
    assert zmax <= 4.5 and 0.8 <= ratio <= 1.25


def test_oracle_whole_gnt_attack_outcome():
    """the whole view-specific GNT attack (tests/golden/attack100_g1.npz: the reference's loop of eval/gnt/eval_adv.py:967-1054 in eval
    mode, 100 iterations, render, PSNR -- in float32, float64 and float32 with another summation order): the oracle's free-running
    loop ends no further from the reference's float64 run than twice the reference's own run-to-run distance"""
    import parity_cases as pcases
    from fixtures import ATTACK100, attack100_gnt_inputs
    from oracle import attack_ref as atk, feature_net_ref as fnet, ibrnet_ref as ib
    g = Golden('attack100_g1')
    c = ATTACK100['g1']
    data, cnn, p, delta0 = attack100_gnt_inputs(c)
    cam = data['camera']
    ro, rd = ib.rays_single_image(c['H'], c['W'], cam[:, 2:18].reshape(-1, 4, 4), cam[:, 18:34].reshape(-1, 4, 4))
    gt = data['rgb'].reshape(-1, 3)
    src = {'src_rgbs': data['src_rgbs'], 'src_cameras': data['src_cameras']}
    rng = atk.new_pixel_rng()
    eps = c['epsilon'] / 255.0
    delta = delta0.clone()
    opt = atk.AdamAscent(delta.shape, c['adam_lr'], c['lr_step_size'], c['lr_gamma'])
    losses = []
    for it in range(c['adv_iters']):
        idx = torch.from_numpy(atk.pick_pixels(rng, c['H'] * c['W'], c['N_rand']))
        batch = {'ray_o': ro[idx], 'ray_d': rd[idx], 'rgb': gt[idx], 'camera': cam, 'depth_range': data['depth_range']}
        d = delta.clone().requires_grad_(True)
        fm = fnet.resunet_forward(cnn, (src['src_rgbs'] + d).squeeze(0).permute(0, 3, 1, 2), coarse_out_ch=32, fine_out_ch=0)[0]
        ret = gr.render_rays(batch, p, (fm, fm), c['S'], c['depth'], inv_uniform=True, det=True, src_ray_batch=src)
        loss = gr.criterion(ret['outputs_coarse'], batch)
        grad, = torch.autograd.grad(loss, d)
        losses.append(float(loss))
        delta = atk.project(opt.step(delta, grad), src['src_rgbs'], eps)
    with torch.no_grad():
        fm = fnet.resunet_forward(cnn, (src['src_rgbs'] + delta).squeeze(0).permute(0, 3, 1, 2), coarse_out_ch=32, fine_out_ch=0)[0]
        rows = []
        for i in range(0, ro.shape[0], c['chunk_size']):
            b = {'ray_o': ro[i:i + c['chunk_size']], 'ray_d': rd[i:i + c['chunk_size']], 'camera': cam, 'depth_range': data['depth_range']}
            rows.append(gr.render_rays(b, p, (fm, fm), c['S'], c['depth'], inv_uniform=True, det=True, src_ray_batch=src)['outputs_coarse']['rgb'])
    image = torch.cat(rows).reshape(c['H'], c['W'], 3).double().numpy()
    mine = dict(losses=np.array(losses), delta=delta.double().numpy().reshape(-1)[::c['delta_stride']], image=image,
                psnr=float(-10. * np.log10(np.mean((image - data['rgb'][0].double().numpy()) ** 2))))
    pcases.attack100_compare('g1', 'GNT oracle (PyTorch-CPU restatement)', mine, g, eps)
