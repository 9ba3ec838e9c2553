"""CPU: the oracle (oracle/) replayed on every committed golden vector of the reference.

These are the pins that make the oracle trustworthy as the checker of the HIP path.
"""
import numpy as np
import pytest
import torch

from fixtures import END_TO_END_ONLY, Golden, STAGE_CASES, assert_close
from oracle import attack_ref as atk
from oracle import feature_net_ref as fnet
from oracle import ibrnet_ref as ib


@pytest.mark.parametrize('case', [c for c in STAGE_CASES if c not in END_TO_END_ONLY])
def test_stages_match_reference(case):
    g = Golden(case)
    cfg = g.stage_cfg()
    rb = g.ray_batch()
    pts, z = ib.sample_along_camera_ray(rb['ray_o'], rb['ray_d'], rb['depth_range'], cfg['S'],
                                        inv_uniform=cfg['inv_uniform'], det=True)
    assert_close(pts, g.np('coarse/pts'), 1e-6, 1e-6, 'pts')
    assert_close(z, g.np('coarse/z'), 1e-6, 1e-6, 'z')
    fm = g.t('in/featmap_coarse')
    rgb_feat, ray_diff, mask, pix = ib.projector_compute(pts, rb['camera'], rb['src_rgbs'], rb['src_cameras'], fm,
                                                         return_pixels=True)
    assert_close(pix, g.np('coarse/pix'), 1e-5, 1e-3, 'pix')
    assert np.array_equal(mask.numpy(), g.np('coarse/mask')), 'mask'
    assert_close(rgb_feat, g.np('coarse/rgb_feat'), 1e-4, 1e-5, 'rgb_feat')
    assert_close(ray_diff, g.np('coarse/ray_diff'), 1e-4, 1e-5, 'ray_diff')
    raw, aux = ib.ibrnet_forward(g.params('coarse'), g.t('coarse/rgb_feat'), g.t('coarse/ray_diff'),
                                 g.t('coarse/mask'), cfg['anti_alias_pooling'], return_aux=True)
    assert_close(aux['base'], g.np('coarse/aux_base'), 1e-4, 1e-5, 'base_fc out')
    assert_close(aux['globalfeat'], g.np('coarse/aux_globalfeat'), 1e-4, 1e-5, 'geometry_fc out')
    assert_close(aux['attn_out'], g.np('coarse/aux_attn_out'), 1e-4, 1e-5, 'ray attention out')
    assert_close(raw, g.np('coarse/raw'), 1e-4, 1e-5, 'raw')


def test_manual_bilinear_equals_grid_sample():
    g = Golden('ibrnet_tiny_invu')
    cfg = g.stage_cfg()
    pix = g.t('coarse/pix')                      # [V,R,S,2]
    fm = g.t('in/featmap_coarse')
    V = cfg['V']
    px = pix[..., 0].reshape(V, -1) * (cfg['Wf'] - 1) / (cfg['W'] - 1)
    py = pix[..., 1].reshape(V, -1) * (cfg['Hf'] - 1) / (cfg['H'] - 1)
    mine = ib.bilinear_zero_pad(fm, px, py)      # [V,C,N]
    ref = g.t('coarse/rgb_feat')[..., 3:].permute(2, 3, 0, 1).reshape(V, 32, -1)
    assert_close(mine, ref, 1e-4, 2e-5, 'manual bilinear')


@pytest.mark.parametrize('case', STAGE_CASES + ['ibrnet_c5_v8'])
def test_render_rays_loss_and_grads(case):
    g = Golden(case)
    cfg = g.stage_cfg()
    rb = g.ray_batch()
    fm_c = g.t('in/featmap_coarse').requires_grad_(True)
    fm_f = g.t('in/featmap_fine').requires_grad_(True)
    ret = ib.render_rays(rb, g.params('coarse'), g.params('fine'), (fm_c, fm_f), cfg['S'],
                         inv_uniform=cfg['inv_uniform'], N_importance=cfg['N_importance'], det=True,
                         white_bkgd=cfg['white_bkgd'], anti_alias_pooling=cfg['anti_alias_pooling'])
    loss = ib.criterion(ret['outputs_coarse'], rb)
    if ret['outputs_fine'] is not None:
        loss = loss + ib.criterion(ret['outputs_fine'], rb)
    for level in ('outputs_coarse', 'outputs_fine'):
        if ret[level] is None:
            assert (level + '/rgb') not in g
            continue
        # fine level: 1-ulp differences in the re-sampled depths are amplified by the (ill-conditioned)
        # `exp_dot - min(exp_dot)` pooling weight of the reference (mlp_network.py:238), hence the looser atol
        rt, at = (2e-4, 2e-5) if level == 'outputs_coarse' else (1e-3, 2e-4)
        for k in ('rgb', 'depth', 'weights', 'alpha', 'z_vals'):
            frac = 1e-3 if (level == 'outputs_fine' and k in ('weights', 'alpha')) else 0.0
            assert_close(ret[level][k], g.np('%s/%s' % (level, k)), rt, at, level + '/' + k, frac_ok=frac)
        assert np.array_equal(ret[level]['mask'].numpy(), g.np(level + '/mask'))
    assert_close(loss, g.np('loss'), 1e-4, 1e-6, 'loss')
    grads = torch.autograd.grad(loss, [fm_c, fm_f] if cfg['N_importance'] else [fm_c])
    gc = g.np('grad/featmap_coarse')
    assert_close(grads[0], gc, 1e-3, 1e-4 * float(np.abs(gc).max()), 'd loss / d featmap_coarse')
    if cfg['N_importance']:
        gf = g.np('grad/featmap_fine')
        assert_close(grads[1], gf, 1e-3, 2e-3 * float(np.abs(gf).max()), 'd loss / d featmap_fine')


def test_ray_generation_matches_reference():
    g = Golden('attack_tiny')
    H, W = [int(x) for x in g.np('cfg')[:2]]
    cam = g.t('in/camera')
    K = cam[:, 2:18].reshape(-1, 4, 4)
    c2w = cam[:, 18:34].reshape(-1, 4, 4)
    ro, rd = ib.rays_single_image(H, W, K, c2w)
    assert_close(ro[::97], g.np('image/ray_o'), 1e-6, 1e-6, 'ray_o')
    assert_close(rd[::97], g.np('image/ray_d'), 1e-6, 1e-6, 'ray_d')


def _attack_setup(g):
    H, W, V, R, S, N_imp, cnn_seed, n_adam, n_sign = [int(x) for x in g.np('cfg')]
    cnn = fnet.random_resunet_state(cnn_seed)
    src = {'src_rgbs': g.t('in/src_rgbs'), 'src_cameras': g.t('in/src_cameras')}
    cam = g.t('in/camera')
    ro, rd = ib.rays_single_image(H, W, cam[:, 2:18].reshape(-1, 4, 4), cam[:, 18:34].reshape(-1, 4, 4))
    gt = g.t('in/rgb').reshape(-1, 3)
    picks = g.np('adam/selected_inds')

    def batch(it):
        idx = torch.from_numpy(picks[it])
        return {'ray_o': ro[idx], 'ray_d': rd[idx], 'rgb': gt[idx], 'camera': cam,
                'depth_range': g.t('in/depth_range'), 'src_rgbs': src['src_rgbs'], 'src_cameras': src['src_cameras']}

    cfg = dict(N_samples=S, N_importance=N_imp, inv_uniform=True, white_bkgd=False)
    return cnn, src, batch, cfg, (H, W, V, R, S, N_imp, n_adam, n_sign), (ro, rd, gt, cam)


def test_pixel_picks_follow_randomstate_234():
    g = Golden('attack_tiny')
    H, W, V, R = [int(x) for x in g.np('cfg')[:4]]
    rng = atk.new_pixel_rng()
    picks = np.stack([atk.pick_pixels(rng, H * W, R) for _ in range(g.np('adam/selected_inds').shape[0])])
    assert np.array_equal(picks, g.np('adam/selected_inds'))


def test_resunet_matches_reference():
    g = Golden('attack_tiny')
    cnn, src, batch, cfg, dims, _ = _attack_setup(g)
    with torch.no_grad():
        x = (src['src_rgbs'] + g.t('in/delta0')).squeeze(0).permute(0, 3, 1, 2)
        fc, ff = fnet.resunet_forward(cnn, x)
    assert list(fc.shape) == list(g.np('cnn/shape'))
    assert_close(fc.reshape(-1)[::37][:1000], g.np('cnn/coarse_sample'), 1e-4, 1e-5, 'cnn coarse sample')
    assert_close(ff.reshape(-1)[::41][:1000], g.np('cnn/fine_sample'), 1e-4, 1e-5, 'cnn fine sample')
    assert_close(fc.double().sum(), g.np('cnn/coarse_sum'), 1e-5, 1e-2, 'cnn coarse sum')


def test_attack_adam_and_sign_pgd_match_reference():
    """The PGD loop is chaotic in fp32 (Adam turns noise-level gradients into full-size +-lr steps), so the pin is
    teacher-forced: from the reference's own delta_t the oracle must reproduce grad_t, and from the reference's
    grad_t the oracle's Adam / sign update must reproduce delta_{t+1} and the moments to rounding."""
    g = Golden('attack_tiny')
    cnn, src, batch, cfg, dims, _ = _attack_setup(g)
    n_adam, n_sign = dims[6], dims[7]
    pc, pf = g.params('coarse'), g.params('fine')
    eps = 8.0 / 255.0
    deltas = [g.t('in/delta0'), g.t('adam/delta_1'), g.t('adam/delta_2'), g.t('adam/delta_3')]
    opt = atk.AdamAscent(deltas[0].shape, 1e-3, step_size=4, gamma=0.5)
    for t in range(3):
        d = deltas[t].clone().requires_grad_(True)
        loss, _ = atk.attack_loss(d, cnn, pc, pf, src, batch(t), cfg)
        grad, = torch.autograd.grad(loss, d)
        ref_grad = g.np('adam/grad_iter%d' % t)
        assert_close(loss, g.np('adam/losses')[t], 2e-4, 1e-6, 'loss at iter %d' % t)
        assert_close(grad, ref_grad, 5e-3, 2e-4 * float(np.abs(ref_grad).max()), 'd loss / d delta, iter %d' % t,
                     frac_ok=1e-3)
        stepped = atk.project(opt.step(deltas[t], torch.from_numpy(ref_grad)), src['src_rgbs'], eps)
        assert_close(stepped, deltas[t + 1], 0, 2e-7, 'delta after Adam step %d (reference gradient)' % (t + 1))
    assert_close(opt.m, g.np('adam/exp_avg_3'), 1e-5, 1e-6 * float(np.abs(g.np('adam/exp_avg_3')).max()), 'adam exp_avg after 3 steps')
    assert_close(opt.v, g.np('adam/exp_avg_sq_3'), 1e-5, 1e-6 * float(np.abs(g.np('adam/exp_avg_sq_3')).max()), 'adam exp_avg_sq after 3 steps')
    # free-running trajectory: statistically the same attack (loose by construction)
    delta, losses, snaps, _ = atk.pgd_attack(deltas[0], cnn, pc, pf, src, batch, cfg, n_adam, use_adam=True,
                                             adam_lr=1e-3, lr_step_size=4, lr_gamma=0.5, epsilon=8.0, record=(1,))
    assert_close(snaps[1], deltas[1], 0, 1e-6, 'free-running delta_1', frac_ok=2e-3)
    assert abs(np.mean(losses[-3:]) - np.mean(g.np('adam/losses')[-3:])) < 0.15 * np.mean(g.np('adam/losses')[-3:])
    assert float((delta - g.t('adam/delta_%d' % n_adam)).abs().mean()) < 0.1 * eps
    # sign-PGD: one teacher-forced step, then the free-running losses
    ref_grad = g.t('sign/grad_iter0')
    stepped = atk.project(deltas[0] + (2.0 / 255.0) * torch.sign(ref_grad), src['src_rgbs'], eps)
    assert_close(stepped, g.np('sign/delta_1'), 0, 1e-7, 'delta after sign step 1 (reference gradient)')
    delta, losses, _, _ = atk.pgd_attack(deltas[0], cnn, pc, pf, src, batch, cfg, n_sign, use_adam=False, adv_lr=2.0,
                                         epsilon=8.0)
    assert_close(np.array(losses[:1]), g.np('sign/losses')[:1], 2e-4, 1e-6, 'sign loss 0')
    assert float((delta - g.t('sign/delta_%d' % n_sign)).abs().mean()) < 0.1 * eps


def test_render_single_image_matches_reference():
    g = Golden('attack_tiny')
    cnn, src, batch, cfg, dims, (ro, rd, gt, cam) = _attack_setup(g)
    H, W, V, R, S, N_imp, n_adam, _ = dims
    with torch.no_grad():
        x = (src['src_rgbs'] + g.t('adam/delta_%d' % n_adam)).squeeze(0).permute(0, 3, 1, 2)
        featmaps = fnet.resunet_forward(cnn, x)
        rb = {'ray_o': ro, 'ray_d': rd, 'rgb': gt, 'camera': cam, 'depth_range': g.t('in/depth_range'),
              'src_rgbs': src['src_rgbs'], 'src_cameras': src['src_cameras']}
        ret = ib.render_single_image(H, W, rb, g.params('coarse'), g.params('fine'), featmaps, 1000, S,
                                     inv_uniform=True, N_importance=N_imp, det=True)
    for level in ('outputs_coarse', 'outputs_fine'):
        assert np.array_equal(ret[level]['mask'].numpy(), g.np('image/%s/mask' % level))
        assert_close(ret[level]['rgb'], g.np('image/%s/rgb' % level), 1e-3, 1e-4, level + ' image rgb', frac_ok=2e-3)
        assert_close(ret[level]['depth'], g.np('image/%s/depth' % level), 1e-3, 1e-4, level + ' image depth', frac_ok=2e-3)
    mse = float(torch.mean((ret['outputs_fine']['rgb'] - g.t('in/rgb')[0]) ** 2))
    assert abs(ib.mse2psnr(mse) - float(g.np('image/psnr_fine'))) < 1e-3


def _hybrid_inputs(device='cpu'):
    from nerfool_amd.synthetic import smooth_featmaps
    h = Golden('hybrid_and_pdf')
    g = Golden(str(h.np('base')))
    cfg = g.stage_cfg()
    fm = (g.t('in/featmap_coarse', device), g.t('in/featmap_fine', device))
    fm_clean = (smooth_featmaps(cfg['V'], 32, cfg['Hf'], cfg['Wf'], seed=50).to(device),
                smooth_featmaps(cfg['V'], 32, cfg['Hf'], cfg['Wf'], seed=51).to(device))
    return h, g, cfg, fm, fm_clean


@pytest.mark.parametrize('tag', ['clean_color', 'clean_density'])
def test_render_rays_hybrid_matches_reference(tag):
    h, g, cfg, fm, fm_clean = _hybrid_inputs()
    with torch.no_grad():
        ret = ib.render_rays_hybrid(g.ray_batch(), g.params('coarse'), g.params('fine'), fm, fm_clean, cfg['S'],
                                    tag == 'clean_color', tag == 'clean_density', inv_uniform=cfg['inv_uniform'],
                                    N_importance=cfg['N_importance'], det=True, white_bkgd=cfg['white_bkgd'])
    for level in ('outputs_coarse', 'outputs_fine'):
        for k in ('rgb', 'depth', 'weights', 'z_vals'):
            assert_close(ret[level][k], h.np('%s/%s/%s' % (tag, level, k)), 1e-3, 2e-4, '%s %s %s' % (tag, level, k),
                         frac_ok=1e-3 if k == 'weights' else 0.0)


def test_sample_pdf_matches_reference():
    h = Golden('hybrid_and_pdf')
    for n in (17, 64):
        got = ib.sample_pdf(h.t('pdf/bins'), h.t('pdf/weights'), n, det=True)
        assert_close(got, h.np('pdf/samples_%d' % n), 1e-5, 1e-5, 'sample_pdf %d' % n)


@pytest.mark.parametrize('case', ['tiny', 'medium'])
def test_oracle_float64_gradient_matches_reference_float64(case):
    """attack_grad64.npz holds the reference's d loss / d delta evaluated in float64 (make_golden_grad64.py).  The oracle
    evaluated in float64 must reproduce it to float32-storage accuracy, and the oracle in fp32 must sit at the reference's
    own fp32 noise floor -- this pins the ground truth the GPU gradient-parity bound is stated against."""
    from fixtures import GRAD64_MEDIUM, grad64_medium_inputs
    g64 = Golden('attack_grad64')
    if case == 'tiny':
        g = Golden('attack_tiny')
        H, W, V, R, S, N_imp, cnn_seed = [int(x) for x in g.np('cfg')[:7]]
        data = {k: g.t('in/' + k) for k in ('rgb', 'camera', 'src_rgbs', 'src_cameras', 'depth_range')}
        cnn, pc_, pf_ = fnet.random_resunet_state(cnn_seed), g.params('coarse'), g.params('fine')
        delta0, picks = g.t('in/delta0'), g.np('adam/selected_inds')[0]
    else:
        c = GRAD64_MEDIUM
        H, W, S, N_imp = c['H'], c['W'], c['S'], c['N_imp']
        data, cnn, pc_, pf_, delta0, picks = grad64_medium_inputs()
    cam = data['camera']
    ro, rd = ib.rays_single_image(H, W, cam[:, 2:18].reshape(-1, 4, 4), cam[:, 18:34].reshape(-1, 4, 4))
    idx = torch.from_numpy(np.asarray(picks, dtype=np.int64))
    cfg = dict(N_samples=S, N_importance=N_imp, inv_uniform=True, white_bkgd=False)

    def run(dtype):
        c_ = lambda t: t.to(dtype) if torch.is_tensor(t) and t.is_floating_point() else t
        batch = {'ray_o': c_(ro[idx]), 'ray_d': c_(rd[idx]), 'rgb': c_(data['rgb'].reshape(-1, 3)[idx]), 'camera': c_(cam),
                 'depth_range': c_(data['depth_range']), 'src_rgbs': c_(data['src_rgbs']), 'src_cameras': c_(data['src_cameras'])}
        src = {'src_rgbs': c_(data['src_rgbs']), 'src_cameras': c_(data['src_cameras'])}
        d = c_(delta0).clone().requires_grad_(True)
        loss, _ = atk.attack_loss(d, {k: c_(v) for k, v in cnn.items()}, {k: c_(v) for k, v in pc_.items()},
                                  {k: c_(v) for k, v in pf_.items()}, src, batch, cfg)
        grad, = torch.autograd.grad(loss, d)
        return float(loss.detach()), grad.double().numpy()

    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    ref64 = g64.np(case + '/grad64').astype(np.float64)
    loss64, grad64 = run(torch.float64)
    assert abs(loss64 - float(g64.np(case + '/loss64'))) < 1e-9
    assert rel(grad64, ref64) < 2e-7, 'oracle float64 vs reference float64: %.3e' % rel(grad64, ref64)
    floor = float(g64.np(case + '/floor/grad'))
    assert abs(rel(g64.np(case + '/grad32').astype(np.float64), ref64) - floor) < 1e-6
    loss32, grad32 = run(torch.float32)
    assert rel(grad32, ref64) < 3 * floor, 'oracle fp32 %.3e vs floor %.3e' % (rel(grad32, ref64), floor)


@pytest.mark.parametrize('tag', ['c1', 'u1'])      # (s1, the sign-PGD loop, runs against the HIP path in the GPU suite; 22 s each here)
def test_oracle_whole_attack_outcome(tag):
    """The OUTCOME of a whole attack (tests/golden/attack100_c1.npz: the reference's eval_adv.py:781-843 loop, 100 Adam-ascent
    iterations at BASELINE config 1's shape, then :863-886's render of the attacked sources and its PSNR -- run by the reference in
    float32, in float64 and in float32 with another summation order).  The oracle's free-running loop on the same seeded inputs must
    end no further from the reference's float64 run than twice the reference's own run-to-run distance, statistic by statistic
    (tests/parity_cases.py:ATTACK100_BARS); the GPU test asserts the same of the HIP path."""
    import parity_cases as pcases
    from fixtures import ATTACK100, attack100_inputs
    from fixtures import second_target_view
    g = Golden('attack100_' + tag)
    c = ATTACK100[tag]
    mode = c.get('mode', 'adam')
    data, cnn, pc, pf, delta0 = attack100_inputs(c)
    views = [data, second_target_view(data)] if mode == 'universal' else [data]
    cam = data['camera']
    rays = []
    for v in views:
        vc = v['camera']
        o, d = ib.rays_single_image(c['H'], c['W'], vc[:, 2:18].reshape(-1, 4, 4), vc[:, 18:34].reshape(-1, 4, 4))
        rays.append((o, d, v['rgb'].reshape(-1, 3), vc))
    ro, rd, gt, _ = rays[0]
    src = {'src_rgbs': data['src_rgbs'], 'src_cameras': data['src_cameras']}
    rng = atk.new_pixel_rng()
    cfg = dict(N_samples=c['S'], N_importance=c['N_imp'], inv_uniform=True, white_bkgd=False)

    def batch(it):       # the universal loop cycles over the target views (eval_adv.py:646-740)
        o, d, rgb, vc = rays[it % len(views)]
        idx = torch.from_numpy(atk.pick_pixels(rng, c['H'] * c['W'], c['N_rand']))
        return {'ray_o': o[idx], 'ray_d': d[idx], 'rgb': rgb[idx], 'camera': vc, 'depth_range': data['depth_range'],
                'src_rgbs': src['src_rgbs'], 'src_cameras': src['src_cameras']}
    n_steps = c['adv_iters'] + (1 if mode == 'universal' else 0)
    delta, losses, _, _ = atk.pgd_attack(delta0, cnn, pc, pf, src, batch, cfg, n_steps, use_adam=mode != 'sign', adam_lr=c['adam_lr'],
                                         lr_step_size=c['lr_step_size'], lr_gamma=c['lr_gamma'], adv_lr=float(c.get('adv_lr', 2)),
                                         epsilon=float(c['epsilon']))
    with torch.no_grad():
        featmaps = fnet.resunet_forward(cnn, (src['src_rgbs'] + delta).squeeze(0).permute(0, 3, 1, 2))
        rb = {'ray_o': ro, 'ray_d': rd, 'rgb': gt, 'camera': cam, 'depth_range': data['depth_range'],
              'src_rgbs': src['src_rgbs'], 'src_cameras': src['src_cameras']}
        ret = ib.render_single_image(c['H'], c['W'], rb, pc, pf, featmaps, c['chunk_size'], c['S'], inv_uniform=True,
                                     N_importance=c['N_imp'], det=True)
    image = ret['outputs_fine']['rgb'].double().numpy()
    mine = dict(losses=np.array(losses), delta=delta.double().numpy().reshape(-1)[::c['delta_stride']], image=image,
                psnr=float(-10. * np.log10(np.mean((image - data['rgb'][0].double().numpy()) ** 2))))
    pcases.attack100_compare(tag, 'oracle (PyTorch-CPU restatement)', mine, g, c['epsilon'] / 255.0)
