#!/usr/bin/env python
"""Round-5 golden vectors: the OUTCOME of a whole view-specific attack of the REFERENCE (imported from /root/reference, see
_refimport.py) -- eval/ibrnet/eval_adv.py:781-843 (init, Adam-ascent loop, both clamps) -> :863-886 (feature_net(src + delta),
render_single_image) -> :888-905 (PSNR of the attacked render) -- run three times on the same seeded inputs:

  ref32   the reference as it is (float32; EA.optimize_adv_perturb drives the module-global RandomState(234) pixel stream)
  ref64   the same modules in float64 on the same fp32 inputs (same delta0, same pixel picks, Adam in float64)
  alt32   float32 again with oneDNN convolutions disabled (another summation order of the same fp32 arithmetic)

What is stored per case (`attack100_<tag>.npz`, data only): the seeds every input is regenerated from (tests/fixtures.py), the pixel
picks' checksum, the loss of every iteration of the three runs, the final perturbation of ref32 and ref64 (full for the small case,
every `delta_stride`-th element for the larger one), the attacked fine-level image of ref32 and ref64, the PSNRs (clean, attacked),
and the reference's OWN run-to-run distances ("floor/*": ref32 vs ref64, alt32 vs ref64, ref32 vs alt32) in the quantities the GPU
test compares: loss trajectory, mean |delta difference| / eps, fraction of entries at +-eps, sign agreement, image rms, PSNR.

    python tests/golden/make_golden_r05.py [c1] [c2]

Runs only in the build container (c1: ~3 min, c2: ~1 h on 8 cores)."""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import make_golden as mg  # noqa: E402  (installs the reference import hooks)
from make_golden import EA, Projector, ResUNet, npy, ref_net, ref_sample_ray, reset_pixel_rng  # noqa: E402

from fixtures import ATTACK100, attack100_inputs, attack_outcome_stats, second_target_view  # noqa: E402  (also runs on the GPU box)


def reference_attack(dtype, c, inputs, use_ea, mkldnn=True, log=None, threads=None, checkpoints=()):
    """One whole attack of the reference in `dtype`.  use_ea: iterate through the reference's own optimize_adv_perturb (float32
    only: it builds its sampler and draws its pixels itself); otherwise the same body (eval_adv.py:264-310) on explicit picks
    from an identical RandomState(234) stream, with the fp32 rays cast to `dtype`.
    Round 6 (make_golden_r06.py): `threads` = torch's intra-op thread count for this run (another partition of the reductions);
    c['render_stride'] renders every n-th pixel of the attacked image through the reference's own render_stride argument;
    `checkpoints` = iterations t at which the state around the step is kept (delta_t, Adam moments before and after, the
    reference's gradient, the picks, delta_t+1)."""
    from ibrnet.render_image import render_single_image
    from ibrnet.render_ray import render_rays
    data, cnn_sd, pc, pf, delta0 = inputs
    S, N_imp, R = c['S'], c['N_imp'], c['N_rand']
    old = torch.get_default_dtype()
    old_threads = torch.get_num_threads()
    if threads:
        torch.set_num_threads(threads)
    stride = c.get('render_stride', 1)
    ckpt = {}
    cast = lambda t: t.to(dtype) if torch.is_tensor(t) and t.is_floating_point() else t
    try:
        sampler = ref_sample_ray.RaySamplerSingleImage(data, 'cpu')       # fp32 rays: INPUTS of every run
        torch.set_default_dtype(dtype)
        feature_net = ResUNet(coarse_out_ch=32, fine_out_ch=32, coarse_only=False)
        feature_net.load_state_dict(cnn_sd, strict=True)
        feature_net = feature_net.to(dtype).eval()
        model = SimpleNamespace(net_coarse=ref_net(pc, S).to(dtype), net_fine=ref_net(pf, S + N_imp).to(dtype),
                                feature_net=feature_net)
        projector = Projector(device='cpu')
        args = SimpleNamespace(gt_depth_path=None, use_patch_sampling=False, N_rand=R, sample_mode='uniform',
                               center_ratio=0.8, use_pseudo_gt=False, N_samples=S, inv_uniform=True, N_importance=N_imp,
                               det=True, white_bkgd=False, density_loss=0, depth_var_loss=0, depth_diff_loss=0,
                               depth_consistency_loss=0, depth_smooth_loss=0, camera_consistency_loss=0,
                               perturb_camera=False, use_clean_color=False, use_clean_density=False)
        src_ray_batch = {k: cast(v) for k, v in sampler.get_all().items()}
        src = src_ray_batch['src_rgbs']
        eps = torch.tensor(c['epsilon'] / 255., dtype=dtype)
        mode = c.get('mode', 'adam')
        # universal (eval_adv.py:609-740): the loop cycles over the TARGET views of the scene -- here two, sharing the perturbed source
        # views -- and its `iters > adv_iters` test makes it run adv_iters + 1 steps; view-specific loops stay on `data`
        views = [data, second_target_view(data)] if mode == 'universal' else [data]
        samplers = [sampler] + [ref_sample_ray.RaySamplerSingleImage(v, 'cpu') for v in views[1:]]
        n_steps = c['adv_iters'] + (1 if mode == 'universal' else 0)
        rs = np.random.RandomState(234)
        reset_pixel_rng()
        delta = delta0.to(dtype).clone().requires_grad_(True)
        if mode != 'sign':
            opt = torch.optim.Adam([delta], lr=c['adam_lr'])                                  # eval_adv.py:789 / :640
            sched = torch.optim.lr_scheduler.StepLR(opt, step_size=c['lr_step_size'], gamma=c['lr_gamma'])
        alpha = torch.tensor(c.get('adv_lr', 2) / 255., dtype=dtype)
        losses, pick_sum = [], 0
        t0 = time.time()
        with torch.backends.mkldnn.flags(enabled=mkldnn):
            for it in range(n_steps):                                                          # eval_adv.py:796-839 / :646-740
                vi = it % len(views)
                smp = samplers[vi]
                rays_o, rays_d, rgb_all = smp.rays_o.to(dtype), smp.rays_d.to(dtype), cast(smp.rgb)
                picks = rs.choice(smp.H * smp.W, size=(R,), replace=False)
                pick_sum += int(picks.astype(np.int64).sum()) * (it + 1)
                if it in checkpoints:
                    st = opt.state[delta]
                    ckpt[it] = dict(picks=picks.astype(np.int64), delta=delta.data.clone().numpy(), exp_avg=st['exp_avg'].clone().numpy(),
                                    exp_avg_sq=st['exp_avg_sq'].clone().numpy(), step=int(st['step']), lr=float(opt.param_groups[0]['lr']))
                if use_ea:
                    loss, _ = EA.optimize_adv_perturb(args, delta, model, projector, src_ray_batch, views[vi], return_loss=True)
                else:
                    sel = torch.from_numpy(picks.astype(np.int64))
                    batch = {'ray_o': rays_o[sel], 'ray_d': rays_d[sel], 'rgb': rgb_all[sel], 'camera': cast(smp.camera),
                             'depth_range': cast(smp.depth_range), 'src_rgbs': src,
                             'src_cameras': cast(smp.src_cameras), 'selected_inds': sel}
                    featmaps = feature_net((src + delta).squeeze(0).permute(0, 3, 1, 2))
                    ret = render_rays(ray_batch=batch, model=model, projector=projector, featmaps=featmaps, N_samples=S,
                                      inv_uniform=True, N_importance=N_imp, det=True, white_bkgd=False, args=args,
                                      src_ray_batch=src_ray_batch)
                    loss, _ = EA.criterion(ret['outputs_coarse'], batch, scalars_to_log=None)
                    lf, _ = EA.criterion(ret['outputs_fine'], batch, scalars_to_log=None)
                    loss = loss + lf
                if mode == 'sign':                                                             # eval_adv.py:822-828
                    loss.backward()
                    delta.data = delta.data + alpha * torch.sign(delta.grad.detach())
                    delta.grad.zero_()
                else:
                    opt.zero_grad()
                    loss.backward()
                    if it in checkpoints:
                        ckpt[it]['grad'] = delta.grad.detach().clone().numpy()      # d loss / d delta (before the sign flip)
                    delta.grad.data *= -1
                    opt.step()
                    sched.step()
                delta.data = EA.clamp(delta.data, -eps, eps)
                delta.data = EA.clamp(delta.data, 0 - src, 1 - src)
                losses.append(float(loss))
                if it in checkpoints:
                    st = opt.state[delta]
                    ckpt[it].update(loss=float(loss), delta_next=delta.data.clone().numpy(), exp_avg_next=st['exp_avg'].clone().numpy(),
                                    exp_avg_sq_next=st['exp_avg_sq'].clone().numpy())
                if log and (it % 10 == 0 or it + 1 == n_steps):
                    print('  %s iter %3d loss %.7f  (%.1f s)' % (log, it, losses[-1], time.time() - t0), flush=True)
            # the attacked render + the clean one (eval_adv.py:863-886; PSNR as :888-905 / utils.py:35 on the fine image)
            images = {}
            with torch.no_grad():
                render_sampler = sampler if stride == 1 else ref_sample_ray.RaySamplerSingleImage(data, 'cpu', render_stride=stride)
                ray_batch = {k: cast(v) for k, v in render_sampler.get_all().items()}
                for tag, d in (('adv', delta.data), ('clean', torch.zeros_like(delta.data))):
                    featmaps = feature_net((src + d).squeeze(0).permute(0, 3, 1, 2))
                    if tag == 'clean' and c.get('skip_clean_render') and dtype != torch.float64:
                        continue
                    ret = render_single_image(ray_sampler=render_sampler, ray_batch=ray_batch, model=model, projector=projector,
                                              chunk_size=c['chunk_size'], det=True, N_samples=S, inv_uniform=True,
                                              N_importance=N_imp, white_bkgd=False, render_stride=stride, featmaps=featmaps, args=args,
                                              src_ray_batch=src_ray_batch)
                    images[tag] = ret['outputs_fine']['rgb'].double().numpy()
        gt = data['rgb'][0].double().numpy()[::stride, ::stride]
        psnr = {k: float(-10. * np.log10(np.mean((v - gt) ** 2))) for k, v in images.items()}
        return dict(losses=np.array(losses), delta=delta.data.double().numpy().copy(), image=images['adv'],
                    image_clean=images.get('clean'), psnr=psnr['adv'], psnr_clean=psnr.get('clean', float('nan')), pick_sum=pick_sum,
                    checkpoints=ckpt)
    finally:
        torch.set_default_dtype(old)
        torch.set_num_threads(old_threads)


def run_case(tag):
    c = ATTACK100[tag]
    inputs = attack100_inputs(c)
    eps = c['epsilon'] / 255.
    t0 = time.time()
    r32 = reference_attack(torch.float32, c, inputs, use_ea=True, log=tag + ' ref32')
    print('%s: ref32 done in %.0f s' % (tag, time.time() - t0), flush=True)
    if c['H'] <= 128:
        # the restated loop body is the reference's: bit-identical trajectory on the explicit picks
        chk = reference_attack(torch.float32, dict(c, adv_iters=3), inputs, use_ea=False)
        assert np.array_equal(chk['losses'][:3], r32['losses'][:3]), (chk['losses'], r32['losses'][:3])
    r64 = reference_attack(torch.float64, c, inputs, use_ea=False, log=tag + ' ref64')
    print('%s: ref64 done at %.0f s' % (tag, time.time() - t0), flush=True)
    alt = reference_attack(torch.float32, c, inputs, use_ea=False, mkldnn=False, log=tag + ' alt32')
    assert r32['pick_sum'] == r64['pick_sum'] == alt['pick_sum']
    out = {'cfg_tag': np.array(tag), 'pick_checksum': np.array(r32['pick_sum'], dtype=np.int64)}
    st = c['delta_stride']
    for name, r in (('ref32', r32), ('ref64', r64), ('alt32', alt)):
        out[name + '/losses'] = r['losses']
        out[name + '/psnr'] = np.array(r['psnr'])
        out[name + '/psnr_clean'] = np.array(r['psnr_clean'])
        out[name + '/frac_at_eps'] = np.array(float((np.abs(r['delta']) >= eps * (1 - 1e-5)).mean()))
    for name, r in (('ref32', r32), ('ref64', r64)):
        out[name + '/delta'] = r['delta'].reshape(-1)[::st].astype(np.float32)
        out[name + '/image'] = r['image'].astype(c.get('image_dtype', 'float32'))       # (the larger case: half precision, 1e-3 of full scale)
    if c['H'] <= 128:
        out['ref64/image_clean'] = r64['image_clean'].astype(np.float32)
    for name, a, b in (('ref32_vs_ref64', r32, r64), ('alt32_vs_ref64', alt, r64), ('ref32_vs_alt32', r32, alt)):
        s = attack_outcome_stats(a, b, eps)
        for k, v in s.items():
            out['floor/%s/%s' % (name, k)] = np.array(v)
        print('%s floor %-15s %s' % (tag, name, '  '.join('%s %.3e' % kv for kv in sorted(s.items()))), flush=True)
    print('%s: PSNR clean %.3f dB | attacked ref32 %.3f ref64 %.3f alt32 %.3f | at +-eps: %.4f %.4f %.4f | losses first/last %.5f -> %.5f'
          % (tag, r64['psnr_clean'], r32['psnr'], r64['psnr'], alt['psnr'], out['ref32/frac_at_eps'], out['ref64/frac_at_eps'],
             out['alt32/frac_at_eps'], r64['losses'][0], r64['losses'][-1]))
    path = os.path.join(HERE, 'attack100_%s.npz' % tag)
    np.savez_compressed(path, **out)
    print('%s %.1f KB  (%.0f s)' % (path, os.path.getsize(path) / 1024., time.time() - t0), flush=True)


if __name__ == '__main__':
    torch.set_num_threads(8)
    for tag in (sys.argv[1:] or ['c1']):
        run_case(tag)
