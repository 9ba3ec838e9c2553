#!/usr/bin/env python
"""Round-6 golden vectors, produced by the REFERENCE (imported from /root/reference, see _refimport.py) through the loop of
make_golden_r05.reference_attack (eval/ibrnet/eval_adv.py:781-843 -> :863-886 -> PSNR):

  c2full  BASELINE config 2 at its real frame size (756 x 1008, V 4, 64 + 64 samples, N_rand 512, eps 8/255, Adam 1e-3), 100 iterations,
          FIVE runs on the same seeded inputs: ref32 (the reference as it is), ref64 (float64), alt32 (oneDNN off), t3_32 (3 intra-op
          threads), t5m32 (5 threads, oneDNN off).  `floor/<a>_vs_<b>/<stat>` for all ten pairs; strided final delta and the attacked image at
          render_stride 4 of ref32 and ref64 -> attack100_c2full.npz

  late    checkpoints of ref32's c1 run LATE in the trajectory (iterations 50 and 99: a few per cent of delta sits on +-eps, the box clamp
          is active, Adam's second moment is tiny): delta_t, both moments before the step, the picks, the reference's gradient and loss,
          delta_t+1 and both moments after -> attack100_c1_late.npz

  center  `sample_mode='center'` pixel picks of the reference's sampler (ibrnet/sample_ray.py:132-152) on the attack_tiny scene
          (48 x 64) and on a non-square 30 x 52 frame, centre ratios 0.8 / 0.5, three consecutive draws from the RandomState(234)
          stream each, and the rays random_sample returns for the first draw -> sampler_center.npz

    python tests/golden/make_golden_r06.py center late c2full

Runs only in the build container (late: ~2 min; c2full: several hours on 6 of 8 cores)."""
import itertools
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from make_golden_r05 import reference_attack  # noqa: E402  (installs the reference import hooks)

from fixtures import ATTACK100, attack100_inputs, attack_outcome_stats  # noqa: E402

LATE_ITERS = (50, 99)


def run_late():
    c = ATTACK100['c1']
    inputs = attack100_inputs(c)
    t0 = time.time()
    r32 = reference_attack(torch.float32, c, inputs, use_ea=True, log='c1 ref32', checkpoints=LATE_ITERS)
    want = np.load(os.path.join(HERE, 'attack100_c1.npz'))
    assert np.array_equal(r32['losses'], want['ref32/losses']), 'this is not the run attack100_c1.npz holds'
    eps = c['epsilon'] / 255.
    out = {'iters': np.array(LATE_ITERS, dtype=np.int64)}
    for t in LATE_ITERS:
        k = r32['checkpoints'][t]
        for name in ('delta', 'exp_avg', 'exp_avg_sq', 'grad', 'delta_next'):
            out['t%d/%s' % (t, name)] = k[name].astype(np.float32)
        for name in ('exp_avg_next', 'exp_avg_sq_next'):           # every 4th element (the update kernel is elementwise)
            out['t%d/%s' % (t, name)] = k[name].reshape(-1)[::4].astype(np.float32)
        out['t%d/picks' % t] = k['picks']
        out['t%d/loss' % t] = np.array(k['loss'])
        out['t%d/adam_step_before' % t] = np.array(k['step'], dtype=np.int64)
        out['t%d/lr' % t] = np.array(k['lr'])
        d, dn = k['delta'], k['delta_next']
        print('late t=%d: loss %.6f  entries at +-eps before %.4f after %.4f | on the [0,1] box after %.4f | rms exp_avg_sq %.3e'
              % (t, k['loss'], (np.abs(d) >= eps * (1 - 1e-6)).mean(), (np.abs(dn) >= eps * (1 - 1e-6)).mean(),
                 ((inputs[0]['src_rgbs'].numpy() + dn <= 0) | (inputs[0]['src_rgbs'].numpy() + dn >= 1)).mean(),
                 np.sqrt((k['exp_avg_sq'] ** 2).mean())), flush=True)
    path = os.path.join(HERE, 'attack100_c1_late.npz')
    np.savez_compressed(path, **out)
    print('%s %.1f KB  (%.0f s)' % (path, os.path.getsize(path) / 1024., time.time() - t0), flush=True)


def run_center():
    import make_golden as mg
    from nerfool_amd.synthetic import make_scene
    tiny = np.load(os.path.join(HERE, 'attack_tiny.npz'))
    data_tiny = {k: torch.from_numpy(tiny['in/' + k]) for k in ('rgb', 'camera', 'src_rgbs', 'src_cameras', 'depth_range')}
    data_tiny['rgb_path'] = ['golden']
    out = {}
    for tag, data, n_rand in (('tiny', data_tiny, 64), ('odd', make_scene(30, 52, 2, seed=3), 40)):
        smp = mg.ref_sample_ray.RaySamplerSingleImage(data, 'cpu')
        for ratio in (0.8, 0.5):
            mg.reset_pixel_rng()
            picks = np.stack([smp.sample_random_pixel(n_rand, 'center', ratio) for _ in range(3)])
            out['%s/r%02d/picks' % (tag, int(ratio * 10))] = picks.astype(np.int64)
            # the stream position after the draws: the next UNIFORM pick
            out['%s/r%02d/next_uniform' % (tag, int(ratio * 10))] = smp.sample_random_pixel(n_rand, 'uniform').astype(np.int64)
            mg.reset_pixel_rng()
            batch = smp.random_sample(n_rand, 'center', ratio)
            assert np.array_equal(np.asarray(batch['selected_inds']), picks[0])
            out['%s/r%02d/ray_d' % (tag, int(ratio * 10))] = batch['ray_d'].numpy()
            out['%s/r%02d/rgb' % (tag, int(ratio * 10))] = batch['rgb'].numpy()
            print('center %s ratio %.1f: picks in [%d, %d], H x W = %d x %d' % (tag, ratio, picks.min(), picks.max(), smp.H, smp.W))
        out[tag + '/n_rand'] = np.array(n_rand)
    path = os.path.join(HERE, 'sampler_center.npz')
    np.savez_compressed(path, **out)
    print('%s %.1f KB' % (path, os.path.getsize(path) / 1024.))


RUNS = (('ref32', dict(dtype=torch.float32, use_ea=True)),
        ('ref64', dict(dtype=torch.float64, use_ea=False)),
        ('alt32', dict(dtype=torch.float32, use_ea=False, mkldnn=False)),
        ('t3_32', dict(dtype=torch.float32, use_ea=False, threads=3)),
        ('t5m32', dict(dtype=torch.float32, use_ea=False, threads=5, mkldnn=False)))


def run_full(tag="c2full", n_threads=6):
    c = ATTACK100[tag]
    inputs = attack100_inputs(c)
    eps = c['epsilon'] / 255.
    t0 = time.time()
    res = {}
    part = os.path.join(HERE, '.%s_partial.npz' % tag)          # finished runs survive an interrupted session (not committed)
    done = dict(np.load(part)) if os.path.exists(part) else {}
    for name, kw in RUNS:
        if name + '/losses' in done:
            res[name] = dict(losses=done[name + '/losses'], delta=done[name + '/delta_full'], image=done[name + '/image_full'],
                             psnr=float(done[name + '/psnr']), psnr_clean=float(done[name + '/psnr_clean']), pick_sum=int(done[name + '/pick_sum']))
            print('%s: %s taken from %s' % (tag, name, part), flush=True)
            continue
        kw = dict(kw)
        dtype = kw.pop('dtype')
        kw.setdefault('threads', n_threads)
        res[name] = r = reference_attack(dtype, c, inputs, log='%s %s' % (tag, name), **kw)
        print('%s: %s done at %.0f s' % (tag, name, time.time() - t0), flush=True)
        done.update({name + '/losses': r['losses'], name + '/delta_full': r['delta'].astype(np.float32), name + '/image_full': r['image'],
                     name + '/psnr': np.array(r['psnr']), name + '/psnr_clean': np.array(r['psnr_clean']),
                     name + '/pick_sum': np.array(r['pick_sum'], dtype=np.int64)})
        np.savez(part, **done)
    assert len({r['pick_sum'] for r in res.values()}) == 1
    out = {'cfg_tag': np.array(tag), 'pick_checksum': np.array(res['ref32']['pick_sum'], dtype=np.int64),
           'runs': np.array([n for n, _ in RUNS])}
    st = c['delta_stride']
    for name, r in res.items():
        out[name + '/losses'] = r['losses']
        out[name + '/psnr'] = np.array(r['psnr'])
        out[name + '/frac_at_eps'] = np.array(float((np.abs(r['delta']) >= eps * (1 - 1e-5)).mean()))
    out['ref64/psnr_clean'] = np.array(res['ref64']['psnr_clean'])
    for name in ('ref32', 'ref64'):
        out[name + '/delta'] = np.asarray(res[name]['delta']).reshape(-1)[::st].astype(np.float32)
        out[name + '/image'] = np.asarray(res[name]['image']).astype(c.get('image_dtype', 'float32'))
    pairs = []
    for (na, ra), (nb, rb) in itertools.combinations(res.items(), 2):
        pair = '%s_vs_%s' % (na, nb)
        pairs.append(pair)
        s = attack_outcome_stats(ra, rb, eps)
        for k, v in s.items():
            out['floor/%s/%s' % (pair, k)] = np.array(v)
        print('%s floor %-15s %s' % (tag, pair, '  '.join('%s %.3e' % kv for kv in sorted(s.items()))), flush=True)
    out['floor_pairs'] = np.array(pairs)
    print('%s: PSNR clean %.3f dB | attacked %s | at +-eps: %s' % (
        tag, res['ref64']['psnr_clean'], ' '.join('%s %.3f' % (n, r['psnr']) for n, r in res.items()),
        ' '.join('%.4f' % float(out[n + '/frac_at_eps']) for n in res)))
    path = os.path.join(HERE, 'attack100_%s.npz' % tag)
    np.savez_compressed(path, **out)
    print('%s %.1f KB  (%.0f s)' % (path, os.path.getsize(path) / 1024., time.time() - t0), flush=True)


if __name__ == '__main__':
    torch.set_num_threads(8)
    for what in (sys.argv[1:] or ['late']):
        if what == 'late':
            run_late()
        elif what == 'center':
            run_center()
        else:
            run_full(what)
