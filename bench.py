#!/usr/bin/env python
"""bench.py -- NeRFool adversarial inner loop on MI355X (BASELINE.json metric, config 2).

One "step" = one PGD iteration of the view-specific IBRNet attack on a synthetic LLFF-'fern'-shaped scene
(756x1008 sources and target, 4 source views, 64 coarse + 64 importance samples, N_rand rays, Adam-ascent lr 1e-3,
eps 8/255): ray picking, ResUNet on the perturbed sources (MIOpen), coarse+fine render, masked MSE, backward to delta,
fused Adam/eps-ball/[0,1] update.  `value` = rays rendered-and-differentiated per second over all ranks.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Multi-GPU (weak scaling): every rank differentiates its own N_rand rays of the step's global batch, the CNN is
replicated, two collectives per step (2-float mask counts, then one RCCL all-reduce of d(delta)).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3      # MI355X fp32 MFMA / vector peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0        # HBM3E spec peak
ROOFLINE_KERNELS = ('nf_ibrnet_fwd', 'nf_ibrnet_bwd', 'nf_ibrnet_fwd_mfma', 'nf_ibrnet_bwd_mfma', 'nf_project_gather_fwd',
                    'nf_project_gather_bwd', 'nf_gnt_fwd', 'nf_gnt_fwd_mfma', 'nf_gnt_bwd', 'nf_gnt_bwd_mfma', 'nf_pgd_adam_step',
                    'nf_conv3x3_wino')


def ibrnet_flops(R, S, V):
    """Algorithmic forward FLOPs of IBRNet.forward on R rays x S samples x V views (SURVEY 8d closed form)."""
    return 2.0 * R * S * (V * 13256 + 6480 + 32 * S)


def pmc_traffic(kernel, a):
    """HBM bytes per launch of `kernel` from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    in their own runs of this command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None when the
    profile does not cover this workload: counters cannot be read from inside the benchmark process."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_traffic.json')
    try:
        with open(path) as f:
            prof = json.load(f)
    except (OSError, ValueError):
        return None
    w = prof.get('workload', {})
    if (w.get('model'), w.get('n_rand'), w.get('height'), w.get('width'), w.get('views')) != (a.model, a.n_rand, a.height, a.width, a.views):
        return None
    entry = prof.get('abi_kernels', {}).get(kernel)
    return None if entry is None else {'hbm_bytes_per_launch': entry['hbm_bytes_per_launch'], 'unit': 'B',
                                       'algorithmic_bytes_per_launch': entry.get('algorithmic_bytes_per_launch'),
                                       'source': 'profiles/r01_pmc_traffic.json'}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--n-rand', type=int, default=512, help='rays per rank per PGD step (reference default N_rand)')
    ap.add_argument('--height', type=int, default=756)
    ap.add_argument('--width', type=int, default=1008)
    ap.add_argument('--views', type=int, default=4)
    ap.add_argument('--samples', type=int, default=64)
    ap.add_argument('--importance', type=int, default=64)
    ap.add_argument('--render-chunks', type=int, default=4, help='4096-ray chunks for the render-throughput leg (0 = skip)')
    ap.add_argument('--model', choices=('ibrnet', 'gnt'), default='ibrnet',
                    help="'gnt' = BASELINE config 4 (GNT depth 8, 800x800, 10 views, 64 samples) -- not the headline line")
    ap.add_argument('--depth', type=int, default=8, help='GNT trans_depth')
    ap.add_argument('--cnn-shard', choices=('view', 'replicated'), default='view',
                    help='N > 1: feature CNN sharded by source view (exchange of feature maps) or replicated on every rank')
    ap.add_argument('--cpu-iters', type=int, default=2, help='timed CPU-oracle PGD iterations for cpu_baseline (0 = skip)')
    return ap.parse_args()


def build_problem(a, dev):
    from nerfool_amd import eval_adv as EA
    from nerfool_amd.ibrnet.model import IBRNetModel
    from nerfool_amd.ibrnet.projection import Projector
    from nerfool_amd.ibrnet.sample_ray import RaySamplerSingleImage
    from nerfool_amd.synthetic import make_scene
    args = SimpleNamespace(anti_alias_pooling=1, N_samples=a.samples, N_importance=a.importance, N_rand=a.n_rand,
                           inv_uniform=True, det=True, white_bkgd=False, epsilon=8, adv_lr=2, use_adam=True, adam_lr=1e-3,
                           lr_step_size=100, lr_gamma=1.0, adv_iters=1000, local_rank=0, coarse_only=False, ckpt_path=None,
                           sample_mode='uniform', center_ratio=0.8, chunk_size=4096)
    torch.manual_seed(0)
    if a.model == 'gnt':
        from nerfool_amd.gnt.model import GNTModel
        args.netwidth, args.trans_depth, args.single_net, args.ret_alpha = 64, a.depth, True, False
        args.N_importance = 0
        data = make_scene(a.height, a.width, a.views, seed=0, blender=True)
        model = GNTModel(args, device=dev)
    else:
        data = make_scene(a.height, a.width, a.views, seed=0)
        model = IBRNetModel(args, device=dev)
        with torch.no_grad():
            for net in (model.net_coarse, model.net_fine):
                net.out_geometry_fc[2].bias += 1.0      # random-init weights: keep sigma / alpha / T non-trivial
    model.switch_to_eval()
    sampler = RaySamplerSingleImage(data, dev)
    src_ray_batch = sampler.get_all()
    return args, data, model, sampler, src_ray_batch, Projector(dev), EA


def cpu_baseline(a, args, data, model):
    """The oracle port (PyTorch-CPU restatement of the reference, oracle/) on the host cores, same workload, bounded:
    1 untimed + `cpu_iters` timed PGD iterations."""
    from oracle import attack_ref as atk
    threads = torch.get_num_threads()
    cnn = {k: v.detach().cpu() for k, v in model.feature_net.state_dict().items()}
    pc = {k: v.detach().cpu() for k, v in model.net_coarse.state_dict().items()}
    pf = {k: v.detach().cpu() for k, v in model.net_fine.state_dict().items()}
    from oracle import ibrnet_ref as ib
    cam = data['camera']
    ro, rd = ib.rays_single_image(a.height, a.width, cam[:, 2:18].reshape(-1, 4, 4), cam[:, 18:34].reshape(-1, 4, 4))
    gt = data['rgb'].reshape(-1, 3)
    src = {'src_rgbs': data['src_rgbs'], 'src_cameras': data['src_cameras']}
    rng = atk.new_pixel_rng()
    cfg = dict(N_samples=a.samples, N_importance=a.importance, inv_uniform=True, white_bkgd=False)

    def batch(_):
        idx = torch.from_numpy(atk.pick_pixels(rng, a.height * a.width, a.n_rand))
        return {'ray_o': ro[idx], 'ray_d': rd[idx], 'rgb': gt[idx], 'camera': cam, 'depth_range': data['depth_range'],
                'src_rgbs': src['src_rgbs'], 'src_cameras': src['src_cameras']}

    delta0 = atk.init_adv_perturb(data['src_rgbs'], 8 / 255., generator=torch.Generator().manual_seed(0)).detach()
    delta, _, _, _ = atk.pgd_attack(delta0, cnn, pc, pf, src, batch, cfg, 1, adam_lr=1e-3, lr_gamma=1.0)
    t0 = time.time()
    atk.pgd_attack(delta, cnn, pc, pf, src, batch, cfg, a.cpu_iters, adam_lr=1e-3, lr_gamma=1.0)
    dt = (time.time() - t0) / a.cpu_iters
    return {'value': a.n_rand / dt, 'unit': 'rays/s', 'cores': threads, 'kind': 'port',
            'sample': '%d timed PGD iterations (after 1 warm-up) of the same workload, oracle/ (PyTorch-CPU restatement of the '
                      'reference), %.2f s/iter => %.0f s per 1000 iters' % (a.cpu_iters, dt, dt * 1000)}


def main():
    a = parse()
    if a.model == 'gnt':          # config 4 defaults unless the user overrode the sizes
        if (a.height, a.width, a.views) == (756, 1008, 4):
            a.height, a.width, a.views = 800, 800, 10
        a.importance, a.cpu_iters = 0, 0
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert world == a.gpus, '--gpus %d but WORLD_SIZE %d (launch with torch.distributed.run for N > 1)' % (a.gpus, world)
    import __graft_entry__ as entry
    if not os.path.exists(entry.LIB):
        entry.build()
    # one rank per GPU; the modulo only matters for the 1-GPU debugging set-up below (several ranks on one device)
    local_dev = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    shard = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # 'nccl' is RCCL.  NERFOOL_DIST_BACKEND=gloo lets two ranks share ONE GPU to exercise the sharded path on a
        # single-GPU box (RCCL refuses duplicate devices); it is a functional check, never a measurement.
        backend = os.environ.get('NERFOOL_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    args, data, model, sampler, src_ray_batch, projector, EA = build_problem(a, dev)
    if world > 1:
        shard = EA.RayShard(shard_views=a.cnn_shard == 'view')
    from nerfool_amd import prof
    attack = EA.PGDAttack(args, model, projector, src_ray_batch, shard=shard)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        attack.step(data)
    barrier()
    # HIP events only around the kernels the roofline table prices (~10 of the ~250 launches of a step): bracketing every
    # launch costs ~1.5 ms of host time per step, which is visible now that the step is close to launch-bound
    timer = prof.KernelTimer(only=ROOFLINE_KERNELS)
    t0 = time.perf_counter()
    with prof.timing(timer):
        for _ in range(a.steps):
            attack.step(data)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    kernels = timer.summary()
    final_loss = float(attack.last_loss)
    # every hand-written launch, timed over two extra steps OUTSIDE the timed region (extra.hand_written_kernel_ms_per_step)
    all_timer = prof.KernelTimer()
    with prof.timing(all_timer):
        for _ in range(2):
            attack.step(data)
    hand_written_ms = sum(k['total_ms'] for k in all_timer.summary().values()) / 2
    barrier()

    # ---- render-throughput leg (forward only, feature maps resident), outside the timed region of the headline value
    render = None
    if a.render_chunks > 0 and rank == 0:
        if a.model == 'gnt':
            from nerfool_amd.gnt.render_ray import render_rays as gnt_render_rays

            def render_rays(rb, model, featmaps, projector, n_samples, **kw):
                kw.pop('N_importance', None)
                return gnt_render_rays(rb, model, featmaps, projector, n_samples, N_importance=0, **kw)
        else:
            from nerfool_amd.ibrnet.render_ray import render_rays
        with torch.no_grad():
            featmaps = model.feature_net((src_ray_batch['src_rgbs'] + attack.delta).squeeze(0).permute(0, 3, 1, 2))
            rays = sampler.get_all()
            chunk = lambda i: {k: (v[i * 4096:(i + 1) * 4096] if k in ('ray_o', 'ray_d', 'rgb') else v) for k, v in rays.items()}
            render_rays(chunk(0), model, featmaps, projector, a.samples, inv_uniform=True, N_importance=a.importance, det=True,
                        src_ray_batch=src_ray_batch)
            torch.cuda.synchronize()
            rtimer = prof.KernelTimer()
            r0 = time.perf_counter()
            with prof.timing(rtimer):
                for i in range(a.render_chunks):
                    render_rays(chunk(i + 1), model, featmaps, projector, a.samples, inv_uniform=True,
                                N_importance=a.importance, det=True, src_ray_batch=src_ray_batch)
            torch.cuda.synchronize()
            rdt = time.perf_counter() - r0
        render = {'rays_per_s': a.render_chunks * 4096 / rdt, 'chunks': a.render_chunks, 'chunk_rays': 4096,
                  'kernels_ms': {k: round(v['mean_ms'], 4) for k, v in rtimer.summary().items()}}

    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    # ---- roofline of the dominant hand-written kernel of the timed region
    V, Sc, Sf, R = a.views, a.samples, a.samples + a.importance, a.n_rand
    table = {}
    wino_direct, wino_bytes = [], []
    for name, k in kernels.items():
        per_launch = []
        for ms, meta in zip(k['ms'], k['meta']):
            if name in ('nf_ibrnet_fwd', 'nf_ibrnet_bwd', 'nf_ibrnet_fwd_mfma', 'nf_ibrnet_bwd_mfma'):
                per_launch.append(('mfma', ibrnet_flops(meta['R'], meta['S'], meta['V']) / (ms * 1e-3) / 1e12))
            elif name == 'nf_project_gather_fwd':
                b = meta['n_pts'] * meta['V'] * (4 * (meta['C'] + 3) * 4 + (3 + meta['C'] + 4 + 1) * 4)
                per_launch.append(('hbm', b / (ms * 1e-3) / 1e9))
            elif name == 'nf_project_gather_bwd':
                b = meta['n_pts'] * meta['V'] * ((3 + meta['C']) * 4 + 4 * meta['C'] * 4)
                per_launch.append(('hbm', b / (ms * 1e-3) / 1e9))
            elif name in ('nf_gnt_fwd', 'nf_gnt_fwd_mfma', 'nf_gnt_bwd', 'nf_gnt_bwd_mfma'):
                fl = 2.0 * meta['R'] * meta['S'] * (meta['V'] * (6336 + meta['depth'] * 9760)
                                                    + meta['depth'] * 98304 + ((meta['depth'] + 1) // 2) * 16256)
                per_launch.append(('mfma', fl / (ms * 1e-3) / 1e12))
            elif name == 'nf_pgd_adam_step':
                per_launch.append(('hbm', meta['n'] * 32 / (ms * 1e-3) / 1e9))
            elif name == 'nf_conv3x3_wino':
                # Winograd F(2x2,3x3): 16 multiply-adds per 2x2 output tile and (c_in, c_out) pair -- the products the kernel
                # puts on the matrix cores (the direct form of the same convolution needs 36: extra.kernels reports that rate too)
                tiles = meta['n_img'] * ((meta['Ho'] + 1) // 2) * ((meta['Wo'] + 1) // 2)
                per_launch.append(('mfma', 2.0 * 16 * meta['c_in'] * meta['c_out'] * tiles / (ms * 1e-3) / 1e12))
                wino_direct.append(2.0 * 9 * meta['c_in'] * meta['c_out'] * meta['n_img'] * meta['Ho'] * meta['Wo'] / (ms * 1e-3) / 1e12)
                wino_bytes.append(4.0 * (meta['n_img'] * (meta['c_in'] * meta['Hi'] * meta['Wi'] + meta['c_out'] * meta['Ho'] * meta['Wo'])
                                         + 16 * meta['c_in'] * meta['c_out']))
        if per_launch:
            bound = per_launch[0][0]
            ach = float(np.mean([x[1] for x in per_launch]))
            peak = PEAK_F32_TFLOPS if bound == 'mfma' else PEAK_HBM_GBS
            table[name] = {'bound': bound, 'achieved': round(ach, 4), 'peak': peak, 'unit': 'TFLOP/s' if bound == 'mfma' else 'GB/s',
                           'frac': round(ach / peak, 5), 'launches': k['launches'], 'mean_ms': round(k['mean_ms'], 4),
                           'total_ms': round(k['total_ms'], 3)}
    if 'nf_conv3x3_wino' in table:
        table['nf_conv3x3_wino']['direct_form_equivalent_tflops'] = round(float(np.mean(wino_direct)), 2)
    dominant = max(table, key=lambda n: table[n]['total_ms']) if table else None
    roofline = None
    if dominant:
        d = table[dominant]
        t = pmc_traffic(dominant, a)
        if t is not None and t.get('algorithmic_bytes_per_launch') is None and dominant == 'nf_conv3x3_wino':
            t['algorithmic_bytes_per_launch'] = int(np.mean(wino_bytes))     # input + output + transformed weights
        # traffic: HBM bytes per launch of the dominant kernel from the committed PMC passes (null when they do not cover this workload)
        roofline = {'kernel': dominant, 'bound': d['bound'], 'achieved': d['achieved'], 'peak': d['peak'], 'unit': d['unit'],
                    'frac': d['frac'], 'traffic': None if t is None else t['hbm_bytes_per_launch'], 'traffic_unit': 'B/launch',
                    'traffic_algorithmic': None if t is None else t.get('algorithmic_bytes_per_launch'),
                    'traffic_source': None if t is None else t['source']}

    rays_per_step = a.n_rand * world
    out = {
        'metric': 'rays/s through the %s PGD attack step (render fwd+bwd + CNN fwd+bwd + delta update), %d src views'
                  % ('IBRNet' if a.model == 'ibrnet' else 'GNT', a.views),
        'value': rays_per_step * a.steps / elapsed,
        'unit': 'rays/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
        'ms_per_step': 1e3 * elapsed / a.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': ('BASELINE config 2: IBRNet view-specific attack, LLFF-fern-shaped synthetic scene %dx%d, %d source '
                                'views, %d+%d samples/ray, N_rand=%d rays per rank per step, Adam lr 1e-3, eps 8/255'
                                % (a.height, a.width, V, Sc, a.importance, a.n_rand)) if a.model == 'ibrnet' else
                               ('BASELINE config 4: GNT depth %d view-specific attack, synthetic scene %dx%d, %d source views, %d '
                                'samples/ray, N_rand=%d rays per rank per step' % (a.depth, a.height, a.width, V, Sc, a.n_rand)),
                   'rays_per_step_all_ranks': rays_per_step, 'parallelism': 'ray-sharded dp%d%s' % (world, ', feature CNN sharded by source view' if world > 1 and a.cnn_shard == 'view' else '')},
        'roofline': roofline,
        'cpu_baseline': None,
        'extra': {'attack_s_per_1000_iters': 1e3 * elapsed / a.steps, 'final_loss': final_loss, 'kernels': table,
                  'hand_written_kernel_ms_per_step': round(hand_written_ms, 4),
                  'render': render},
    }
    if world == 1 and a.cpu_iters > 0:
        out['cpu_baseline'] = cpu_baseline(a, args, data, model)
    print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
