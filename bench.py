#!/usr/bin/env python
"""bench.py -- NeRFool adversarial inner loop on MI355X (BASELINE.json metric, config 2).

One "step" = one PGD iteration of the view-specific IBRNet attack on a synthetic LLFF-'fern'-shaped scene
(756x1008 sources and target, 4 source views, 64 coarse + 64 importance samples, N_rand rays, Adam-ascent lr 1e-3,
eps 8/255): ray picking, ResUNet on the perturbed sources, coarse+fine render, masked MSE, backward to delta, fused
Adam/eps-ball/[0,1] update.  `value` = rays rendered-and-differentiated per second over all ranks.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N --steps K --warmup W            # starts its own N rank processes (see spawn_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Multi-GPU: rays are sharded over the ranks (nerfool_amd.eval_adv.RayShard).  `--scaling weak` (default): every rank
differentiates its own N_rand rays (global batch N_rand * N); `--scaling strong`: the N_rand rays of the single-GPU step are
split over the ranks.  `--cnn-shard replicated` (default = the north-star form): the feature CNN runs on every rank, 2
collectives per step (16-byte counts/loss all-reduce + ONE RCCL all-reduce of d delta); `--cnn-shard view`: the feature CNN is
sharded by source view, 4 collectives per step (the 16-byte one, all-gather of the feature maps, reduce-scatter of their
gradients, all-gather of d delta).  Whatever the headline flags, an N > 1 run also times the other three forms and reports them
under `extra.multi_gpu`.  The render legs (`extra.render*`) run on ALL ranks: every rank renders its own contiguous block of
4096-ray chunks (no collective in the data path) and the figure is the all-rank rays/s between barriers; `render_single_image`
is the whole image sharded over the ranks and gathered to rank 0 by one collective.  Prints ONE JSON line on rank 0.
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
PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA peak (MI355X_MICROARCH.md; the 5 PF headline figure includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0        # HBM3E spec peak
PMC_PROFILE = 'profiles/r06_pmc_traffic.json'
# kernels whose matrix products run on the bf16 pipe with every fp32 operand split in three (round 4): priced against the bf16 peak
X3_KERNELS = {
    'nf_conv3x3_wino': 'achieved = executed bf16 products: 6 per Winograd-domain product (three operand parts); each product counted once in '
                       'product_tflops; direct-form rate in direct_form_equivalent_tflops',
    'nf_conv3x3_wino_bwd': 'the backward-data launches of nf_conv3x3_wino_bf on two operand parts: achieved = 3 executed bf16 products per '
                           'Winograd-domain product',
    'nf_ibrnet_fwd_mfma': 'sample-on-the-lane forward: achieved = 6 x the algorithmic FLOPs of IBRNet.forward -- an UPPER bound on the executed '
                          'bf16 FLOPs (this form multiplies the view-invariant part of base_fc.0 once per sample; the per-ray attention runs in fp32)',
    'nf_gnt_fwd_mfma': 'achieved = 6 x the algorithmic FLOPs: upper bound (the streamed GEMMs run split, the attention products in fp32)',
    'nf_gnt_bwd_mfma': 'achieved = 6 x the algorithmic FLOPs: upper bound (the streamed GEMMs run split, the attention products in fp32)',
}
ROOFLINE_KERNELS = ('nf_ibrnet_fwd', 'nf_ibrnet_bwd', 'nf_ibrnet_fwd_mfma', 'nf_ibrnet_bwd_mfma', 'nf_ibrnet_fwd_mfma_bf16',
                    'nf_ibrnet_bwd_mfma_bf16', 'nf_project_gather_fwd',
                    'nf_project_gather_bwd', 'nf_gnt_fwd', 'nf_gnt_fwd_mfma', 'nf_gnt_bwd', 'nf_gnt_bwd_mfma', 'nf_pgd_adam_step',
                    'nf_conv3x3_wino')


def ibrnet_flops(R, S, V):
    """Algorithmic forward FLOPs of IBRNet.forward on R rays x S samples x V views (SURVEY 8d closed form)."""
    return 2.0 * R * S * (V * 13256 + 6480 + 32 * S)


def resunet_flops(H, W):
    """Algorithmic forward FLOPs of the ResUNet per image (FlopCounterMode at the two benchmark sizes;
    other sizes scaled by area)."""
    known = {(756, 1008): 121.9e9, (800, 800): 101.3e9, (512, 512): 41.5e9}
    return known.get((H, W), 121.9e9 * H * W / (756.0 * 1008.0))


def pmc_traffic(kernel, a):
    """HBM bytes per launch of `kernel` from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    in their own runs of this command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None when the
    profile does not cover this workload: counters cannot be read from inside the benchmark process."""
    for rel in (PMC_PROFILE, 'profiles/r05_pmc_traffic.json', 'profiles/r04_pmc_traffic.json'):
        try:
            with open(os.path.join(ROOT, rel)) as f:
                prof = json.load(f)
        except (OSError, ValueError):
            continue
        w = prof.get('workload', {})
        if (w.get('model'), w.get('n_rand'), w.get('height'), w.get('width'), w.get('views')) != (a.model, a.n_rand, a.height, a.width, a.views):
            continue
        entry = prof.get('abi_kernels', {}).get(kernel)
        if entry is None:
            continue
        return {'hbm_bytes_per_launch': entry['hbm_bytes_per_launch'], 'unit': 'B',
                'algorithmic_bytes_per_launch': entry.get('algorithmic_bytes_per_launch'), 'source': rel}
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--n-rand', type=int, default=512, help='rays per PGD step: per rank (weak scaling) or in total (strong)')
    ap.add_argument('--height', type=int, default=756)
    ap.add_argument('--width', type=int, default=1008)
    ap.add_argument('--views', type=int, default=4)
    ap.add_argument('--samples', type=int, default=64)
    ap.add_argument('--importance', type=int, default=64)
    ap.add_argument('--render-chunks', type=int, default=16, help='4096-ray chunks per rank for the render-throughput leg (0 = skip)')
    ap.add_argument('--model', choices=('ibrnet', 'gnt'), default='ibrnet',
                    help="'gnt' = BASELINE config 4 (GNT depth 8, 800x800, 10 views, 64 samples) -- not the headline line")
    ap.add_argument('--config', choices=('c2', 'c4', 'c5'), default='c2',
                    help="BASELINE config: c2 = the headline line (IBRNet, LLFF-fern shape); c4 = --model gnt; c5 = IBRNet DeepVoxels "
                         "shape, 512x512, 8 source views, 128 + 128 samples, white background, bf16 matrix-core operands in the row network")
    ap.add_argument('--precision', choices=('fp32', 'bf16'), default=None, help='IBRNet row-network operand precision (default: fp32; c5: bf16)')
    ap.add_argument('--depth', type=int, default=8, help='GNT trans_depth')
    ap.add_argument('--cnn-shard', choices=('view', 'replicated'), default='replicated',
                    help='N > 1: feature CNN replicated on every rank (north-star form: one all-reduce of d delta; the default) or '
                         'sharded by source view (exchange of feature maps)')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='N > 1: --n-rand rays per rank (weak) or split over the ranks (strong)')
    ap.add_argument('--cpu-iters', type=int, default=10, help='timed CPU-oracle PGD iterations for cpu_baseline after 2 warm-ups (0 = skip)')
    ap.add_argument('--extras', type=int, default=1, help='0 = only the headline timed region (profiling runs)')
    ap.add_argument('--attack-iters', type=int, default=1000,
                    help='iterations of the MEASURED whole-attack leg (extra.attack_<N>_iters_wall_s; single GPU, config 2; 0 = skip)')
    ap.add_argument('--conv-operands', choices=('bf16x3', 'fp32'), default='bf16x3',
                    help="operand form of the 3x3 convolutions' Winograd products: 'bf16x3' (default of the package: every fp32 operand as "
                         "three bf16 parts on the bf16 matrix cores, six cross terms, error at fp32 rounding level) or 'fp32' "
                         "(v_mfma_f32_32x32x2_f32, the form of rounds 1-3)")
    ap.add_argument('--universal-views', type=int, default=8,
                    help='target views of the universal-loop leg (extra.universal: BASELINE config 3 on this many GPUs; 0 = skip)')
    ap.add_argument('--device', choices=('gpu', 'cpu-standin'), default='gpu',
                    help="'cpu-standin': FUNCTIONAL check of the launcher / sharding on a box without a GPU -- the kernels' CPU stand-in "
                         "build (tests/host_harness), gloo, shape-generic kernels; tiny sizes only, never a measurement")
    ap.add_argument('--event-every', type=int, default=10,
                    help='HIP-event brackets around the roofline kernels on every N-th timed step (1 = every step, 0 = never)')
    return ap.parse_args()


def make_args(a, n_rand):
    args = SimpleNamespace(anti_alias_pooling=1, N_samples=a.samples, N_importance=a.importance, N_rand=n_rand,
                           inv_uniform=True, det=True, white_bkgd=False, epsilon=8, adv_lr=2, use_adam=True, adam_lr=1e-3,
                           lr_step_size=100, lr_gamma=1.0, adv_iters=1000, local_rank=0, coarse_only=False, ckpt_path=None,
                           sample_mode='uniform', center_ratio=0.8, chunk_size=4096, ibrnet_precision=a.precision or 'fp32')
    if a.config == 'c5':
        args.white_bkgd = True          # configs/ibrnet/eval_deepvoxels.txt:20
    if a.model == 'gnt':
        args.netwidth, args.trans_depth, args.single_net, args.ret_alpha = 64, a.depth, True, False
        args.N_importance = 0
    return args


def build_problem(a, dev, height=None, width=None):
    from nerfool_amd import eval_adv as EA
    from nerfool_amd.ibrnet.model import IBRNetModel
    from nerfool_amd.ibrnet.projection import Projector
    from nerfool_amd.ibrnet.sample_ray import RaySamplerSingleImage
    from nerfool_amd.synthetic import make_scene
    H, W = height or a.height, width or a.width
    args = make_args(a, a.n_rand)
    torch.manual_seed(0)
    if a.model == 'gnt':
        from nerfool_amd.gnt.model import GNTModel
        data = make_scene(H, W, a.views, seed=0, blender=True)
        model = GNTModel(args, device=dev)
    else:
        # c5: DeepVoxels depth range = origin depth +- 0.8 (ibrnet/data_loaders/deepvoxels.py:134-143)
        data = make_scene(H, W, a.views, seed=0, **({'depth_range': (3.2, 4.8)} if a.config == 'c5' else {}))
        model = IBRNetModel(args, device=dev)
        with torch.no_grad():
            for net in (model.net_coarse, model.net_fine):
                net.out_geometry_fc[2].bias += 1.0      # random-init weights: keep sigma / alpha / T non-trivial
    model.switch_to_eval()
    sampler = RaySamplerSingleImage(data, dev)
    src_ray_batch = sampler.get_all()
    return args, data, model, sampler, src_ray_batch, Projector(dev), EA


def host_cpu():
    """(physical cores, model string) of the host from /proc/cpuinfo"""
    cores, model = set(), None
    try:
        phys = core = None
        with open('/proc/cpuinfo') as f:
            for line in f:
                k, _, v = line.partition(':')
                k, v = k.strip(), v.strip()
                if k == 'model name' and model is None:
                    model = v
                elif k == 'physical id':
                    phys = v
                elif k == 'core id':
                    core = v
                elif not k and phys is not None:
                    cores.add((phys, core))
                    phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    n = len(cores) or (os.cpu_count() or 1)
    return n, model or 'unknown'


def cpu_baseline(a, args, data, model):
    """The oracle port (PyTorch-CPU restatement of the reference, oracle/) on the host's physical cores, same workload,
    bounded: 2 untimed + `cpu_iters` timed PGD iterations, and 3 timed 4096-ray render chunks after 1 warm-up."""
    from oracle import attack_ref as atk
    from oracle import feature_net_ref as fnet
    from oracle import ibrnet_ref as ib
    cores, cpu_model = host_cpu()
    torch.set_num_threads(cores)
    cnn = {k: v.detach().cpu() for k, v in model.feature_net.state_dict().items()}
    pc = {k: v.detach().cpu() for k, v in model.net_coarse.state_dict().items()}
    pf = {k: v.detach().cpu() for k, v in model.net_fine.state_dict().items()}
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
    delta, _, _, _ = atk.pgd_attack(delta0, cnn, pc, pf, src, batch, cfg, 2, adam_lr=1e-3, lr_gamma=1.0)
    t0 = time.time()
    atk.pgd_attack(delta, cnn, pc, pf, src, batch, cfg, a.cpu_iters, adam_lr=1e-3, lr_gamma=1.0)
    dt = (time.time() - t0) / a.cpu_iters
    # render leg: feature maps once, then 4096-ray chunks, forward only
    with torch.no_grad():
        fm = fnet.resunet_forward(cnn, (data['src_rgbs'] + delta).squeeze(0).permute(0, 3, 1, 2))

        def chunk(i):
            sl = slice(i * 4096, (i + 1) * 4096)
            return {'ray_o': ro[sl], 'ray_d': rd[sl], 'rgb': gt[sl], 'camera': cam, 'depth_range': data['depth_range'],
                    'src_rgbs': src['src_rgbs'], 'src_cameras': src['src_cameras']}
        ib.render_rays(chunk(0), pc, pf, fm, a.samples, inv_uniform=True, N_importance=a.importance, det=True)
        r0 = time.time()
        for i in range(3):
            ib.render_rays(chunk(i + 1), pc, pf, fm, a.samples, inv_uniform=True, N_importance=a.importance, det=True)
        rdt = (time.time() - r0) / 3
    return {'value': a.n_rand / dt, 'unit': 'rays/s', 'cores': cores, 'kind': 'port', 'cpu_model': cpu_model,
            'torch_threads': torch.get_num_threads(),
            'sample': '%d timed PGD iterations (after 2 warm-ups) of the same workload, oracle/ (PyTorch-CPU restatement of the '
                      'reference), %.2f s/iter => %.0f s per 1000 iters; render leg: 3 timed 4096-ray chunks after 1 warm-up'
                      % (a.cpu_iters, dt, dt * 1000),
            'attack_s_per_iter': dt, 'render_rays_per_s': 4096 / rdt}


def prime(attack, data):
    """untimed preparation in front of the W warm-up steps of a timed region: a fresh PGDAttack captures its step into a hipGraph on its
    third step on a target view (two eager steps, capture, replay) -- without this the capture would fall inside a timed region
    whose warm-up is shorter than three steps"""
    if getattr(attack, 'use_graph', False) and not getattr(attack, '_bench_primed', False):
        for _ in range(3):
            attack.step(data)
        attack._bench_primed = True


def time_steps(attack, data, steps, warmup, barrier, timer=None, every=1):
    """`warmup` untimed + exactly `steps` timed PGD steps between barrier + synchronize brackets -> seconds (this rank).
    timer: a prof.KernelTimer whose HIP-event brackets are live on every `every`-th timed step (steps 0, every, 2 every, ...): an
    event record is a marker packet that costs the GPU ~5.6 us before and after the bracketed launch (profiles/r02_step_timeline.txt:
    every gap of the step sits next to a bracketed kernel), ~0.7 ms per step with the ~65 bracketed launches of a step -- sampling
    the steps keeps the per-launch durations live and inside the timed region without putting that cost on every step."""
    from nerfool_amd import prof
    prime(attack, data)
    for _ in range(warmup):
        attack.step(data)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        if timer is not None and every > 0 and i % every == 0:
            with prof.timing(timer):
                attack.step(data)
        else:
            attack.step(data)
    barrier()
    return time.perf_counter() - t0


def universal_leg(a, EA, make_attack, data, dev, barrier, max_over_ranks, world):
    """BASELINE config 3's loop on this many GPUs: PGDAttack.run_universal (eval_adv.py:634-740) cycling over `universal_views`
    synthetic TARGET views of the scene, one shared perturbation on the source views.  Each target view gets its own cached sampler
    and its own captured step (graph segments when sharded); reported: the time of the first cycles (two eager steps + the capture per
    view), ms/step once every view replays, and the allocator's peak."""
    from nerfool_amd.synthetic import target_views
    views = target_views(data, a.universal_views)
    uni = make_attack(a.cnn_shard, a.scaling)
    torch.cuda.reset_peak_memory_stats(dev)
    barrier()
    t0 = time.perf_counter()
    uni.run_universal(views, n_iters=3 * len(views) - 1)          # three cycles: eager, eager, capture + replay
    barrier()
    t_capture = max_over_ranks(time.perf_counter() - t0)
    replays0 = uni.graph_replays
    n = 4 * len(views)
    barrier()
    t0 = time.perf_counter()
    uni.run_universal(views, n_iters=n - 1)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    rays = a.n_rand * (world if a.scaling == 'weak' else 1)
    out = {'target_views': len(views), 'ms_per_step': round(1e3 * dt / n, 4), 'rays_per_s': rays * n / dt, 'steps_timed': n,
           'graphs_captured': len(uni._graphs), 'replayed_steps_in_timed_region': uni.graph_replays - replays0,
           'first_three_cycles_s': round(t_capture, 3), 'hbm_peak_allocated_gb': round(torch.cuda.max_memory_allocated(dev) / 1e9, 3),
           'final_loss': float(uni.last_loss), 'n_gpus': world,
           'note': 'one cached sampler (rays + target image) and one captured step per target view, all in one activation pool'}
    del uni
    return out


def render_leg(model, projector, sampler, src_ray_batch, featmaps, n_chunks, samples, importance, gnt, prof, par=None):
    """forward-only throughput on 4096-ray chunks with resident feature maps: 1 warm-up chunk + n_chunks timed PER RANK -- rank r
    renders the contiguous chunk block [r (n_chunks + 1), (r + 1)(n_chunks + 1)) of the image, nothing is exchanged; the rate is
    all ranks' rays over the slowest rank's time between barriers.  par = (rank, world, barrier, max_over_ranks)."""
    rank, world, barrier, max_over_ranks = par if par is not None else (0, 1, SYNC, lambda x: x)
    if gnt:
        from nerfool_amd.gnt.render_ray import render_rays as gnt_render_rays

        def render_rays(rb, model, featmaps, projector, n_samples, **kw):
            kw.pop('N_importance', None)
            return gnt_render_rays(rb, model, featmaps, projector, n_samples, N_importance=0, **kw)
    else:
        from nerfool_amd.ibrnet.render_ray import camera_workspace, render_rays as ibr_render_rays
        # a chunk loop over ONE view builds the cameras' projection workspace once, as render_single_image does
        ws = camera_workspace(sampler.get_all(), src_ray_batch)

        def render_rays(rb, model, featmaps, projector, n_samples, **kw):
            return ibr_render_rays(rb, model, featmaps, projector, n_samples, cam_ws=ws, **kw)
    rays = sampler.get_all()
    total = rays['ray_o'].shape[0] // 4096          # whole chunks of the image; a rank's block wraps around when the image is short
    first = rank * (n_chunks + 1)

    def chunk(i):
        c = (first + i) % total
        return {k: (v[c * 4096:(c + 1) * 4096] if k in ('ray_o', 'ray_d', 'rgb') else v) for k, v in rays.items()}
    with torch.no_grad():
        render_rays(chunk(0), model, featmaps, projector, samples, inv_uniform=True, N_importance=importance, det=True,
                    src_ray_batch=src_ray_batch)
        # timed pass WITHOUT HIP-event brackets: a chunk is ~14 launches of which a dozen take 4-5 us, and an event pair costs the GPU
        # ~11 us around the launch it brackets -- bracketing every launch (as this leg did until round 3) took a quarter of the chunk
        barrier()
        r0 = time.perf_counter()
        for i in range(n_chunks):
            render_rays(chunk(i + 1), model, featmaps, projector, samples, inv_uniform=True, N_importance=importance,
                        det=True, src_ray_batch=src_ray_batch)
        barrier()
        rdt = max_over_ranks(time.perf_counter() - r0)
        # per-kernel durations from a second, bracketed pass over two of the same chunks (outside the timed pass)
        rtimer = prof.KernelTimer()
        with prof.timing(rtimer):
            for i in range(min(2, n_chunks)):
                render_rays(chunk(i + 1), model, featmaps, projector, samples, inv_uniform=True, N_importance=importance,
                            det=True, src_ray_batch=src_ray_batch)
        barrier()
    return {'rays_per_s': world * n_chunks * 4096 / rdt, 'rays_per_s_per_gpu': n_chunks * 4096 / rdt, 'n_gpus': world,
            'chunks_per_rank': n_chunks, 'chunk_rays': 4096, 'samples': '%d+%d' % (samples, importance),
            'kernels_ms': {k: round(v['mean_ms'], 4) for k, v in rtimer.summary().items()}}


def exchange_microbench(shard, V, C, Hf, Wf, H, W, dev, view_sharded, reps=10):
    """the collectives of one step in isolation, on buffers of the step's sizes -> ms per step spent in them"""
    fm = torch.zeros(V, Hf, Wf, C, device=dev).permute(0, 3, 1, 2)        # channels-last [V,C,Hf,Wf]
    grad = torch.zeros(1, V, H, W, 3, device=dev)
    small = torch.zeros(4, device=dev)
    lo, hi = shard.view_range(V)

    def once():
        shard.dist.all_reduce(small, group=shard.group)
        if view_sharded:
            full = shard.gather_views_nhwc(fm[lo:hi].contiguous(memory_format=torch.channels_last), lo, hi, V)
            shard.scatter_views_nhwc(full, lo, hi)
        shard.all_reduce_grad(grad)
    saved = (shard.collectives, shard.bytes, shard.shard_views)
    shard.shard_views = view_sharded
    once()
    SYNC()
    shard.dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    SYNC()
    dt = (time.perf_counter() - t0) / reps
    shard.collectives, shard.bytes, shard.shard_views = saved
    return 1e3 * dt


def SYNC():
    torch.cuda.synchronize()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: this process becomes a pure parent -- it has not touched the GPU (importing
    torch does not) and never will -- and starts N fresh rank processes of this same script with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, one per GPU (the reference's multi-process scripts initialise themselves the same way, from LOCAL_RANK:
    eval/gnt/eval_adv.py:1211-1214).  Rank 0's stdout (the ONE JSON line) is relayed; the other ranks' stdout goes to stderr.  If a
    rank fails, the others are stopped (exact PIDs) and the parent exits with that rank's code."""
    import socket
    import subprocess
    import tempfile
    if os.environ.get('MASTER_PORT'):          # the caller chose it (a driver that runs several benches side by side)
        port = int(os.environ['MASTER_PORT'])
    else:
        # bind-then-close leaves a window in which another process could take the port; rank 0's rendezvous then fails loudly (address in
        # use) and the parent exits non-zero -- a retry is the caller's call, never a silent second attempt with other ranks half started
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
    base = dict(os.environ)
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC only on this pool (RCCL needs it)
    base.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
    procs = []
    out0 = tempfile.TemporaryFile(mode='w+')
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else sys.stderr))
    rc = 0
    alive = list(procs)
    deadline = None           # set when a rank failed: the survivors get 10 s to honour SIGTERM, then SIGKILL (exact PIDs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in alive:         # a dead rank leaves the others waiting in a collective: stop them, by PID
                    q.terminate()
                deadline = time.monotonic() + 10.0
        if deadline is not None and alive and time.monotonic() > deadline:
            for q in alive:             # stuck inside a collective / kernel that ignores SIGTERM
                q.kill()
            deadline = time.monotonic() + 10.0
    out0.seek(0)
    for line in out0:            # the JSON line to stdout; anything else a library printed on rank 0's stdout (gloo does) to stderr
        (sys.stdout if line.lstrip().startswith('{') else sys.stderr).write(line)
    sys.stdout.flush()
    return rc


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    if a.config == 'c4':
        a.model = 'gnt'
    if a.config == 'c5':          # config 5 defaults unless the user overrode the sizes
        if (a.height, a.width, a.views, a.samples, a.importance) == (756, 1008, 4, 64, 64):
            a.height, a.width, a.views, a.samples, a.importance = 512, 512, 8, 128, 128
        a.precision = a.precision or 'bf16'
        a.cpu_iters = 0
    if a.model == 'gnt':          # config 4 defaults unless the user overrode the sizes
        if (a.height, a.width, a.views) == (756, 1008, 4):
            a.height, a.width, a.views = 800, 800, 10
        a.importance, a.cpu_iters = 0, 0
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if os.environ.get('NERFOOL_BENCH_FAIL_RANK') == str(rank):        # test hook of the launcher (tests/test_distributed_gloo.py)
        sys.exit(3)
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert world == a.gpus, '--gpus %d but WORLD_SIZE %d (launch with torch.distributed.run for N > 1)' % (a.gpus, world)
    standin = a.device == 'cpu-standin'
    if standin:
        a.extras, a.cpu_iters = 0, 0
        # functional check without a GPU (tests/test_distributed_gloo.py): the CPU stand-in build of the kernel sources, bound from
        # the test tree; the product package itself has no CPU path
        global SYNC
        SYNC = lambda: None          # noqa: E731
        sys.path.insert(0, os.path.join(ROOT, 'tests', 'host_harness'))
        import standin as _standin
        _standin.install_emulated()
        from nerfool_amd.ibrnet import feature_network as _fn, mlp_network as _mn
        _mn.KERNEL_PATH, _fn.CNN_PATH = 'generic', 'torch'        # the matrix-core kernels / fused executor emulate ~30x slower
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // max(world, 1)))
        dev = torch.device('cpu')
    else:
        # one rank per GPU; the modulo only matters for the 1-GPU debugging set-up below (several ranks on one device)
        local_dev = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_dev)
        dev = torch.device('cuda', local_dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # 'nccl' is RCCL.  NERFOOL_DIST_BACKEND=gloo lets several ranks share ONE GPU to exercise the sharded path on a
        # single-GPU box (RCCL refuses duplicate devices); it is a functional check, never a measurement.
        backend = 'gloo' if standin else os.environ.get('NERFOOL_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    # the library is built by ONE rank (a no-op when it is fresh, as on the GPU box where the built .so travels in the tree)
    import __graft_entry__ as entry
    if rank == 0 and not standin and not os.path.exists(entry.LIB):
        entry.build()
    if world > 1:
        torch.distributed.barrier()
    args, data, model, sampler, src_ray_batch, projector, EA = build_problem(a, dev)
    from nerfool_amd import prof
    from nerfool_amd.ibrnet import feature_network
    feature_network.WINO_OPERANDS = a.conv_operands

    graph_ok = [True]       # N > 1: cleared when the segmented capture of a sharded step fails on this node (all ranks agree, below)
    graph_note = [None]

    def make_attack(cnn_shard, scaling, n_rand=None):
        """a fresh PGDAttack (own delta / moments) in the given multi-GPU form; n_rand as on the command line: rays per rank
        (weak) or per step in total (strong)"""
        shard = None
        if world > 1:
            shard = EA.RayShard(shard_views=cnn_shard == 'view', split_n_rand=scaling == 'strong')
        return EA.PGDAttack(make_args(a, a.n_rand if n_rand is None else n_rand), model, projector, src_ray_batch, shard=shard,
                            graph=None if graph_ok[0] else False)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        SYNC()

    def max_over_ranks(seconds):
        if world > 1:
            t = torch.tensor([seconds], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            return float(t)
        return seconds

    def host_issue(atk):
        """host time to ENQUEUE a step (no synchronisation inside the bracket; 3 steps stay below the queue depth): the step is
        launch-bound when this approaches ms_per_step.  Sharded steps: segments + collectives as the step issues them."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            atk.step(data)
        ms = 1e3 * (time.perf_counter() - t0) / 3
        barrier()
        return ms

    attack = make_attack(a.cnn_shard, a.scaling)
    if attack.shard is not None:
        # (prime here so that the collective counters below cover warm-up + timed steps only.)  Sharded steps replay as hipGraph
        # segments between their collectives; that path has run on a one-rank RCCL group and with two gloo ranks on one GPU, never on
        # several GPUs -- if the capture raises here (on every rank alike: same code, same hardware), all ranks agree to run the bench
        # with eager steps instead of losing the scaling measurement, and the line says so
        ok = 1
        try:
            prime(attack, data)
        except Exception as e:          # noqa: BLE001
            ok, graph_note[0] = 0, 'segmented graph capture failed, steps enqueued launch by launch: %r' % (e,)
            print('[bench rank %d] %s' % (rank, graph_note[0]), file=sys.stderr, flush=True)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        if int(flag) == 0:
            graph_ok[0] = False
            graph_note[0] = graph_note[0] or 'segmented graph capture failed on another rank, steps enqueued launch by launch'
            attack = make_attack(a.cnn_shard, a.scaling)
        c0, b0 = attack.shard.collectives, attack.shard.bytes
    # HIP events only around the kernels the roofline table prices (~10 of the ~250 launches of a step): bracketing every
    # launch costs ~1.5 ms of host time per step, which is visible now that the step is close to launch-bound
    timer = None if standin else prof.KernelTimer(only=ROOFLINE_KERNELS)
    elapsed = max_over_ranks(time_steps(attack, data, a.steps, a.warmup, barrier, timer, a.event_every))
    timed_steps = len(range(0, a.steps, a.event_every)) if (a.event_every > 0 and timer is not None) else 0
    kernels = {} if timer is None else timer.summary()
    final_loss = float(attack.last_loss)
    rays_per_step = a.n_rand * (world if a.scaling == 'weak' else 1)
    collectives_per_step = payload_per_step = None
    if attack.shard is not None:
        collectives_per_step = (attack.shard.collectives - c0) / float(a.steps + a.warmup)
        payload_per_step = (attack.shard.bytes - b0) / float(a.steps + a.warmup)

    hand_written_ms = None
    multi = None
    host_issue_ms = None
    if not standin:
        host_issue_ms = host_issue(attack)
    if a.extras:
        # every hand-written launch, timed over two extra steps OUTSIDE the timed region (extra.hand_written_kernel_ms_per_step)
        all_timer = prof.KernelTimer()
        with prof.timing(all_timer):
            for _ in range(2):
                attack.step(data)
        hand_written_ms = sum(k['total_ms'] for k in all_timer.summary().values()) / 2
        barrier()
    if a.extras and world > 1:
        # ---- the other multi-GPU forms, so that ONE driver run per N yields weak and strong scaling of both CNN placements
        Hf, Wf = model.feature_net.describe_output(a.height, a.width)[3:5]
        C = sum(model.feature_net.describe_output(a.height, a.width)[0])
        multi = {'headline': {'cnn_shard': a.cnn_shard, 'scaling': a.scaling}, 'forms': {}}
        for cnn_shard, scaling in (('view', 'weak'), ('view', 'strong'), ('replicated', 'weak'), ('replicated', 'strong')):
            if (cnn_shard, scaling) == (a.cnn_shard, a.scaling):
                ms, rays = 1e3 * elapsed / a.steps, rays_per_step
                coll, payload = collectives_per_step, payload_per_step
                issue, segs = host_issue_ms, [len(v[0].graphs) for v in attack._graphs.values()]
            else:
                other = make_attack(cnn_shard, scaling)
                prime(other, data)
                c1, b1 = other.shard.collectives, other.shard.bytes
                n = max(5, a.steps // 2)
                ms = 1e3 * max_over_ranks(time_steps(other, data, n, 2, barrier)) / n
                rays = a.n_rand * (world if scaling == 'weak' else 1)
                coll, payload = (other.shard.collectives - c1) / float(n + 2), (other.shard.bytes - b1) / float(n + 2)
                issue, segs = host_issue(other), [len(v[0].graphs) for v in other._graphs.values()]
                del other
            multi['forms']['%s/%s' % (cnn_shard, scaling)] = {
                'ms_per_step': round(ms, 4), 'rays_per_step_all_ranks': rays, 'rays_per_s': rays / (ms * 1e-3),
                'collectives_per_step': coll, 'collective_payload_bytes_per_step': payload,
                # the step is replayed as hipGraph segments split at its collectives (eval_adv._SegmentedCapture)
                'host_issue_ms_per_step': round(issue, 4), 'graph_segments_per_step': segs[0] if segs else None}
        for cnn_shard in ('view', 'replicated'):
            multi['exchange_ms_per_step_isolated_' + cnn_shard] = round(max_over_ranks(
                exchange_microbench(attack.shard, a.views, C, Hf, Wf, a.height, a.width, dev, cnn_shard == 'view') * 1e-3) * 1e3, 4)

    # ---- legs outside the timed region of the headline value.  The render legs run on EVERY rank (ray-range sharding).
    render = None
    extra_legs = {}
    featmaps = None
    par = (rank, world, barrier, max_over_ranks)
    if a.extras and (a.render_chunks > 0 or a.model == 'ibrnet'):
        with torch.no_grad():       # the same delta on every rank (replicated state of the attack)
            featmaps = model.feature_net((src_ray_batch['src_rgbs'] + attack.delta).squeeze(0).permute(0, 3, 1, 2))
    if a.extras and a.render_chunks > 0:
        render = render_leg(model, projector, sampler, src_ray_batch, featmaps, a.render_chunks, a.samples, a.importance,
                            a.model == 'gnt', prof, par)
    if a.extras and a.attack_iters > 0 and world == 1 and a.model == 'ibrnet' and a.config == 'c2':
        # BASELINE.json's second figure, MEASURED: wall-clock of a whole view-specific attack of `attack_iters` (1000) iterations on a
        # fresh perturbation -- PGDAttack.run_view_specific (eval_adv.py:796-843), pixel picks from the RandomState(234) stream
        # included, between barrier + synchronize brackets
        run1k = make_attack(a.cnn_shard, a.scaling)
        prime(run1k, data)
        barrier()
        t0 = time.perf_counter()
        run1k.run_view_specific(data, n_iters=a.attack_iters)
        barrier()
        dt = time.perf_counter() - t0
        extra_legs['attack_%d_iters_wall_s' % a.attack_iters] = round(dt, 4)
        extra_legs['attack_%d_iters' % a.attack_iters] = {'wall_s': round(dt, 4), 'ms_per_iter': round(1e3 * dt / a.attack_iters, 4),
                                                          'rays_per_s': a.n_rand * a.attack_iters / dt, 'final_loss': float(run1k.last_loss),
                                                          'measured': True}
        del run1k
    if a.extras and world == 1 and a.model == 'ibrnet' and a.config == 'c2':
        # N_rand = 4096 attack step (SURVEY 8d asks for 512 and 4096)
        big = make_attack(a.cnn_shard, a.scaling, n_rand=4096)
        n = max(5, a.steps // 2)
        ms = 1e3 * time_steps(big, data, n, 2, barrier) / n
        extra_legs['attack_n_rand_4096'] = {'ms_per_step': round(ms, 4), 'rays_per_s': 4096 / (ms * 1e-3)}
        del big
    if a.extras and world == 1 and a.model == 'ibrnet' and a.config == 'c2' and a.conv_operands == 'bf16x3':
        # the same step with the backward-data convolutions on THREE operand parts like the forward (the timed headline runs them on two:
        # config.conv_operands_bwd)
        saved_bwd, feature_network.WINO_BWD_OPERANDS = feature_network.WINO_BWD_OPERANDS, 'bf16x3'
        try:
            x3 = make_attack(a.cnn_shard, a.scaling)
            n = max(5, a.steps // 2)
            extra_legs['step_ms_all_bf16x3'] = round(1e3 * time_steps(x3, data, n, 2, barrier) / n, 4)
            del x3
        finally:
            feature_network.WINO_BWD_OPERANDS = saved_bwd
    if a.extras and a.model == 'gnt':
        # BASELINE config 4's UNIVERSAL loop runs the network in training mode (Dropout(0.1) live: eval/gnt/eval_adv.py:739-878 before
        # switch_to_eval at :959): the same step with the training-mode matrix-core kernels, replayed as a hipGraph with a fresh seed per step
        model.switch_to_train()
        try:
            tr = make_attack(a.cnn_shard, a.scaling)
            n = max(5, a.steps // 2)
            ms_tr = 1e3 * max_over_ranks(time_steps(tr, data, n, 2, barrier)) / n
            t_timer = prof.KernelTimer(only=('nf_gnt_fwd_mfma', 'nf_gnt_bwd_mfma', 'nf_gnt_fwd', 'nf_gnt_bwd'))
            with prof.timing(t_timer):
                tr.step(data)
            extra_legs['training_mode_step'] = {
                'ms_per_step': round(ms_tr, 4), 'vs_eval_mode_step': round(ms_tr / (1e3 * elapsed / a.steps), 4),
                'graph_replays': tr.graph_replays, 'kernels_ms': {k: round(v['mean_ms'], 4) for k, v in t_timer.summary().items()},
                'note': 'Dropout(0.1) at the eight sites of every layer, masks from the counter-based generator (nf_gnt.h), seeds read from '
                        'device words refreshed before each replay'}
            del tr
        finally:
            model.switch_to_eval()
    if a.extras and a.universal_views > 0 and a.model == 'ibrnet' and a.config == 'c2':
        extra_legs['universal'] = universal_leg(a, EA, make_attack, data, dev, barrier, max_over_ranks, world)
    if a.extras and a.model == 'ibrnet' and a.config == 'c2':
        # the whole 756x1008 image through render_single_image: 187 chunks of 4096 rays, outputs moved to the host (D2H) and
        # reshaped as the reference does (render_image.py:52-102).  N > 1: the chunks are dealt to the ranks as contiguous blocks
        # and one gather of the packed per-ray records assembles the image on rank 0 (strong scaling of one image).
        from nerfool_amd.ibrnet.render_image import render_single_image
        rays = sampler.get_all()
        img_shard = EA.RayShard(shard_views=False) if world > 1 else None
        t_img = []
        ret = None
        for _ in range(3):
            ret = None          # the page-locked host tensors of the previous image go back to torch's host allocator
            barrier()
            t0 = time.perf_counter()
            ret = render_single_image(ray_sampler=sampler, ray_batch=rays, model=model, projector=projector, chunk_size=4096,
                                      N_samples=a.samples, inv_uniform=True, N_importance=a.importance, det=True, white_bkgd=False,
                                      featmaps=featmaps, src_ray_batch=src_ray_batch, shard=img_shard)
            barrier()
            t_img.append(max_over_ranks(time.perf_counter() - t0))
        n_rays = a.height * a.width
        extra_legs['render_single_image'] = {
            'image': '%dx%d' % (a.height, a.width), 'chunks': -(-n_rays // 4096), 'samples': '%d+%d' % (a.samples, a.importance),
            'n_gpus': world, 'seconds': round(min(t_img), 4), 'rays_per_s': n_rays / min(t_img),
            'includes': 'chunk loop, D2H of all six output fields per level (second stream, page-locked host tensors), reshape' if world == 1 else
                        'chunk loop on all ranks, ONE gather of the packed per-ray records to rank 0 (%.2f GB), D2H there, reshape'
                        % (img_shard.bytes / 3 / 1e9),
            'rgb_shape': None if ret is None else list(ret['outputs_fine']['rgb'].shape)}
        del ret
        # north-star render case: 800x800 scene, 64 samples per ray -- coarse only and 64+64
        a8 = argparse.Namespace(**vars(a))
        a8.height, a8.width = 800, 800
        args8, data8, model8, sampler8, src8, projector8, _ = build_problem(a8, dev)
        with torch.no_grad():
            fm8 = model8.feature_net(src8['src_rgbs'].squeeze(0).permute(0, 3, 1, 2))
        for imp, tag in ((0, 'render_800x800_64'), (64, 'render_800x800_64+64')):
            # 32 chunks per rank (a 800x800 image is 156): a short loop mostly measures its own start-up (the host needs ~0.3 ms to
            # enqueue a chunk's first launches while the GPU idles) -- 8 chunks read 12 % low
            r = render_leg(model8, projector8, sampler8, src8, fm8, 32, 64, imp, False, prof, par)
            fl = ibrnet_flops(1, 64, a.views) + (ibrnet_flops(1, 64 + imp, a.views) if imp else 0)
            # priced like extra.kernels: the row network of this leg (k_ibr_sol_fwd) issues its products on the bf16 matrix pipe with
            # every fp32 operand in three parts -- executed bf16 FLOPs <= 6 x the algorithmic FLOPs (upper bound: the view-invariant
            # part of base_fc.0 is multiplied once per sample, the per-ray attention runs in fp32), against the dense bf16 peak
            tf = r['rays_per_s_per_gpu'] * fl / 1e12
            r['mfma_frac_of_peak'] = round(6.0 * tf / PEAK_BF16_TFLOPS, 4)
            r['mfma_peak'] = 'dense bf16 matrix peak %.0f TFLOP/s; achieved = 6 x algorithmic FLOPs (upper bound on the executed bf16 products), whole leg incl. per-ray kernels' % PEAK_BF16_TFLOPS
            r['product_frac_of_fp32_peak'] = round(tf / PEAK_F32_TFLOPS, 4)
            extra_legs[tag] = r
        del model8, fm8, sampler8, src8, data8

    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    # ---- roofline of the dominant hand-written kernel of the timed region
    V, Sc, R = a.views, a.samples, a.n_rand
    table = {}
    # the Winograd entry point runs two kernels since round 5 -- k_wino3x3_bf<.., 3> (forward, three operand parts) and
    # k_wino3x3_bf<.., 2> (backward-data, two parts): priced as two rows, like rocprofv3 lists them
    split_kernels = {}
    for name, k in kernels.items():
        if name == 'nf_conv3x3_wino' and any(m.get('n_split') == 2 for m in k['meta']) and any(m.get('n_split') != 2 for m in k['meta']):
            for tag, pick in (('nf_conv3x3_wino', lambda m: m.get('n_split') != 2), ('nf_conv3x3_wino_bwd', lambda m: m.get('n_split') == 2)):
                ms = [t for t, m in zip(k['ms'], k['meta']) if pick(m)]
                split_kernels[tag] = {'ms': ms, 'meta': [m for m in k['meta'] if pick(m)], 'launches': len(ms),
                                      'mean_ms': sum(ms) / len(ms), 'total_ms': sum(ms)}
        else:
            split_kernels[name] = k
    wino_stats = {}
    for name, k in split_kernels.items():
        per_launch = []
        wino_direct, wino_bytes, wino_mult = [], [], []
        for ms, meta in zip(k['ms'], k['meta']):
            if name in ('nf_ibrnet_fwd', 'nf_ibrnet_bwd', 'nf_ibrnet_fwd_mfma', 'nf_ibrnet_bwd_mfma'):
                per_launch.append(('mfma', ibrnet_flops(meta['R'], meta['S'], meta['V']) / (ms * 1e-3) / 1e12))
            elif name in ('nf_ibrnet_fwd_mfma_bf16', 'nf_ibrnet_bwd_mfma_bf16'):
                per_launch.append(('mfma_bf16', ibrnet_flops(meta['R'], meta['S'], meta['V']) / (ms * 1e-3) / 1e12))
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
            elif name in ('nf_conv3x3_wino', 'nf_conv3x3_wino_bwd'):
                # Winograd: multiply-adds per output tile and (c_in, c_out) pair that the kernel puts on the matrix cores
                # (F(2x2,3x3): 16 per 2x2 tile; F(4x4,3x3): 36 per 4x4 tile) -- the direct form of the same convolution needs
                # 9 per output: extra.kernels reports that rate too
                m = meta.get('m', 2)
                tiles = meta['n_img'] * ((meta['Ho'] + m - 1) // m) * ((meta['Wo'] + m - 1) // m)
                # bf16 products executed per Winograd-domain product: 6 with three operand parts (forward), 3 with two (backward-data
                # since round 5), 1 with plain bf16 operands; n_split 0 = the fp32 matrix instruction
                wino_mult.append({0: 1.0, 1: 1.0, 2: 3.0, 3: 6.0}[meta.get('n_split', 0)])
                per_launch.append(('mfma', 2.0 * (m + 2) ** 2 * meta['c_in'] * meta['c_out'] * tiles / (ms * 1e-3) / 1e12))
                wino_direct.append(2.0 * 9 * meta['c_in'] * meta['c_out'] * meta['n_img'] * meta['Ho'] * meta['Wo'] / (ms * 1e-3) / 1e12)
                wino_bytes.append(4.0 * (meta['n_img'] * (meta['c_in'] * meta['Hi'] * meta['Wi'] + meta['c_out'] * meta['Ho'] * meta['Wo'])
                                         + (m + 2) ** 2 * meta['c_in'] * meta['c_out']))
        if per_launch:
            bound = per_launch[0][0]
            ach = float(np.mean([x[1] for x in per_launch]))
            peak = {'mfma': PEAK_F32_TFLOPS, 'mfma_bf16': PEAK_BF16_TFLOPS, 'hbm': PEAK_HBM_GBS}[bound]
            entry = {'launches': k['launches'], 'mean_ms': round(k['mean_ms'], 4), 'total_ms': round(k['total_ms'], 3)}
            x3 = X3_KERNELS.get(name)
            is_wino = name in ('nf_conv3x3_wino', 'nf_conv3x3_wino_bwd')
            if is_wino:
                wino_stats[name] = (float(np.mean(wino_direct)), int(np.mean(wino_bytes)))
            if x3 is not None and (not is_wino or a.conv_operands == 'bf16x3'):
                # a kernel whose products ISSUE ON THE BF16 MATRIX PIPE -- every product as six (three operand parts) or three (two
                # parts: the backward-data convolutions) bf16 products -- is priced against THAT pipe's dense peak: achieved =
                # executed bf16 TFLOP/s, launch by launch
                if is_wino:
                    ex = float(np.mean([x[1] * mlt for x, mlt in zip(per_launch, wino_mult)]))
                else:
                    ex = 6.0 * ach
                entry.update(bound='mfma', achieved=round(ex, 3), peak=PEAK_BF16_TFLOPS, unit='TFLOP/s',
                             frac=round(ex / PEAK_BF16_TFLOPS, 5), pipe='bf16 matrix cores, operands split into bf16 parts',
                             product_tflops=round(ach, 4), product_frac_of_fp32_peak=round(ach / PEAK_F32_TFLOPS, 5), note=x3)
            else:
                if bound == 'mfma_bf16':
                    bound = 'mfma'          # priced against the dense bf16 matrix peak
                entry.update(bound=bound, achieved=round(ach, 4), peak=peak, unit='TFLOP/s' if bound == 'mfma' else 'GB/s',
                             frac=round(ach / peak, 5))
            table[name] = entry
    for wn, (direct, nbytes) in wino_stats.items():
        table[wn]['direct_form_equivalent_tflops'] = round(direct, 2)
        table[wn]['operands'] = a.conv_operands if wn == 'nf_conv3x3_wino' else 'bf16x2 (two operand parts: three products)'
        table[wn]['rocprof_kernel'] = 'k_wino3x3_bf<*, 3>' if wn == 'nf_conv3x3_wino' and a.conv_operands == 'bf16x3' else (
            'k_wino3x3_bf<*, 2>' if wn == 'nf_conv3x3_wino_bwd' else 'k_wino3x3<*>')
    if 'nf_ibrnet_bwd_mfma' in table and 'nf_project_gather_bwd' not in table:
        table['nf_ibrnet_bwd_mfma']['includes'] = ('the scatter of d rgb_feat into the feature-map gradient (float atomics, formerly '
                                                   'nf_project_gather_bwd: 0.11 ms per launch) -- not counted in the FLOPs')
    dominant = max(table, key=lambda n: table[n]['total_ms']) if table else None
    roofline = None
    if dominant:
        d = table[dominant]
        t = pmc_traffic(dominant, a)
        if t is not None and t.get('algorithmic_bytes_per_launch') is None and dominant in wino_stats:
            t['algorithmic_bytes_per_launch'] = wino_stats[dominant][1]     # input + output + transformed weights
        # traffic: HBM bytes per launch of the dominant kernel from the committed PMC passes (null when they do not cover this workload)
        roofline = {'kernel': dominant, 'bound': d['bound'], 'achieved': d['achieved'], 'peak': d['peak'], 'unit': d['unit'],
                    'frac': d['frac'], 'traffic': None if t is None else t['hbm_bytes_per_launch'], 'traffic_unit': 'B/launch',
                    'traffic_algorithmic': None if t is None else t.get('algorithmic_bytes_per_launch'),
                    'traffic_source': None if t is None else t['source']}
    # whole step against the fp32 matrix peak: algorithmic FLOPs of the step (CNN forward + backward-data on the views this rank
    # computes, render forward + input-gradient backward on its rays) / step time
    ms_step = 1e3 * elapsed / a.steps
    whole = None
    if a.model == 'ibrnet':
        views_here = V if (world == 1 or a.cnn_shard == 'replicated') else -(-V // world)
        rays_here = rays_per_step / world
        fl = 2.0 * views_here * resunet_flops(a.height, a.width) + rays_here * 2.0 * (
            ibrnet_flops(1, Sc, V) + (ibrnet_flops(1, Sc + a.importance, V) if a.importance else 0))
        whole = {'algorithmic_tflop_per_step_per_gpu': round(fl / 1e12, 4), 'achieved_tflops': round(fl / (ms_step * 1e-3) / 1e12, 3),
                 'frac_of_fp32_mfma_peak': round(fl / (ms_step * 1e-3) / 1e12 / PEAK_F32_TFLOPS, 4),
                 'note': 'direct-form FLOPs; the Winograd convolutions execute fewer multiplications than that'}

    par = 'single GPU' if world == 1 else ('ray-sharded dp%d, %s scaling, feature CNN %s' % (
        world, a.scaling, 'sharded by source view' if a.cnn_shard == 'view' else 'replicated (one all-reduce of d delta)'))
    if a.model == 'ibrnet':
        workload = ('BASELINE config %s: IBRNet view-specific attack, %s-shaped synthetic scene %dx%d, %d source views, %d+%d samples/ray, '
                    'N_rand=%d rays per %s per step, Adam lr 1e-3, eps 8/255%s'
                    % ((('5', 'DeepVoxels') if a.config == 'c5' else ('2', 'LLFF-fern')) + (
                        a.height, a.width, V, Sc, a.importance, a.n_rand, 'rank' if a.scaling == 'weak' else 'step (all ranks)',
                        ', IBRNet row network on bf16 matrix-core operands' if a.precision == 'bf16' else '')))
    else:
        workload = ('BASELINE config 4: GNT depth %d view-specific attack, synthetic scene %dx%d, %d source views, %d samples/ray, '
                    'N_rand=%d rays per rank per step' % (a.depth, a.height, a.width, V, Sc, a.n_rand))
    out = {
        'metric': 'rays/s through the %s PGD attack step (render fwd+bwd + CNN fwd+bwd + delta update), %d src views'
                  % ('IBRNet' if a.model == 'ibrnet' else 'GNT', a.views),
        'value': rays_per_step * a.steps / elapsed,
        'unit': 'rays/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
        'ms_per_step': ms_step,
        'higher_is_better': True, 'scaling': a.scaling, 'vs_baseline': None,
        # the arithmetic type of the path: c5 runs the IBRNet row network on bf16 matrix-core operands (fp32 accumulate; CNN, per-ray
        # part, compositing and update in fp32)
        'dtype': 'bf16' if (a.precision == 'bf16') else 'f32', 'data': 'synthetic',
        'device': 'MI355X (gfx950)' if not standin else 'cpu-standin: FUNCTIONAL check of the launcher / sharding, not a measurement',
        'config': {'workload': workload, 'conv_operands': a.conv_operands,
                   # operand form of the 27 backward-data Winograd launches of a step (round 5: two bf16 parts per operand, three products;
                   # the all-bf16x3 step is timed beside it: extra.step_ms_all_bf16x3)
                   'conv_operands_bwd': (feature_network.WINO_BWD_OPERANDS if a.conv_operands == 'bf16x3' else a.conv_operands),
                   'rays_per_step_all_ranks': rays_per_step, 'parallelism': par,
                   'collectives_per_step': collectives_per_step, 'collective_payload_bytes_per_step': payload_per_step,
                   'step_launch': ('one hipGraph replay per step' if world == 1 else 'hipGraph segments between the collectives')
                                  if (graph_ok[0] and not standin) else (graph_note[0] or 'launch by launch')},
        'roofline': roofline,
        'host_issue_ms_per_step': None if host_issue_ms is None else round(host_issue_ms, 4),
        'roofline_sampling': {'steps_with_hip_events': timed_steps, 'of_timed_steps': a.steps, 'every': a.event_every,
                              'note': 'per-launch durations from HIP events on the launch stream, live inside the timed region on every '
                                      'N-th step (an event pair costs the GPU ~11 us around the launch it brackets)'},
        'cpu_baseline': None,
        'extra': {'attack_s_per_1000_iters_from_ms_per_step': ms_step, 'final_loss': final_loss,
                  'hbm_peak_allocated_gb': None if standin else round(torch.cuda.max_memory_allocated(dev) / 1e9, 3), 'kernels': table, 'whole_step': whole,
                  'hand_written_kernel_ms_per_step': None if hand_written_ms is None else round(hand_written_ms, 4),
                  'conv3x3_choice': {'%s %s' % (k[0], 'x'.join(map(str, k[1:]))): v for k, v in feature_network._CONV_CHOICE.items()},
                  'render': render, 'multi_gpu': multi, **extra_legs},
    }
    if world == 1 and a.cpu_iters > 0 and a.extras:
        out['cpu_baseline'] = cpu_baseline(a, args, data, model)
    print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
