"""CPU, world_size 2 / 3 / 8, gloo: the sharded PGD step.  Each rank differentiates its slice of the step's rays
through the CPU stand-in build of the kernels; the mask counts are all-reduced before the loss is normalised and
d(delta) is all-reduced once; with view sharding each rank also runs the feature CNN only for its own source views and
the feature maps / their gradients are exchanged.  The result must equal the single-process gradient of the union of the
rays, and all ranks must hold the same delta after the fused update."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, 'tests', 'host_harness')

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, %(harness)r)
import standin
standin.use_library(os.path.join(%(harness)r, 'libnerfool_emu.so'))
from nerfool_amd.ibrnet import mlp_network, feature_network
mlp_network.KERNEL_PATH = 'generic'          # the shape-generic kernels emulate ~30x faster than the MFMA ones; the sharding
feature_network.CNN_PATH = 'torch'           # logic under test is the same
import parity_cases as pc
from nerfool_amd import eval_adv as EA
from nerfool_amd.ibrnet.projection import Projector

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
if world > 1:
    dist.init_process_group('gloo', rank=rank, world_size=world)
torch.set_num_threads(2)
mode = os.environ.get('MODE', 'ibrnet')
draw = os.environ.get('DRAW') == '1'        # the step draws its own pixels: args.N_rand is the GLOBAL batch (split_n_rand)
shard = EA.RayShard(shard_views=os.environ['SHARD_VIEWS'] == '1', split_n_rand=draw) if world > 1 else None
if mode == 'gnt':
    # BASELINE config 4's flavour: GNT renderer on single_net feature maps (one map serves both levels: the view exchange moves it
    # once), unmasked MSE; V = 3 source views over 2 ranks is a ragged 2 + 1 split
    from types import SimpleNamespace
    from nerfool_amd.gnt import eval_adv as GEA
    from nerfool_amd.gnt.model import GNTModel
    from nerfool_amd.ibrnet.sample_ray import RaySamplerSingleImage
    from nerfool_amd.synthetic import make_scene
    torch.manual_seed(0)
    H, W, V, R, S, depth = 48, 64, 3, 24, 8, 2
    args = SimpleNamespace(netwidth=64, trans_depth=depth, single_net=True, ret_alpha=False, coarse_feat_dim=32, fine_feat_dim=32,
                           N_rand=R, N_samples=S, N_importance=0, inv_uniform=True, det=True, white_bkgd=False, epsilon=8, adv_lr=2,
                           use_adam=True, adam_lr=1e-3, lr_step_size=100, lr_gamma=0.5, adv_iters=1, sample_mode='uniform',
                           center_ratio=0.8, ckpt_path=None)
    model = GNTModel(args, device='cpu')
    model.switch_to_eval()
    data = make_scene(H, W, V, seed=21, tilt=0.3)
    sampler = RaySamplerSingleImage(data, 'cpu')
    src = sampler.get_all()
    picks = np.random.RandomState(5).choice(H * W, size=(R,), replace=False)
    delta0 = (torch.rand(data['src_rgbs'].shape, generator=torch.Generator().manual_seed(9)) * 2 - 1) * (8.0 / 255.0)
    atk = GEA.PGDAttack(args, model, Projector('cpu'), src, shard=shard, delta=delta0.clone().requires_grad_(True))
else:
    g, args, model, data, sampler, dims = pc._attack_setup('cpu')
    if mode == 'bf16':
        # BASELINE config 5's precision: the row network on bf16 matrix-core operands -- needs the matrix-core kernels
        mlp_network.KERNEL_PATH = 'auto'
        for net in (model.net_coarse, model.net_fine):
            net.precision = 'bf16'
    src = sampler.get_all()
    picks = g.np('adam/selected_inds')[0]
    atk = EA.PGDAttack(args, model, Projector('cpu'), src, shard=shard, delta=g.t('in/delta0').clone().requires_grad_(True))
if mode == 'universal':
    # the universal loop (eval_adv.py:634-740) over two target views, adv_iters + 1 = 3 steps, pixels drawn by the loop itself
    from fixtures import second_target_view
    from nerfool_amd.ibrnet import sample_ray
    sample_ray.rng.seed(234)
    losses, inner = [], atk.step
    atk.step = lambda d, select_inds=None, lookahead=True: losses.append(float(inner(d, select_inds, lookahead))) or losses[-1]
    atk.run_universal([data, second_target_view(data)], n_iters=2)
    assert atk.iters == 3 and len(losses) == 3
    grad = torch.tensor(losses)
elif draw:
    from nerfool_amd.ibrnet import sample_ray
    sample_ray.rng.seed(234)
    grad = atk.gradient(data, lookahead=False).clone()
else:
    mine = picks if world == 1 else picks[rank::world]
    grad = atk.gradient(data, select_inds=mine).clone()
if mode != 'universal':
    atk.apply(grad)
tag = os.environ['SHARD_VIEWS'] + ('d' if draw else '') + ('' if mode == 'ibrnet' else mode)
np.savez(os.path.join(%(out)r, 'rank%%d_of_%%d_%%s.npz' %% (rank, world, tag)), grad=grad.numpy(), delta=atk.delta.detach().numpy(),
         loss=float(atk.last_loss), collectives=0 if shard is None else shard.collectives)
if world > 1:
    dist.destroy_process_group()
'''


def _build_and_script(tmp_path):
    if not os.path.exists('/opt/rocm/lib/llvm/bin/clang++'):
        pytest.skip('clang++ of the ROCm toolchain is needed to build the CPU stand-in')
    subprocess.run([os.path.join(HARNESS, 'build.sh')], check=True, capture_output=True)
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT, harness=HARNESS, out=str(tmp_path)))
    return script


def _run_world(script, env, world, shard_views, draw=False, mode='ibrnet'):
    procs = [subprocess.Popen([sys.executable, str(script)],
                              env=dict(env, RANK=str(r), WORLD_SIZE=str(world), SHARD_VIEWS='1' if shard_views else '0',
                                       DRAW='1' if draw else '0', MODE=mode))
             for r in range(world)]
    assert all(p.wait() == 0 for p in procs)


@pytest.mark.timeout(2400)
def test_sharded_step_equals_single_process(tmp_path):
    script = _build_and_script(tmp_path)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29531', OMP_NUM_THREADS='2')
    _run_world(script, env, 1, False)
    ref = np.load(tmp_path / 'rank0_of_1_0.npz')
    scale = np.abs(ref['grad']).max()
    # (world, view sharding): rays only = the north-star form (16-byte all-reduce + ONE all-reduce of d delta); rays + CNN by
    # view with all-gather / reduce-scatter / all-gather (2 + 2 views); ragged view blocks (2 + 1 + 1) and 8 ranks over 4
    # views (ranks 4-7 own no view) with one all-reduce per exchange
    port = 29532
    for world, shard_views in ((2, False), (2, True), (3, True), (8, True)):
        port += 1
        _run_world(script, dict(env, MASTER_PORT=str(port), OMP_NUM_THREADS='1' if world > 4 else '2'), world, shard_views)
        ranks = [np.load(tmp_path / ('rank%d_of_%d_%d.npz' % (r, world, shard_views))) for r in range(world)]
        # all-reduced gradient == gradient of the union of the rays (different summation order only)
        assert np.abs(ranks[0]['grad'] - ref['grad']).max() <= 2e-4 * scale, (world, shard_views)
        for r in ranks:
            assert np.array_equal(ranks[0]['grad'], r['grad']), 'ranks must hold the identical all-reduced gradient'
            assert np.array_equal(ranks[0]['delta'], r['delta']), 'delta must stay replicated'
            # every rank reports the GLOBAL loss (numerators and denominators travel in the same 16-byte all-reduce)
            assert abs(float(r['loss']) - float(ref['loss'])) <= 1e-5 * abs(float(ref['loss']))
            assert int(r['collectives']) == (4 if shard_views else 2), 'collectives per step'
        assert np.abs(ranks[0]['delta'] - ref['delta']).mean() <= 1e-6
    # strong-scaling semantics: args.N_rand is the global batch, the step draws the pixels itself -- exactly the single-process
    # draw from RandomState(234), split over the ranks
    _run_world(script, env, 1, False, draw=True)
    ref_d = np.load(tmp_path / 'rank0_of_1_0d.npz')
    assert np.abs(ref_d['grad'] - ref['grad']).max() <= 1e-5 * scale      # pick 0 of the stream == the recorded pick (atomics: order noise)
    _run_world(script, dict(env, MASTER_PORT=str(port + 1)), 2, True, draw=True)
    got = np.load(tmp_path / 'rank1_of_2_1d.npz')
    assert np.abs(got['grad'] - ref_d['grad']).max() <= 2e-4 * scale
    assert abs(float(got['loss']) - float(ref_d['loss'])) <= 1e-5 * abs(float(ref_d['loss']))


@pytest.mark.timeout(1200)
@pytest.mark.parametrize('mode', ['gnt', 'bf16'])
def test_sharded_step_of_the_8_gpu_configs(tmp_path, mode):
    """The two BASELINE configurations that name 8 GPUs besides config 3: the GNT attack step (config 4; precedent
    eval/gnt/eval_adv.py:1211-1214) and the IBRNet step with the bf16 row network (config 5), world 2, both CNN placements --
    replicated (2 collectives) and sharded by source view (4 collectives; GNT: V = 3 over 2 ranks, the single_net map) -- each
    equal to its single-process result on the same rays."""
    script = _build_and_script(tmp_path)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29581' if mode == 'gnt' else '29591', OMP_NUM_THREADS='2')
    _run_world(script, env, 1, False, mode=mode)
    ref = np.load(tmp_path / ('rank0_of_1_0%s.npz' % mode))
    scale = np.abs(ref['grad']).max()
    assert scale > 0
    for i, shard_views in enumerate((False, True)):
        _run_world(script, dict(env, MASTER_PORT=str(int(env['MASTER_PORT']) + 1 + i)), 2, shard_views, mode=mode)
        ranks = [np.load(tmp_path / ('rank%d_of_2_%d%s.npz' % (r, shard_views, mode))) for r in range(2)]
        assert np.abs(ranks[0]['grad'] - ref['grad']).max() <= 2e-4 * scale, (mode, shard_views, np.abs(ranks[0]['grad'] - ref['grad']).max() / scale)
        for r in ranks:
            assert np.array_equal(ranks[0]['grad'], r['grad']) and np.array_equal(ranks[0]['delta'], r['delta'])
            assert abs(float(r['loss']) - float(ref['loss'])) <= 1e-5 * abs(float(ref['loss']))
            assert int(r['collectives']) == (4 if shard_views else 2)
        assert np.abs(ranks[0]['delta'] - ref['delta']).mean() <= 1e-6


@pytest.mark.timeout(1200)
def test_sharded_universal_loop_equals_single_process(tmp_path):
    """PGDAttack.run_universal (eval_adv.py:634-740; BASELINE config 3's loop) over two target views x 3 steps on two ranks, strong-scaling
    semantics (the step draws exactly the single-process pixels and splits them), both CNN placements: the perturbation is identical on
    both ranks and equals the single-process trajectory (mean within 1e-2 of eps -- 3e-5 without a flipped ReLU unit --, no entry further than the loop's three Adam steps; first loss 1e-5, later ones 2e-3) -- three consecutive sharded Adam steps, not one."""
    script = _build_and_script(tmp_path)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29561', OMP_NUM_THREADS='2')
    _run_world(script, env, 1, False, draw=True, mode='universal')
    ref = np.load(tmp_path / 'rank0_of_1_0duniversal.npz')
    eps = 8 / 255.
    for i, shard_views in enumerate((False, True)):
        _run_world(script, dict(env, MASTER_PORT=str(29562 + i)), 2, shard_views, draw=True, mode='universal')
        ranks = [np.load(tmp_path / ('rank%d_of_2_%dduniversal.npz' % (r, shard_views))) for r in range(2)]
        assert np.array_equal(ranks[0]['delta'], ranks[1]['delta']), 'delta must stay replicated over the steps of the loop'
        assert np.array_equal(ranks[0]['grad'], ranks[1]['grad']), 'every rank reports the global loss'
        # (losses: the first one to rounding; the third one is taken after two sign-like Adam steps -- see below)
        assert abs(ranks[0]['grad'][0] - ref['grad'][0]) <= 1e-5 * ref['grad'][0], (ranks[0]['grad'], ref['grad'])
        assert np.abs(ranks[0]['grad'] - ref['grad']).max() <= 2e-3 * np.abs(ref['grad']).max(), (ranks[0]['grad'], ref['grad'])
        # Adam's first steps are sign-like (m / sqrt(v) with v from one or two gradients): an entry whose gradient is at rounding level
        # moves by a fraction of lr = 0.032 eps in either direction depending on the summation order -- bounded on the mean, and on the
        # worst entry by one step's length
        dd = np.abs(ranks[0]['delta'] - ref['delta'])
        print('universal world-2 (views sharded: %s): mean |d delta| %.2e eps, max %.2e eps' % (shard_views, dd.mean() / eps, dd.max() / eps))
        # measured over repeated runs: mean 3e-5 .. 6e-5 eps when every ReLU of the CNN decides alike, 2.7e-3 eps when one unit flips
        # (the view-sharded CNN convolves 2 images per rank instead of 4: another CPU code path, DESIGN section 2 on ReLU flips);
        # no entry may be further away than the three steps of the loop can move it
        assert dd.mean() <= 1e-2 * eps and dd.max() <= 3e-3, (dd.mean() / eps, dd.max() / eps)
        assert int(ranks[0]['collectives']) == 3 * (4 if shard_views else 2)
        assert np.abs(ref['delta']).max() > 0.5 * eps          # (the loop moved the perturbation)


VIEW_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
import torch.distributed as dist
from nerfool_amd import eval_adv as EA
from nerfool_amd.ibrnet import feature_network
from nerfool_amd.ibrnet.feature_network import ResUNet
feature_network.CNN_PATH = 'torch'

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
torch.set_num_threads(2)
torch.manual_seed(0)
net = ResUNet(coarse_out_ch=8, fine_out_ch=8).eval()
for p in net.parameters():
    p.requires_grad_(False)
V, H, W = 2, 40, 56
src = torch.rand(1, V, H, W, 3)
delta = (0.03 * torch.randn(1, V, H, W, 3)).requires_grad_(True)
shard = EA.RayShard()
assert [shard.view_range(V)] == [[(0, 1), (1, 2), (2, 2)][rank]]
coarse, fine = shard.view_sharded_featmaps(net, src, delta)
ref_c, ref_f = net((src + delta.detach()).squeeze(0).permute(0, 3, 1, 2))
assert coarse.shape == ref_c.shape and fine.shape == ref_f.shape
tol = 1e-4 * float(ref_c.abs().max())      # batch-1 vs batch-2 convolutions take different CPU code paths
assert float((coarse.detach() - ref_c).abs().max()) <= tol and float((fine.detach() - ref_f).abs().max()) <= tol, (
    float((coarse.detach() - ref_c).abs().max()), tol)
# every rank weights the maps differently (as if it had rendered other rays); d(delta) must be the gradient of the sum
gen = torch.Generator().manual_seed(5)
wts = [(torch.randn(ref_c.shape, generator=gen), torch.randn(ref_f.shape, generator=gen)) for _ in range(world)]
((coarse * wts[rank][0]).sum() + (fine * wts[rank][1]).sum()).backward()
grad = delta.grad.clone()
lo, hi = shard.view_range(V)
assert float(grad[:, :lo].abs().sum()) == 0 and float(grad[:, hi:].abs().sum()) == 0      # only the own views
shard.all_reduce_grad(grad)
d2 = delta.detach().clone().requires_grad_(True)
c2, f2 = net((src + d2).squeeze(0).permute(0, 3, 1, 2))
((c2 * sum(w[0] for w in wts)).sum() + (f2 * sum(w[1] for w in wts)).sum()).backward()
assert float((grad - d2.grad).abs().max()) <= 2e-4 * float(d2.grad.abs().max())
dist.destroy_process_group()
'''


@pytest.mark.timeout(600)
def test_view_sharded_feature_maps_with_idle_rank(tmp_path):
    """world 3 over V = 2 source views: rank 2 owns no view but takes part in both exchanges (torch CNN path, no kernels)."""
    script = tmp_path / 'view_worker.py'
    script.write_text(VIEW_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29547', OMP_NUM_THREADS='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), WORLD_SIZE='3')) for r in range(3)]
    assert all(p.wait() == 0 for p in procs)


RENDER_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
from types import SimpleNamespace
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, %(harness)r)
import standin
standin.use_library(os.path.join(%(harness)r, 'libnerfool_emu.so'))
from nerfool_amd.ibrnet import mlp_network, feature_network
mlp_network.KERNEL_PATH = 'generic'          # the matrix-core kernels emulate ~30x slower; the sharding under test is the same
feature_network.CNN_PATH = 'torch'
import parity_cases as pc
from nerfool_amd import eval_adv as EA
from nerfool_amd.ibrnet.projection import Projector
from nerfool_amd.ibrnet.render_image import render_single_image, chunk_block
from nerfool_amd.ibrnet.sample_ray import RaySamplerSingleImage

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
if world > 1:
    dist.init_process_group('gloo', rank=rank, world_size=world)
torch.set_num_threads(2)
shard = EA.RayShard(shard_views=False) if world > 1 else None
out = {}

# ---- IBRNet flavour: the first 6 rows of the attack fixture's image (384 rays), coarse + fine
g, args, model, data, sampler, dims = pc._attack_setup('cpu')
H, W, S, N_imp = dims[0], dims[1], dims[4], dims[5]
src = sampler.get_all()
with torch.no_grad():
    featmaps = model.feature_net((src['src_rgbs'] + g.t('in/delta0')).squeeze(0).permute(0, 3, 1, 2))
rows = 6
rb = {k: (v[:rows * W] if k in ('ray_o', 'ray_d', 'rgb') else v) for k, v in src.items()}
# chunk sizes: 50 -> 8 chunks, the last one ragged (34 rays); 200 -> 2 chunks, fewer than 3 ranks (rank 2 renders nothing and
# learns the record layout from rank 0)
for chunk in (50, 200):
    for to in ((0, None) if chunk == 50 else (0,)):
        if shard is not None:
            shard.gather_render_to = to
            before = shard.collectives
        ret = render_single_image(ray_sampler=SimpleNamespace(H=rows, W=W), ray_batch=rb, model=model, projector=Projector('cpu'),
                                  chunk_size=chunk, det=True, N_samples=S, inv_uniform=True, N_importance=N_imp, white_bkgd=False,
                                  featmaps=featmaps, args=None, src_ray_batch=src, shard=shard)
        if shard is not None:
            n_chunks = -(-rows * W // chunk)
            # ONE collective per image (+ the record-layout broadcast when a rank had no chunk)
            assert shard.collectives - before == 1, shard.collectives - before
            if to == 0 and rank != 0:
                assert ret is None
                continue
        for level in ('outputs_coarse', 'outputs_fine'):
            for k, v in ret[level].items():
                out['ibr/%%d/%%s/%%s/%%s' %% (chunk, 'all' if to is None else 'r0', level, k)] = v.numpy()

# ---- GNT flavour (no compositing stage; weights / depth stay None without ret_alpha)
from nerfool_amd.gnt.model import GNTModel
from nerfool_amd.gnt.render_image import render_single_image as gnt_render_single_image
from nerfool_amd.synthetic import make_scene
torch.manual_seed(0)
for ret_alpha in (False, True):
    gargs = SimpleNamespace(netwidth=64, trans_depth=2, single_net=True, ret_alpha=ret_alpha, coarse_feat_dim=32, fine_feat_dim=32,
                            N_rand=16, N_samples=8, N_importance=0, inv_uniform=True, det=True, white_bkgd=False, chunk_size=512,
                            ckpt_path=None)
    torch.manual_seed(3)
    gmodel = GNTModel(gargs, device='cpu')
    gmodel.switch_to_eval()
    gdata = make_scene(24, 32, 3, seed=21, tilt=0.3)
    gs = RaySamplerSingleImage(gdata, 'cpu')
    grb = gs.get_all()
    with torch.no_grad():
        fm = gmodel.feature_net(grb['src_rgbs'].squeeze(0).permute(0, 3, 1, 2))
    grb = {k: (v[:8 * 32] if k in ('ray_o', 'ray_d', 'rgb') else v) for k, v in grb.items()}       # the first 8 image rows
    gs = SimpleNamespace(H=8, W=32)
    if shard is not None:
        shard.gather_render_to = 0
    ret = gnt_render_single_image(ray_sampler=gs, ray_batch=grb, model=gmodel, projector=Projector('cpu'), chunk_size=60, N_samples=8,
                                  inv_uniform=True, det=True, N_importance=0, white_bkgd=False, featmaps=fm, ret_alpha=ret_alpha,
                                  single_net=True, shard=shard)
    if rank == 0:
        assert ret['outputs_fine'] is None
        for k, v in ret['outputs_coarse'].items():
            assert (v is None) == (k != 'rgb' and not ret_alpha)
            if v is not None:
                out['gnt/%%d/%%s' %% (ret_alpha, k)] = v.numpy()
    else:
        assert ret is None
assert chunk_block(8, 0, 3) == (0, 3) and chunk_block(8, 1, 3) == (3, 6) and chunk_block(8, 2, 3) == (6, 8)
assert chunk_block(2, 2, 3) == (2, 2)
np.savez(os.path.join(%(out)r, 'render_rank%%d_of_%%d.npz' %% (rank, world)), **out)
if world > 1:
    dist.destroy_process_group()
'''


@pytest.mark.timeout(2400)
def test_sharded_render_equals_single_process_bit_for_bit(tmp_path):
    """render_single_image(shard=RayShard) -- SURVEY 8e "contiguous ray ranges per rank, gather to rank 0": world 2 and world 3
    (ragged chunk count, a ragged last chunk, a rank without any chunk), gather to rank 0 and all-gather, both flavours: every
    field of both levels equals the single-process image bit for bit, one collective per image."""
    _build_and_script(tmp_path)
    script = tmp_path / 'render_worker.py'
    script.write_text(RENDER_WORKER % dict(root=ROOT, harness=HARNESS, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='2')
    port = 29561
    for world in (1, 2, 3):
        port += 1
        procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world)))
                 for r in range(world)]
        assert all(p.wait() == 0 for p in procs)
    ref = np.load(tmp_path / 'render_rank0_of_1.npz')
    assert any(k.startswith('gnt/1/weights') for k in ref.files) and any(k.startswith('ibr/50/all/') for k in ref.files)
    for world in (2, 3):
        got = np.load(tmp_path / ('render_rank0_of_%d.npz' % world))
        assert set(got.files) == set(ref.files)
        for k in ref.files:
            assert got[k].dtype == ref[k].dtype and got[k].shape == ref[k].shape, k
            assert np.array_equal(got[k], ref[k]), 'world %d: %s differs from the single-process render' % (world, k)
        # the all-gather form delivers the same image to the other ranks too
        other = np.load(tmp_path / ('render_rank%d_of_%d.npz' % (world - 1, world)))
        keys = [k for k in ref.files if k.startswith('ibr/50/all/')]
        assert keys and set(other.files) == set(keys)
        for k in keys:
            assert np.array_equal(other[k], ref[k]), k


BENCH_TINY = ['--device', 'cpu-standin', '--height', '48', '--width', '64', '--n-rand', '24', '--samples', '8', '--importance', '8',
              '--steps', '2', '--warmup', '1']


def _bench(extra_args, env=None):
    import json
    if not os.path.exists('/opt/rocm/lib/llvm/bin/clang++'):
        pytest.skip('clang++ of the ROCm toolchain is needed to build the CPU stand-in')
    subprocess.run([os.path.join(HARNESS, 'build.sh')], check=True, capture_output=True)
    clean = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    clean.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + extra_args + BENCH_TINY, env=clean, capture_output=True,
                       text=True, timeout=900)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    return p.returncode, [json.loads(l) for l in lines], p.stderr


@pytest.mark.timeout(1200)
def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus N` (the driver's command shape, no torch.distributed.run in front): the parent starts N fresh rank
    processes itself and relays rank 0's ONE JSON line (reference precedent for self-initialising ranks:
    eval/gnt/eval_adv.py:1211-1214).  Here on the kernels' CPU stand-in over gloo -- a functional check of launcher + sharding."""
    rc, out, err = _bench(['--gpus', '2'])
    assert rc == 0, err[-2000:]
    assert len(out) == 1, 'exactly ONE JSON line on stdout'
    line = out[0]
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['warmup'] == 1
    assert line['config']['collectives_per_step'] == 2          # north-star form: 16-byte all-reduce + ONE d delta all-reduce
    assert line['config']['rays_per_step_all_ranks'] == 48 and line['scaling'] == 'weak'
    assert line['value'] > 0 and abs(line['value'] - 48 * 2 / (line['ms_per_step'] * 2e-3)) <= 1e-6 * line['value']
    assert 'cpu-standin' in line['device']
    # N = 1 keeps its shape (no spawn, no collectives)
    rc1, out1, err1 = _bench(['--gpus', '1'])
    assert rc1 == 0, err1[-2000:]
    assert len(out1) == 1 and out1[0]['n_gpus'] == 1 and out1[0]['config']['collectives_per_step'] is None
    assert set(out1[0]) == set(line)
    # the same final loss on the union of the rays is NOT expected (weak scaling doubles the batch); strong scaling reproduces N = 1
    rc2, out2, err2 = _bench(['--gpus', '2', '--scaling', 'strong'])
    assert rc2 == 0, err2[-2000:]
    assert abs(out2[0]['extra']['final_loss'] - out1[0]['extra']['final_loss']) <= 1e-4 * abs(out1[0]['extra']['final_loss'])


@pytest.mark.timeout(600)
def test_bench_parent_fails_when_a_rank_fails():
    """a rank that dies takes the whole run down with a non-zero exit code instead of leaving its peers in a collective"""
    rc, out, err = _bench(['--gpus', '2'], env={'NERFOOL_BENCH_FAIL_RANK': '1'})
    assert rc != 0 and out == []


@pytest.mark.timeout(600)
def test_bench_under_torch_distributed_run_still_works():
    import json
    if not os.path.exists('/opt/rocm/lib/llvm/bin/clang++'):
        pytest.skip('clang++ of the ROCm toolchain is needed to build the CPU stand-in')
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29577', os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + BENCH_TINY,
                       capture_output=True, text=True, timeout=580)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.lstrip().startswith('{')]
    assert len(lines) == 1 and lines[0]['n_gpus'] == 2 and lines[0]['config']['collectives_per_step'] == 2
