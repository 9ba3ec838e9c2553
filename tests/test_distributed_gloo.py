"""CPU, world_size 2, gloo: the ray-sharded PGD step.  Each rank differentiates its slice of the step's rays
through the CPU stand-in build of the kernels; the mask counts are all-reduced before the loss is normalised and
d(delta) is all-reduced once; the result must equal the single-process gradient of the union of the rays, and both
ranks must hold the same delta after the fused update."""
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
from nerfool_amd import _lib
_lib.use_library_for_tests(os.path.join(%(harness)r, 'libnerfool_emu.so'))
import parity_cases as pc
from nerfool_amd import eval_adv as EA
from nerfool_amd.ibrnet.projection import Projector

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
if world > 1:
    dist.init_process_group('gloo', rank=rank, world_size=world)
torch.set_num_threads(2)
g, args, model, data, sampler, dims = pc._attack_setup('cpu')
src = sampler.get_all()
picks = g.np('adam/selected_inds')[0]
shard = EA.RayShard() if world > 1 else None
atk = EA.PGDAttack(args, model, Projector('cpu'), src, shard=shard, delta=g.t('in/delta0').clone().requires_grad_(True))
mine = picks if world == 1 else picks[rank::world]
grad = atk.gradient(data, select_inds=mine).clone()
atk.apply(grad)
np.savez(os.path.join(%(out)r, 'rank%%d_of_%%d.npz' %% (rank, world)), grad=grad.numpy(), delta=atk.delta.detach().numpy(),
         loss=float(atk.last_loss))
if world > 1:
    dist.destroy_process_group()
'''


@pytest.mark.timeout(900)
def test_ray_sharded_step_equals_single_process(tmp_path):
    if not os.path.exists('/opt/rocm/lib/llvm/bin/clang++'):
        pytest.skip('clang++ of the ROCm toolchain is needed to build the CPU stand-in')
    subprocess.run([os.path.join(HARNESS, 'build.sh')], check=True, capture_output=True)
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT, harness=HARNESS, out=str(tmp_path)))
    # the shape-generic kernels emulate ~30x faster than the MFMA ones; the sharding logic under test is the same
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29531', OMP_NUM_THREADS='2',
               NERFOOL_IBRNET_KERNELS='generic', NERFOOL_CNN='torch')
    single = subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK='0', WORLD_SIZE='1'))
    assert single.wait() == 0
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), WORLD_SIZE='2')) for r in range(2)]
    assert all(p.wait() == 0 for p in procs)
    ref = np.load(tmp_path / 'rank0_of_1.npz')
    r0 = np.load(tmp_path / 'rank0_of_2.npz')
    r1 = np.load(tmp_path / 'rank1_of_2.npz')
    scale = np.abs(ref['grad']).max()
    # all-reduced gradient == gradient of the union of the rays (different summation order only)
    assert np.abs(r0['grad'] - ref['grad']).max() <= 2e-4 * scale
    assert np.array_equal(r0['grad'], r1['grad']), 'ranks must hold the identical all-reduced gradient'
    assert np.array_equal(r0['delta'], r1['delta']), 'delta must stay replicated'
    # per-rank losses are partial sums over the global denominator: they add up to the single-process loss
    assert abs(float(r0['loss']) + float(r1['loss']) - float(ref['loss'])) <= 1e-5 * abs(float(ref['loss']))
    assert np.abs(r0['delta'] - ref['delta']).mean() <= 1e-6
