"""Diagnostic (GPU): which ATen operators (not our C-ABI kernels) still launch GPU kernels inside one PGD step, with call stacks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
sys.argv = [sys.argv[0], '--extras', '0']
a = bench.parse()
dev = torch.device('cuda', 0)
args, data, model, sampler, src_ray_batch, projector, EA = bench.build_problem(a, dev)
attack = EA.PGDAttack(args, model, projector, src_ray_batch)
for _ in range(3):
    attack.step(data)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    attack.step(data)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    cuda_t = getattr(e, 'self_device_time_total', 0) or getattr(e, 'self_cuda_time_total', 0)
    if cuda_t > 0 and e.key.startswith('aten::'):
        rows.append((cuda_t, e.key, e.count, str(e.input_shapes)[:90], [s for s in e.stack if 'nerfool_amd' in s or 'bench.py' in s][:3]))
for r in sorted(rows, reverse=True)[:30]:
    print('%8.1f us  %-28s x%-3d %s\n            %s' % r)
