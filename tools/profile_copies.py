"""Which Python call sites launch ATen copy / fill / add kernels inside one PGD step (torch.profiler with stacks).
usage: python tools/profile_copies.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                      # noqa: E402

sys.argv = [sys.argv[0], '--cpu-iters', '0', '--render-chunks', '0']
a = bench.parse()
dev = torch.device('cuda', 0)
args, data, model, sampler, src_ray_batch, projector, EA = bench.build_problem(a, dev)
attack = EA.PGDAttack(args, model, projector, src_ray_batch)
for _ in range(3):
    attack.step(data)
torch.cuda.synchronize()
import traceback
_orig = {}
def _wrap(name):
    fn = getattr(torch.Tensor, name)
    _orig[name] = fn
    def w(self, *a, **k):
        if self.is_cuda and self.numel() > 500000 and not (name == 'contiguous' and self.is_contiguous()):
            fr = [f for f in traceback.extract_stack()[:-1] if 'nerfool_amd' in f.filename or 'bench' in f.filename][-2:]
            print('PY', name, tuple(self.shape), tuple(self.stride()), ' <- '.join('%s:%d' % (f.filename.split('/')[-1], f.lineno) for f in fr))
        return fn(self, *a, **k)
    setattr(torch.Tensor, name, w)
for nm in ('contiguous', 'clone', 'copy_', 'to', '__add__', '__iadd__', 'add_', 'add'):
    _wrap(nm)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    attack.step(data)
    torch.cuda.synchronize()
want = ('aten::copy_', 'aten::fill_', 'aten::add', 'aten::add_', 'aten::zero_', 'aten::upsample_bilinear2d_backward', 'aten::sub', 'aten::mul')
for ev in prof.events():
    if ev.name in want and ev.device_time_total > 3:
        stack = [s for s in ev.stack if 'nerfool_amd' in s or 'bench.py' in s][:3]
        print('%-36s %8.1f us  shape %s  | %s' % (ev.name, ev.device_time_total, ev.input_shapes if ev.input_shapes else '', ' <- '.join(s.split('/')[-1] for s in stack)))
