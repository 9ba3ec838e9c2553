"""Does MIOpen's exhaustive find mode (torch.backends.cudnn.benchmark) pick faster kernels for the four convolutions left on it?
usage: python tools/exp_miopen_find.py [0|1]   (prints ms/step)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
flag = len(sys.argv) > 1 and sys.argv[1] == '1'
torch.backends.cudnn.benchmark = flag
import bench                                                      # noqa: E402

sys.argv = [sys.argv[0], '--cpu-iters', '0', '--render-chunks', '0']
a = bench.parse()
dev = torch.device('cuda', 0)
args, data, model, sampler, src_ray_batch, projector, EA = bench.build_problem(a, dev)
attack = EA.PGDAttack(args, model, projector, src_ray_batch)
for _ in range(5):
    attack.step(data)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    attack.step(data)
torch.cuda.synchronize()
print('cudnn.benchmark=%s: %.3f ms/step' % (flag, (time.perf_counter() - t0) / 50 * 1e3))
