import cProfile, pstats, io, sys, os, time
sys.path.insert(0, '/root/repo')
sys.argv = ['bench.py', '--steps', '1', '--warmup', '1', '--cpu-iters', '0', '--render-chunks', '0']
import torch
import bench
a = bench.parse()
dev = torch.device('cuda', 0)
args, data, model, sampler, src_ray_batch, projector, EA = bench.build_problem(a, dev)
attack = EA.PGDAttack(args, model, projector, src_ray_batch)
for _ in range(4):
    attack.step(data)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    attack.step(data)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue %.2f ms/step, total %.2f ms/step' % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    attack.step(data)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
