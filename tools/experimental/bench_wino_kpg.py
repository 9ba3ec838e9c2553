"""k_wino3x3_bf at 64 vs 32 output channels per workgroup on thin grids (config 5's 32 x 32 / 64 x 64 planes, config 2's 48 x 63):
usage: python tools/experimental/bench_wino_kpg.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nerfool_amd import ops


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


dev = torch.device('cuda', 0)
torch.manual_seed(0)
for (n, ci, co, H, W) in ((8, 256, 256, 32, 32), (8, 128, 128, 64, 64), (8, 64, 64, 128, 128), (4, 256, 256, 48, 63), (4, 128, 128, 95, 126),
                          (8, 256, 128, 64, 64), (8, 128, 64, 128, 128)):
    x = torch.randn(n, ci, H + 2, W + 2, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    g = torch.randn(n, co, H, W, device=dev)
    line = '%d x %3d->%3d %3dx%3d:' % (n, ci, co, H, W)
    for kpg in (64, 32):
        if kpg == 32 and min(ci, co) <= 32:
            continue
        rf, rb = ops.wino_pack(w, False, dev, kpg, 3), ops.wino_pack(w, True, dev, kpg, 3)
        tf = timed(lambda: ops.conv3x3_wino(rf, x, co, 0, k_per_group=kpg, n_split=3))
        tb = timed(lambda: ops.conv3x3_wino(rb, g, ci, 2, k_per_group=kpg, n_split=3))
        line += '  kpg %d fwd %6.1f bwd %6.1f us' % (kpg, tf, tb)
    print(line, flush=True)
