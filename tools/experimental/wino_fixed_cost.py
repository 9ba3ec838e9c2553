"""Fixed vs per-chunk cost of a k_wino3x3_bf launch: forward at the three plane sizes of config 2 with 1, 2, 4, 8, 16 chunks of input channels,
operand forms bf16x3 / bf16x2 / bf16 (profiles/r06_wino_ablation.txt).  usage: python tools/experimental/wino_fixed_cost.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from nerfool_amd import ops
def timed(fn, iters=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dev = torch.device('cuda', 0)
for (co, H, W) in ((64, 189, 252), (128, 95, 126), (256, 48, 63)):
    for ci in (16, 32, 64, 128, 256):
        x = torch.randn(4, ci, H + 2, W + 2, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        line = '%3d->%3d %3dx%3d chunks %2d:' % (ci, co, H, W, ci // 16)
        for ns in (3, 2, 1):
            rf = ops.wino_pack(w, False, dev, None, ns)
            line += '  ns%d %6.1f us' % (ns, timed(lambda: ops.conv3x3_wino(rf, x, co, 0, n_split=ns)))
        print(line, flush=True)
