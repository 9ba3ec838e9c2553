"""Time the Winograd kernel of a tuning build (tools/build_variant.sh) on the ResUNet shapes: python tools/exp_wino_variants.py <lib.so>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfool_amd import _lib, ops                                   # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] != '-':
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'host_harness'))
    import standin          # test hook: bind a tuning build of the kernel sources
    standin.use_library(sys.argv[1], emulated=False)
dev = torch.device('cuda', 0)


def timed(fn, iters=30):
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


out = []
for (ci, co, H, W) in ((64, 64, 189, 252), (128, 128, 95, 126), (256, 256, 48, 63), (256, 128, 96, 126), (128, 64, 192, 252)):
    x = torch.randn(4, ci, H + 2, W + 2, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    g = torch.randn(4, co, H, W, device=dev)
    rf, rb = ops.wino_pack(w, False, dev), ops.wino_pack(w, True, dev)
    out.append('%d->%d %dx%d fwd %.1f bwd %.1f' % (ci, co, H, W, timed(lambda: ops.conv3x3_wino(rf, x, co, 0)), timed(lambda: ops.conv3x3_wino(rb, g, ci, 2))))
print(sys.argv[1] if len(sys.argv) > 1 else 'product', ' | '.join(out))
