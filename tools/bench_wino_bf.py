"""Timing of the bf16-split Winograd kernel (csrc/nf_wino_bf.hip) on the ResUNet's 3x3 layers, forward and backward-data, for the
operand forms fp32 / bf16x3 / bf16; with a library argument: a tuning / ablation build (tools/build_variant.sh).
usage: python tools/bench_wino_bf.py [iters] [library.so]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerfool_amd import ops                             # noqa: E402

if len(sys.argv) > 2:
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'host_harness'))
    import standin
    standin.use_library(sys.argv[2], emulated=False)


def timed(fn, iters):
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


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    tot = {0: 0.0, 3: 0.0, 1: 0.0}
    for (ci, co, H, W, count) in ((64, 64, 189, 252, 6), (128, 128, 95, 126, 7), (256, 256, 48, 63, 11), (256, 128, 96, 126, 2), (128, 64, 192, 252, 2)):
        x = torch.randn(4, ci, H + 2, W + 2, device=dev)
        w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        g = torch.randn(4, co, H, W, device=dev)
        line = '%3d->%3d %3dx%3d:' % (ci, co, H, W)
        for ns in (0, 3, 1):
            rf, rb = ops.wino_pack(w, False, dev, None, ns), ops.wino_pack(w, True, dev, None, ns)
            tf = timed(lambda: ops.conv3x3_wino(rf, x, co, 0, n_split=ns), iters)
            tb = timed(lambda: ops.conv3x3_wino(rb, g, ci, 2, n_split=ns), iters)
            tot[ns] += count * (tf + tb)
            line += '  %s fwd %6.1f bwd %6.1f us' % ({0: 'fp32  ', 3: 'bf16x3', 1: 'bf16  '}[ns], tf, tb)
        print(line, flush=True)
    print('weighted per step (unsplit backward): fp32 %.2f ms, bf16x3 %.2f ms, bf16 %.2f ms' % (tot[0] / 1e3, tot[3] / 1e3, tot[1] / 1e3))


if __name__ == '__main__':
    main()
