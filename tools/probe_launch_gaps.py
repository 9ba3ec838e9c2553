"""Is there a per-launch cost that belongs to a kernel and not to the work it does?  Pairs (kernel, small glue kernel) issued
back to back on one stream, wall time per pair against the sum of the two kernels timed alone (no profiler attached).
usage: python tools/probe_launch_gaps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfool_amd import ops                             # noqa: E402


def timed(fn, iters=200):
    for _ in range(5):
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
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    c, H, W = 256, 48, 63
    x = torch.randn(4, c, H + 2, W + 2, device=dev)
    w = torch.randn(c, c, 3, 3, device=dev) * 0.05
    rf = ops.wino_pack(w, False, dev, 32)
    t = torch.randn(4, c, H, W, device=dev)
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    xs2 = torch.randn(4, 128, 97, 128, device=dev)
    w2 = torch.randn(256, 128, 3, 3, device=dev) * 0.05
    r2 = ops.conv_s2_pack(w2, False, dev)
    conv = lambda: ops.conv3x3_wino(rf, x, c, 0, k_per_group=32)
    glue = lambda: ops.in_act_pad_fwd(t, gamma, beta, None, ops.ACT_RELU, 1)
    s2 = lambda: ops.conv_s2_fwd(r2, xs2, 256, 3)
    a, b, d = timed(conv), timed(glue), timed(s2)
    ab = timed(lambda: (conv(), glue()))
    db = timed(lambda: (s2(), glue()))
    print('alone: winograd %.1f us, glue %.1f us, stride-2 conv %.1f us' % (a, b, d))
    print('pair winograd + glue %.1f us (sum %.1f, extra %.1f);  pair stride-2 + glue %.1f us (sum %.1f, extra %.1f)' % (ab, a + b, ab - a - b, db, d + b, db - d - b))


if __name__ == '__main__':
    main()
