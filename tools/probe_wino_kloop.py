"""How much of the Winograd kernel's time is workgroup prologue / epilogue?  Same output block, growing reduction length:
Winograd-domain TFLOP/s against c_in (189 x 252 x 4 images, 64 output channels).  usage: python tools/probe_wino_kloop.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfool_amd import ops                             # noqa: E402
from bench_conv3x3 import timed                          # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    for (co, H, W) in ((64, 189, 252), (128, 95, 126), (256, 48, 63)):
        for ci in (16, 64, 128, 256, 512, 1024):
            x = torch.randn(4, ci, H + 2, W + 2, device=dev)
            w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
            for kpg in (64, 32):
                rf = ops.wino_pack(w, False, dev, kpg)
                t, _ = timed(lambda: ops.conv3x3_wino(rf, x, co, 0, k_per_group=kpg), 20)
                fl = 2.0 * 4 * H * W * ci * co * 4          # Winograd-domain products
                print('%4d->%3d %3dx%3d kpg %d: %7.1f us  %5.1f TF (Winograd domain)' % (ci, co, H, W, kpg, t, fl / t / 1e6), flush=True)


if __name__ == '__main__':
    main()
