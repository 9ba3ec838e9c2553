"""Do a matrix-core-bound kernel (Winograd convolution) and an HBM-bound one (norm / activation / padding glue) of two
independent half batches overlap when they are issued on two HIP streams?  usage: python tools/probe_stream_overlap.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfool_amd import ops                             # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    n = 2
    for (c, H, W) in ((64, 189, 252), (128, 95, 126), (256, 48, 63)):
        x = torch.randn(n, c, H + 2, W + 2, device=dev)
        w = torch.randn(c, c, 3, 3, device=dev) * 0.05
        rf = ops.wino_pack(w, False, dev)
        t = torch.randn(n, c, H, W, device=dev)
        gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        reps = 30

        def conv():
            for _ in range(reps):
                ops.conv3x3_wino(rf, x, c, 0)

        def glue():
            for _ in range(reps):
                ops.in_act_pad_fwd(t, gamma, beta, None, ops.ACT_RELU, 1)

        def run(fa, fb):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if fa is not None:
                with torch.cuda.stream(sa):
                    fa()
            if fb is not None:
                with torch.cuda.stream(sb):
                    fb()
            host = time.perf_counter() - t0
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e6 / reps, host * 1e6 / reps

        for _ in range(2):
            run(conv, glue)
        a = run(conv, None)
        b = run(None, glue)
        ab = run(conv, glue)
        # interleaved issue (the host alternates between the streams, as a two-stream executor would)
        def both():
            for _ in range(reps):
                with torch.cuda.stream(sa):
                    ops.conv3x3_wino(rf, x, c, 0)
                with torch.cuda.stream(sb):
                    ops.in_act_pad_fwd(t, gamma, beta, None, ops.ACT_RELU, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        both()
        torch.cuda.synchronize()
        il = (time.perf_counter() - t0) * 1e6 / reps
        print('%d ch %dx%d x%d images: conv alone %.1f us (host %.1f), glue alone %.1f us (host %.1f), two streams %.1f us, interleaved issue %.1f us'
              % (c, H, W, n, a[0], a[1], b[0], b[1], ab[0], il), flush=True)


if __name__ == '__main__':
    main()
