"""Isolated timing of the fused CNN glue kernels (csrc/nf_cnn.hip) at the ResUNet's activation shapes of BASELINE
config 2 (4 x 756 x 1008 sources).  usage: python tools/bench_cnn_glue.py [iters] [library.so]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfool_amd import _lib, ops                             # noqa: E402


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
    return e0.elapsed_time(e1) / iters * 1e3        # us


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    if len(sys.argv) > 2:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'host_harness'))
        import standin          # test hook: bind a tuning build of the kernel sources
        standin.use_library(sys.argv[2], emulated=False)
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    tot = 0.0
    #        N  C    H    W   pad act           norm  res   count per step (fwd + bwd pairs in the ResUNet)
    cases = ((4, 3, 756, 1008, 3, ops.ACT_NONE, False, False, 1),
             (4, 64, 378, 504, 1, ops.ACT_RELU, True, False, 1),
             (4, 64, 189, 252, 1, ops.ACT_RELU, True, False, 3), (4, 64, 189, 252, 1, ops.ACT_RELU, True, True, 3),
             (4, 128, 95, 126, 1, ops.ACT_RELU, True, False, 4), (4, 128, 95, 126, 1, ops.ACT_RELU, True, True, 4),
             (4, 256, 48, 63, 1, ops.ACT_RELU, True, False, 6), (4, 256, 48, 63, 1, ops.ACT_RELU, True, True, 6),
             (4, 128, 96, 126, 0, ops.ACT_ELU, True, False, 2), (4, 256, 96, 126, 1, ops.ACT_NONE, False, False, 1),
             (4, 64, 192, 252, 0, ops.ACT_ELU, True, False, 2), (4, 128, 192, 252, 1, ops.ACT_NONE, False, False, 1))
    for (N, C, H, W, pad, act, norm, use_res, count) in cases:
        x = torch.randn(N, C, H, W, device=dev)
        gamma = torch.rand(C, device=dev) + 0.5 if norm else None
        beta = torch.randn(C, device=dev) * 0.1 if norm else None
        res_store = torch.randn(N, C, H + 2, W + 2, device=dev) if use_res else None
        res = res_store[:, :, 1:-1, 1:-1] if use_res else None
        yp, mean, rstd = ops.in_act_pad_fwd(x, gamma, beta, res, act, pad)
        dyp = torch.randn_like(yp)
        A = x.numel() * 4 / 1e6
        t_f = timed(lambda: ops.in_act_pad_fwd(x, gamma, beta, res, act, pad), iters)
        t_b = timed(lambda: ops.in_act_pad_bwd(dyp, None, yp, x if norm else None, gamma, mean, rstd, act, pad, use_res, beta=beta), iters)
        bytes_f = A * ((2 if norm else 1) + (1 if use_res else 0)) + yp.numel() * 4 / 1e6
        bytes_b = 2 * yp.numel() * 4 / 1e6 + A * ((1 + 1 + 3) if norm else 1)
        tot += count * (t_f + t_b)
        print('[%d,%3d,%3d,%4d] pad %d norm %d res %d: fwd %6.1f us (%5.0f GB/s)  bwd %6.1f us (%5.0f GB/s)  x%d' %
              (N, C, H, W, pad, norm, use_res, t_f, bytes_f / t_f * 1e3, t_b, bytes_b / t_b * 1e3, count), flush=True)
    xs = torch.randn(4, 256, 48, 63, device=dev)
    t_u = timed(lambda: ops.upsample2x_pad_fwd(xs, 1), iters)
    xs2 = torch.randn(4, 128, 96, 126, device=dev)
    t_u2 = timed(lambda: ops.upsample2x_pad_fwd(xs2, 1), iters)
    print('upsample+pad: %.1f us, %.1f us' % (t_u, t_u2))
    print('weighted glue total per step: %.3f ms' % ((tot + t_u + t_u2) / 1e3))


if __name__ == '__main__':
    main()
