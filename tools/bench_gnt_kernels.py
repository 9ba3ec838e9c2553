"""Isolated timing of the GNT network kernels at BASELINE config 4 (512 rays, 64 samples, 10 views, depth 8): the
shape-generic forward, the matrix-core forward, and the backward.  usage: python tools/bench_gnt_kernels.py [iters] [lib.so]"""
import os
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfool_amd import _lib, ops                             # noqa: E402
from nerfool_amd.gnt.transformer_network import GNT           # noqa: E402

GNT_FWD_FLOPS_PER_RAY = 217.0e6       # V = 10, S = 64, depth 8


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    if len(sys.argv) > 2:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'host_harness'))
        import standin          # test hook: bind a tuning build of the kernel sources
        standin.use_library(sys.argv[2], emulated=False)
    dev = torch.device('cuda', 0)
    R, S, V, depth = 512, 64, 10, 8
    torch.manual_seed(0)
    net = GNT(SimpleNamespace(netwidth=64, trans_depth=depth), in_feat_ch=32, posenc_dim=63, viewenc_dim=63)
    blob = ops.pack_gnt_blob(net.state_dict(), depth, dev)
    mblob = ops.pack_gnt_mfma_blob(blob, depth)
    gen = torch.Generator().manual_seed(1)
    rgb_feat = torch.randn(R, S, V, 35, generator=gen).to(dev)
    rd = torch.randn(R, S, V, 4, generator=gen).to(dev)
    mask = (torch.rand(R, S, V, generator=gen) > 0.1).float().to(dev)
    pts = torch.randn(R, S, 3, generator=gen).to(dev)
    ray_d = torch.randn(R, 3, generator=gen).to(dev)
    d_rgb = torch.randn(R, 3, generator=gen).to(dev)
    args = (rgb_feat, rd, mask, pts, ray_d, depth)

    def timed(fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters, out

    t_gen, (rgb_a, ws_a) = timed(lambda: ops.gnt_fwd(blob, *args, save=True))
    t_mf, (rgb_b, ws_b) = timed(lambda: ops.gnt_fwd_mfma(mblob, *args, save=True))
    t_mf0, (rgb_c, _) = timed(lambda: ops.gnt_fwd_mfma(mblob, *args, save=False))
    t_bwd, g_b = timed(lambda: ops.gnt_bwd(blob, rd, mask, d_rgb, ws_a, (R, S, V), depth))
    t_bmf, g_c = timed(lambda: ops.gnt_bwd_mfma(mblob, mask, d_rgb, ws_b, (R, S, V), depth))
    F = GNT_FWD_FLOPS_PER_RAY * R
    print('rgb max diff %.2e (no-save %.2e)' % (float((rgb_a - rgb_b).abs().max()), float((rgb_a - rgb_c).abs().max())))
    print('backward: generic %.2f ms, matrix cores %.2f ms (%.1f TFLOP/s algorithmic), rel diff %.2e rel-L2 %.2e' %
          (t_bwd, t_bmf, F / t_bmf / 1e9, float((g_c - g_b).abs().max() / g_b.abs().max()), float((g_c - g_b).norm() / g_b.norm())))
    print('forward generic %.2f ms (%.1f TFLOP/s) | matrix cores %.2f ms (%.1f TFLOP/s), no save %.2f ms | backward %.2f ms' %
          (t_gen, F / t_gen / 1e9, t_mf, F / t_mf / 1e9, t_mf0, t_bwd))


if __name__ == '__main__':
    main()
