"""Isolated timing of the IBRNet network kernels (forward + backward) at the attack and render sizes; also the program
profiled with rocprofv3 --pmc for the counters quoted in DESIGN.md.  usage: python tools/bench_ibrnet_kernels.py [iters] [library.so]"""
import os
import sys

from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfool_amd import _lib, ops                             # noqa: E402
from nerfool_amd.ibrnet.mlp_network import IBRNet             # noqa: E402


def flops(R, S, V):
    return 2.0 * R * S * (V * 13256 + 6480 + 32 * S)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    if len(sys.argv) > 2:          # a tuning build (tools/build_variant.sh)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'host_harness'))
        import standin          # test hook: bind a tuning build of the kernel sources
        standin.use_library(sys.argv[2], emulated=False)
    dev = torch.device('cuda', 0)
    gen = torch.Generator().manual_seed(0)
    for (R, S, V, prec) in ((512, 64, 4, 'fp32'), (512, 128, 4, 'fp32'), (4096, 64, 4, 'fp32'), (4096, 128, 4, 'fp32'),
                            (512, 128, 8, 'fp32'), (512, 128, 8, 'bf16'), (512, 256, 8, 'fp32'), (512, 256, 8, 'bf16'),
                            (4096, 128, 8, 'fp32'), (4096, 128, 8, 'bf16'), (4096, 256, 8, 'fp32'), (4096, 256, 8, 'bf16'),
                            (4096, 128, 4, 'bf16'), (512, 64, 10, 'fp32'), (4096, 64, 10, 'fp32'), (4096, 64, 16, 'fp32')):
        torch.manual_seed(3)
        net = IBRNet(SimpleNamespace(anti_alias_pooling=1, ibrnet_precision=prec), in_feat_ch=32, n_samples=S).to(dev)
        blob, mblob = net._packed(dev)
        bb = net._bf16_blob if prec == 'bf16' else None
        rgb_feat = torch.randn(R, S, V, 35, generator=gen).to(dev)
        rd = torch.randn(R, S, V, 4, generator=gen)
        rd[..., :3] = torch.nn.functional.normalize(rd[..., :3], dim=-1)
        rd = rd.to(dev)
        mask = (torch.rand(R, S, V, generator=gen) > 0.1).float().to(dev)
        pe = net.pos_encoding
        d_raw = torch.randn(R, S, 4, generator=gen).to(dev)
        for _ in range(3):
            raw, ws = ops.ibrnet_fwd_mfma(mblob, blob, pe, rgb_feat, rd, mask, True, bf16_blob=bb)
            ops.ibrnet_bwd_mfma(mblob, blob, pe, rgb_feat, rd, mask, ws, d_raw, True, bf16_blob=bb)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        torch.cuda.synchronize()
        e[0].record()
        for _ in range(iters):
            raw, ws = ops.ibrnet_fwd_mfma(mblob, blob, pe, rgb_feat, rd, mask, True, bf16_blob=bb)
        e[1].record()
        for _ in range(iters):
            ops.ibrnet_bwd_mfma(mblob, blob, pe, rgb_feat, rd, mask, ws, d_raw, True, bf16_blob=bb)
        e[2].record()
        torch.cuda.synchronize()
        tf, tb = e[0].elapsed_time(e[1]) / iters, e[1].elapsed_time(e[2]) / iters
        F = flops(R, S, V)
        print('R %5d S %3d V %d %s: fwd %.3f ms (%.1f TFLOP/s)  bwd %.3f ms (%.1f TFLOP/s algorithmic)' %
              (R, S, V, prec, tf, F / tf / 1e9, tb, F / tb / 1e9), flush=True)


if __name__ == '__main__':
    main()
