"""3x3 stride-1 convolutions of the ResUNet at BASELINE config 2 sizes: MIOpen (aten.convolution / convolution_backward on the
pre-padded activation) vs the Winograd matrix-core kernel (csrc/nf_wino.hip).
usage: python tools/bench_conv3x3.py [iters] [tuning build of the library, tools/build_variant.sh]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfool_amd import _lib, ops                             # noqa: E402

if len(sys.argv) > 2:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'host_harness'))
    import standin          # test hook: bind a tuning build of the kernel sources
    standin.use_library(sys.argv[2], emulated=False)

aten = torch.ops.aten


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, out


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    tot_m = tot_w = 0.0
    #        cin cout   H    W   count (forward + backward pairs per PGD step)
    for (ci, co, H, W, count) in ((64, 64, 189, 252, 6), (128, 128, 95, 126, 7), (256, 256, 48, 63, 11), (256, 128, 96, 126, 2),
                                  (128, 64, 192, 252, 2)):
        x = torch.randn(4, ci, H + 2, W + 2, device=dev)
        w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        rf, rb = ops.wino_pack(w, False, dev), ops.wino_pack(w, True, dev)
        t_mf, y = timed(lambda: aten.convolution(x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1), iters)
        g = torch.randn_like(y)
        t_mb, dx = timed(lambda: aten.convolution_backward(g, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0], iters)
        t_wf, y2 = timed(lambda: ops.conv3x3_wino(rf, x, co, 0), iters)
        t_wb, dx2 = timed(lambda: ops.conv3x3_wino(rb, g, ci, 2), iters)
        rf32, rb32 = ops.wino_pack(w, False, dev, 32), ops.wino_pack(w, True, dev, 32)
        t_f32, y3 = timed(lambda: ops.conv3x3_wino(rf32, x, co, 0, k_per_group=32), iters)
        t_b32, dx3 = timed(lambda: ops.conv3x3_wino(rb32, g, ci, 2, k_per_group=32), iters)
        t1 = (t_wf, t_f32, t_wb, t_b32, float((y - y3).abs().max() / y.abs().max()), float((dx - dx3).abs().max() / dx.abs().max()))
        # bf16-split operand forms (csrc/nf_wino_bf.hip): the three-way split (fp32-grade) and plain bf16
        for ns, tag in ((3, 'bf16x3'), (1, 'bf16')):
            r_f, r_b = ops.wino_pack(w, False, dev, None, ns), ops.wino_pack(w, True, dev, None, ns)
            t_f, y4 = timed(lambda: ops.conv3x3_wino(r_f, x, co, 0, n_split=ns), iters)
            t_b, dx4 = timed(lambda: ops.conv3x3_wino(r_b, g, ci, 2, n_split=ns), iters)
            msg = '      %-6s operands: fwd %.1f us, bwd %.1f us (rel err vs MIOpen %.1e %.1e)' % (
                tag, t_f, t_b, float((y - y4).abs().max() / y.abs().max()), float((dx - dx4).abs().max() / dx.abs().max()))
            if co > 64:
                r_f32, r_b32 = ops.wino_pack(w, False, dev, 32, ns), ops.wino_pack(w, True, dev, 32, ns)
                t_f2, _ = timed(lambda: ops.conv3x3_wino(r_f32, x, co, 0, k_per_group=32, n_split=ns), iters)
                msg += ' | 32 per workgroup: fwd %.1f us' % t_f2
            if ci > 64:
                t_b2, _ = timed(lambda: ops.conv3x3_wino(r_b32 if co > 64 else ops.wino_pack(w, True, dev, 32, ns), g, ci, 2, k_per_group=32, n_split=ns), iters)
                msg += ' bwd %.1f us' % t_b2
            print(msg)
        fl = 2.0 * 4 * H * W * ci * co * 9
        err_f = float((y - y2).abs().max() / y.abs().max())
        err_b = float((dx - dx2).abs().max() / dx.abs().max())
        tot_m += count * (t_mf + t_mb)
        tot_w += count * (t_wf + t_wb)
        print('      output channels per workgroup 64 / 32: fwd %.1f / %.1f us, bwd %.1f / %.1f us (rel err of the 32 form %.1e %.1e)' % t1)
        print('%3d->%3d %3dx%3d: MIOpen fwd %6.1f us (%5.1f TF) bwd %6.1f us | Winograd MFMA fwd %6.1f us (%5.1f TF) bwd %6.1f us | rel err %.1e %.1e  x%d'
              % (ci, co, H, W, t_mf, fl / t_mf / 1e6, t_mb, t_wf, fl / t_wf / 1e6, t_wb, err_f, err_b, count), flush=True)
    print('weighted per step: MIOpen %.2f ms, Winograd MFMA %.2f ms' % (tot_m / 1e3, tot_w / 1e3))


if __name__ == '__main__':
    main()
