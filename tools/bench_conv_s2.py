"""The stride-2 direct convolutions (csrc/nf_conv_s2.hip) against MIOpen on the four layers of the ResUNet at BASELINE config 2
(4 images 756 x 1008): time per call and accuracy against a float64 CPU convolution (small crop).  usage: python tools/bench_conv_s2.py [library.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from nerfool_amd import ops
if len(sys.argv) > 1:      # a tuning build (tools/build_variant.sh)
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'host_harness'))
    import standin
    standin.use_library(sys.argv[1], False)

aten = torch.ops.aten


def timed(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, out


gen = torch.Generator().manual_seed(1)
for (cin, cout, ks, Hi, Wi) in ((3, 64, 7, 762, 1014), (64, 64, 3, 380, 506), (64, 128, 3, 191, 254), (128, 256, 3, 97, 128)):
    N = 4
    x = torch.randn(N, cin, Hi, Wi, generator=gen).cuda()
    w = (torch.randn(cout, cin, ks, ks, generator=gen) * 0.05).cuda()
    rf, rb = ops.conv_s2_pack(w, False, 'cuda'), ops.conv_s2_pack(w, True, 'cuda')
    t_m, ym = timed(lambda: aten.convolution(x, w, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1))
    t_o, yo = timed(lambda: ops.conv_s2_fwd(rf, x, cout, ks))
    g = torch.randn(ym.shape, generator=gen).cuda()
    t_mb, dm = timed(lambda: aten.convolution_backward(g, x, w, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0])
    t_ob, do = timed(lambda: ops.conv_s2_bwd(rb, g, cin, ks, Hi, Wi))
    fl = 2.0 * N * cout * cin * ks * ks * ym.shape[2] * ym.shape[3]
    # accuracy on a crop against float64
    xc = x[:1, :, :41, :73].cpu().double().requires_grad_(True)
    yc = F.conv2d(xc, w.cpu().double(), stride=2)
    ef = float((ops.conv_s2_fwd(rf, x[:1, :, :41, :73].contiguous(), cout, ks).cpu().double() - yc).abs().max() / yc.abs().max())
    gc = g[:1, :, :yc.shape[2], :yc.shape[3]].contiguous()
    gref, = torch.autograd.grad(yc, xc, gc.cpu().double())
    eb = float((ops.conv_s2_bwd(rb, gc, cin, ks, 41, 73).cpu().double() - gref).abs().max() / gref.abs().max())
    if ks == 3:
        r3 = ops.conv_s2_pack_x3(w, True, 'cuda')
        t_3, d3 = timed(lambda: ops.conv_s2_bwd_x3(r3, g, cin, Hi, Wi))
        e3 = float((ops.conv_s2_bwd_x3(r3, gc, cin, 41, 73).cpu().double() - gref).abs().max() / gref.abs().max())
        print('      bf16x3 backward-data: %6.1f us (fp32 operands %6.1f us), err vs float64 %.1e' % (t_3, t_ob, e3))
        f3 = ops.conv_s2_pack_x3(w, False, 'cuda')
        t_f3, _ = timed(lambda: ops.conv_s2_fwd_x3(f3, x, cout))
        ef3 = float((ops.conv_s2_fwd_x3(f3, x[:1, :, :41, :73].contiguous(), cout).cpu().double() - yc).abs().max() / yc.abs().max())
        print('      bf16x3 forward:       %6.1f us (fp32 operands %6.1f us), err vs float64 %.1e' % (t_f3, t_o, ef3))
    if ks == 7:
        f7 = ops.conv_s2_stem_pack_x3(w, 'cuda')
        t_f7, _ = timed(lambda: ops.conv_s2_stem_fwd_x3(f7, x, cout))
        ef7 = float((ops.conv_s2_stem_fwd_x3(f7, x[:1, :, :41, :73].contiguous(), cout).cpu().double() - yc).abs().max() / yc.abs().max())
        print('      bf16x3 stem forward:  %6.1f us (fp32 operands %6.1f us), err vs float64 %.1e' % (t_f7, t_o, ef7))
    print('%3d -> %3d %dx%d s2 at %dx%d: fwd MIOpen %6.1f us  own %6.1f us (%.1f TFLOP/s) | bwd-data MIOpen %6.1f us  own %6.1f us (%.1f TFLOP/s) | '
          'err fwd %.1e bwd %.1e | own vs MIOpen %.1e %.1e'
          % (cin, cout, ks, ks, Hi, Wi, t_m, t_o, fl / t_o / 1e6, t_mb, t_ob, fl / t_ob / 1e6, ef, eb,
             float((yo - ym).abs().max() / ym.abs().max()), float((do - dm).abs().max() / dm.abs().max())), flush=True)
