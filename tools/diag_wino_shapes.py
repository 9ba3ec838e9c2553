"""Diagnostic (GPU): the Winograd 3x3 kernel over small and ragged shapes (forward on pre-padded input and backward-data),
both workgroup widths, repeated launches, against a float64 CPU convolution."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from nerfool_amd import ops

gen = torch.Generator().manual_seed(5)
shapes = [(4, 64, 64, 12, 16), (4, 128, 128, 6, 8), (4, 256, 256, 3, 4), (4, 256, 128, 6, 8), (4, 128, 64, 12, 16),
          (4, 64, 64, 24, 32), (4, 128, 128, 12, 16), (4, 256, 256, 6, 8), (4, 256, 128, 12, 16), (4, 128, 64, 24, 32),
          (1, 64, 64, 1, 1), (1, 64, 64, 2, 3), (3, 256, 256, 1, 2), (2, 128, 64, 17, 9), (1, 64, 32, 5, 40), (10, 128, 128, 50, 50)]
bad = 0
for (N, ci, co, H, W) in shapes:
    wgt = torch.randn(co, ci, 3, 3, generator=gen) * 0.05
    x = torch.randn(N, ci, H + 2, W + 2, generator=gen)
    gy = torch.randn(N, co, H, W, generator=gen)
    ref = F.conv2d(x.double(), wgt.double())
    gref = F.conv_transpose2d(gy.double(), wgt.double())
    wg, xg, gg = wgt.cuda(), x.cuda(), gy.cuda()
    for kg in (64, 32):
        if co % kg or ci % kg:
            continue
        rf, rb = ops.wino_pack(wg, False, 'cuda', kg), ops.wino_pack(wg, True, 'cuda', kg)
        ef = eb = 0.0
        for _ in range(5):
            got = ops.conv3x3_wino(rf, xg, co, 0, k_per_group=kg).cpu().double()
            ggot = ops.conv3x3_wino(rb, gg, ci, 2, k_per_group=kg).cpu().double()
            ef = max(ef, float((got - ref).abs().max() / ref.abs().max()))
            eb = max(eb, float((ggot - gref).abs().max() / gref.abs().max()))
        flag = '' if max(ef, eb) < 2e-5 else '   <-- BAD'
        bad += bool(flag)
        print('N %2d  %3d -> %3d  %3dx%-3d  kg %2d   fwd %.2e   bwd-data %.2e%s' % (N, ci, co, H, W, kg, ef, eb, flag), flush=True)
print('bad:', bad)
