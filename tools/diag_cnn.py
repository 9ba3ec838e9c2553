"""Diagnostic (GPU): accuracy of the MIOpen ResUNet forward / backward-data against a float64 CPU evaluation of the same
module, and its speed at the benchmark size.  Usage: python tools/diag_cnn.py [small|full]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nerfool_amd.ibrnet.feature_network import ResUNet

mode = sys.argv[1] if len(sys.argv) > 1 else 'small'
if os.environ.get('NF_CUDNN_BENCHMARK'):
    torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
net = ResUNet()
for p in net.parameters():
    p.requires_grad_(False)
if mode == 'small':
    x = torch.rand(4, 3, 96, 128)
    gen = torch.Generator().manual_seed(1)
    net64 = ResUNet().double(); net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    for p in net64.parameters(): p.requires_grad_(False)
    x64 = x.double().requires_grad_(True)
    c64, f64 = net64(x64)
    G = torch.randn(c64.shape, generator=gen, dtype=torch.float64)
    g64, = torch.autograd.grad((c64 * G).sum() + (f64 * G).sum(), x64)
    xc = x.clone().requires_grad_(True)
    cc, fc = net(xc)
    gc, = torch.autograd.grad((cc * G.float()).sum() + (fc * G.float()).sum(), xc)
    print('CPU fp32 : fwd rel-L2 %.2e  bwd rel-L2 %.2e' % (float((cc.double() - c64).norm() / c64.norm()), float((gc.double() - g64).norm() / g64.norm())))
    netg = net.cuda()
    xg = x.cuda().requires_grad_(True)
    cg, fg = netg(xg)
    gg, = torch.autograd.grad((cg * G.float().cuda()).sum() + (fg * G.float().cuda()).sum(), xg)
    print('GPU fp32 : fwd rel-L2 %.2e  bwd rel-L2 %.2e   [MIOPEN_DEBUG_CONV_WINOGRAD=%s deterministic=%s]' % (
        float((cg.cpu().double() - c64).norm() / c64.norm()), float((gg.cpu().double() - g64).norm() / g64.norm()),
        os.environ.get('MIOPEN_DEBUG_CONV_WINOGRAD'), torch.backends.cudnn.deterministic))
else:
    netg = net.cuda()
    x = torch.rand(4, 3, 756, 1008, device='cuda', requires_grad=True)
    for tag in ('nchw',):
        xin = x if tag == 'nchw' else x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        if tag == 'channels_last':
            netg = netg.to(memory_format=torch.channels_last)
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.time()
            c, f = netg(xin)
            torch.cuda.synchronize(); t1 = time.time()
            g, = torch.autograd.grad(c.sum() + f.sum(), xin)
            torch.cuda.synchronize(); t2 = time.time()
        print('%-14s fwd %.2f ms (%.1f TFLOP/s)  bwd-data %.2f ms   [WINOGRAD=%s benchmark=%s]' % (
            tag, 1e3 * (t1 - t0), 4 * 121.9e9 / (t1 - t0) / 1e12, 1e3 * (t2 - t1), os.environ.get('MIOPEN_DEBUG_CONV_WINOGRAD'),
            torch.backends.cudnn.benchmark))
