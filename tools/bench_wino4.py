"""EXPERIMENT (not product): F(4x4,3x3) (tools/experimental/nf_wino4.hip) against the shipped F(2x2,3x3) kernel (csrc/nf_wino.hip) on
the stride-1 3x3 layers of the ResUNet at BASELINE config 2 (4 images), forward and backward-data; accuracy against a float64 CPU
convolution on a crop.  Builds tools/experimental/libnf_wino4.so (linked against the product library for nf_set_error) on first use.
usage: python tools/bench_wino4.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from nerfool_amd import _lib, ops
if os.environ.get('NF_VARIANT_LIB'):
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'host_harness'))
    import standin          # test hook: bind a tuning build of the kernel sources
    standin.use_library(os.environ['NF_VARIANT_LIB'], emulated=False)

EXP = os.path.join(ROOT, 'tools', 'experimental')
SO = os.path.join(EXP, 'libnf_wino4.so')
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(EXP, 'nf_wino4.hip')):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-ffp-contract=off', '-fPIC', '-std=c++17', '-shared',
                           '-I', os.path.join(ROOT, 'nerfool_amd', 'csrc'), '-I', EXP, os.path.join(EXP, 'nf_wino4.hip'), '-o', SO,
                           '-L', os.path.join(ROOT, 'nerfool_amd'), '-lnerfool_hip', '-Wl,-rpath,' + os.path.join(ROOT, 'nerfool_amd')])
_lib.lib()                                       # the product library first (nf_set_error)
W4 = ctypes.CDLL(SO, mode=ctypes.RTLD_GLOBAL)
W4.nf_wino4_pack_floats.restype = ctypes.c_int64


class _Wino4:
    """host wrappers of the experimental entry points (same shape as ops.wino_pack / ops.conv3x3_wino)"""

    @staticmethod
    def wino4_pack(weight, backward, device):
        w = weight.detach().to('cpu', torch.float32).contiguous()
        c_out, c_in = w.shape[0], w.shape[1]
        n_out, n_in = (c_in, c_out) if backward else (c_out, c_in)
        out = torch.empty(W4.nf_wino4_pack_floats(n_out, n_in), dtype=torch.float32)
        assert W4.nf_wino4_pack(ctypes.c_void_p(w.data_ptr()), c_out, c_in, int(bool(backward)), ctypes.c_void_p(out.data_ptr())) == 0
        return out.to(device)

    @staticmethod
    def conv3x3_wino4(records, x, c_out, pad):
        N, c_in, Hi, Wi = x.shape
        Ho, Wo = Hi - 2 + 2 * pad, Wi - 2 + 2 * pad
        y = torch.empty(N, c_out, Ho, Wo, dtype=torch.float32, device=x.device)
        xs, ys = x.stride(), y.stride()
        i64, vp = ctypes.c_int64, ctypes.c_void_p
        rc = W4.nf_conv3x3_wino4(vp(records.data_ptr()), vp(x.data_ptr()), i64(xs[0]), i64(xs[1]), i64(xs[2]), Hi, Wi, int(pad), vp(y.data_ptr()),
                                 i64(ys[0]), i64(ys[1]), i64(ys[2]), Ho, Wo, N, c_in, c_out, vp(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, _lib.last_error() if hasattr(_lib, 'last_error') else rc
        return y


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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for (ci, co, H, W) in ((64, 64, 189, 252), (128, 128, 95, 126), (256, 256, 48, 63), (256, 128, 96, 126), (128, 64, 192, 252)):
    w = (torch.randn(co, ci, 3, 3, generator=gen) * 0.05).cuda()
    x = torch.randn(N, ci, H + 2, W + 2, generator=gen).cuda()
    gy = torch.randn(N, co, H, W, generator=gen).cuda()
    r4f, r4b = _Wino4.wino4_pack(w, False, 'cuda'), _Wino4.wino4_pack(w, True, 'cuda')
    best2f = best2b = 1e9
    for kg in (64, 32):
        if co % kg == 0:
            rf = ops.wino_pack(w, False, 'cuda', kg)
            best2f = min(best2f, timed(lambda: ops.conv3x3_wino(rf, x, co, 0, k_per_group=kg))[0])
        if ci % kg == 0:
            rb = ops.wino_pack(w, True, 'cuda', kg)
            best2b = min(best2b, timed(lambda: ops.conv3x3_wino(rb, gy, ci, 2, k_per_group=kg))[0])
    t4f, y4 = timed(lambda: _Wino4.conv3x3_wino4(r4f, x, co, 0))
    t4b, d4 = timed(lambda: _Wino4.conv3x3_wino4(r4b, gy, ci, 2))
    xc = x[:1, :, :40, :70].cpu().double()
    ref = F.conv2d(xc, w.cpu().double())
    ef = float((_Wino4.conv3x3_wino4(r4f, x[:1, :, :40, :70].contiguous(), co, 0).cpu().double() - ref).abs().max() / ref.abs().max())
    gc = gy[:1, :, :38, :68].contiguous()
    gref = F.conv_transpose2d(gc.cpu().double(), w.cpu().double())
    eb = float((_Wino4.conv3x3_wino4(r4b, gc, ci, 2).cpu().double() - gref).abs().max() / gref.abs().max())
    fl = 2.0 * N * co * ci * 9 * H * W
    print('%3d -> %3d at %3dx%-3d: fwd F(2x2) %6.1f us  F(4x4) %6.1f us (%.0f TFLOP/s direct-equivalent) | bwd-data F(2x2) %6.1f us  F(4x4) %6.1f us | '
          'err fwd %.1e bwd %.1e' % (ci, co, H, W, best2f, t4f, fl / t4f / 1e6, best2b, t4b, ef, eb), flush=True)
