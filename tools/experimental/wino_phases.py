"""Phase cycles of k_wino3x3_bf from a -DWB_PHASE_TIMERS tuning build (tools/build_variant.sh wb_timers -DWB_PHASE_TIMERS with
NF_VARIANT_SRC=nf_wino_bf.hip): share of a wave's chunk time in window reads, record waits, hand-over + barrier and the rest.
usage: python tools/experimental/wino_phases.py build/variants/wb_timers.so"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'host_harness'))
from nerfool_amd import ops, _lib                       # noqa: E402
import standin                                          # noqa: E402

standin.use_library(sys.argv[1], emulated=False)
L = _lib.lib()
L.nf_wino_bf_phase_read.restype = ctypes.c_int
L.nf_wino_bf_phase_read.argtypes = [ctypes.c_void_p, ctypes.c_int]


def read(reset=True):
    buf = (ctypes.c_ulonglong * 16)()
    assert L.nf_wino_bf_phase_read(ctypes.addressof(buf), int(reset)) == 0
    return list(buf)


dev = torch.device('cuda', 0)
torch.manual_seed(0)
names = ['chunks', 'chunk', 'E reads', 'record waits', 'hand-over+barrier', '', 'timer', 'prologue', 'output']
for (ci, co, H, W) in ((64, 64, 189, 252), (128, 128, 95, 126), (256, 256, 48, 63), (256, 128, 96, 126)):
    x = torch.randn(4, ci, H + 2, W + 2, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    rf = ops.wino_pack(w, False, dev, None, 3)
    for _ in range(3):
        ops.conv3x3_wino(rf, x, co, 0, n_split=3)
    torch.cuda.synchronize()
    read(True)
    n = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.conv3x3_wino(rf, x, co, 0, n_split=3)
    e1.record()
    torch.cuda.synchronize()
    p = read(True)
    chunks = p[0]
    waves = chunks / ((ci + 15) // 16)
    tm = p[6] / chunks
    print('%3d->%3d %3dx%3d: %.1f us/launch; per wave: prologue %.0f, %d chunks x %.0f cycles, output %.0f  (timer pair %.0f cycles)'
          % (ci, co, H, W, e0.elapsed_time(e1) / n * 1e3, p[7] / waves, (ci + 15) // 16, p[1] / chunks, p[8] / waves, tm))
    nst = 8
    print('      per chunk: E reads %.0f | record waits %.0f (8 waits, timers %.0f) | hand-over + barrier %.0f | rest (split, products, issue) %.0f'
          % (p[2] / chunks - tm, p[3] / chunks - nst * tm, nst * tm, p[4] / chunks - tm,
             (p[1] - p[2] - p[3] - p[4]) / chunks))
