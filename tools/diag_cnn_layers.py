"""Diagnostic (GPU): where the feature CNN's backward-data pass departs from a float64 evaluation, convolution by
convolution.  The fused executor's gradient w.r.t. every convolution OUTPUT (and w.r.t. the input) is captured while the
tape unwinds and compared with autograd of the float64 oracle CNN on the CPU (same weights, same input, same upstream
gradient); the forward outputs are compared the same way.  Printed walking the network backward, so the first row whose
error jumps names the operation between it and the row above.

    python tools/diag_cnn_layers.py [tiny|medium] [repeat]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch

import parity_cases as pc
from nerfool_amd.ibrnet import feature_network as fn
from oracle import feature_net_ref as fnet

dev = 'cuda'


def rel(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    if a.shape != b.shape:
        return float('nan')
    return float((a - b).norm() / b.norm())


class TracedSlot:
    registry = []

    def __init__(self, v):
        self.v, self._g, self.last = v, None, None
        TracedSlot.registry.append(self)

    @property
    def g(self):
        return self._g

    @g.setter
    def g(self, val):
        self._g = val
        if val is not None:
            self.last = val

    def add(self, g):
        self.g = g if self._g is None else self._g + g


def oracle_trace(sd, x64, up64):
    names, outs = [], []
    orig = fnet._conv

    def traced(sd_, name, x, stride=1):
        y = orig(sd_, name, x, stride)
        y.retain_grad()
        names.append(name)
        outs.append(y)
        return y
    fnet._conv = traced
    try:
        c, f = fnet.resunet_forward(sd, x64)
    finally:
        fnet._conv = orig
    torch.autograd.backward([c, f], [up64[:, :32], up64[:, 32:]])
    return names, outs


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else 'tiny'
    repeat = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    g64, args, model, data, sampler, delta0, picks = pc.grad64_setup(case, dev)
    src = sampler.get_all()
    sd64 = {k: v.detach().cpu().double() for k, v in model.feature_net.state_dict().items()}
    x = (src['src_rgbs'] + delta0).squeeze(0).permute(0, 3, 1, 2)
    x64 = x.detach().cpu().double().requires_grad_(True)
    up = g64.t(case + '/dfm64', dev)
    names, outs64 = oracle_trace(sd64, x64, up.cpu().double())
    fn._Slot = TracedSlot
    for it in range(repeat):
        TracedSlot.registry = []
        xg = x.detach().clone().requires_grad_(True)
        fc, ff = model.feature_net(xg)
        gx, = torch.autograd.grad([fc, ff], xg, [up[:, :32].contiguous(), up[:, 32:].contiguous()])
        slots = TracedSlot.registry
        assert len(slots) == len(names) + 1, (len(slots), len(names))
        print('--- %s, pass %d: d input rel err %.3e' % (case, it, rel(gx, x64.grad)))
        choice = {k: v for k, v in fn._CONV_CHOICE.items()}
        for name, o64, s in reversed(list(zip(names, outs64, slots[1:]))):
            g = s.last
            if isinstance(g, tuple):
                g = torch.cat(list(g), 1)
            v = s.v
            bias = sd64.get(name + '.bias')
            o = o64 if bias is None or name == 'out_conv' else o64 - bias.view(1, -1, 1, 1)    # fused path drops biases in front of a norm
            shape = tuple(o64.shape)
            print('%-22s out %-18s fwd %.2e   d out %.2e' % (name, 'x'.join(map(str, shape)), rel(v, o), rel(g, o64.grad)))
        print('choices:', {str(k): v for k, v in choice.items()})


if __name__ == '__main__':
    main()
