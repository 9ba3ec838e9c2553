"""Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): ResUNet feature extractor, functional form.

ref: ibrnet/feature_network.py:154-268 (ResUNet, resnet34 encoder truncated after layer3, InstanceNorm with
affine parameters and no running statistics, reflect padding everywhere, bilinear x2 decoder with
align_corners=True, zero-padded skip tensors).  State-dict keys are the reference module paths so that a
reference checkpoint's `feature_net` entry can be replayed directly.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

_STAGES = (('layer1', 3, 64), ('layer2', 4, 128), ('layer3', 6, 256))


def _conv(sd, name, x, stride=1):
    w = sd[name + '.weight']
    b = sd.get(name + '.bias')
    pad = (w.shape[-1] - 1) // 2
    if pad:
        x = F.pad(x, (pad, pad, pad, pad), mode='reflect')
    return F.conv2d(x, w, b, stride=stride)


def _inorm(sd, name, x):
    return F.instance_norm(x, weight=sd[name + '.weight'], bias=sd[name + '.bias'], eps=1e-5)


class ReluTrace:
    """Checker hook for the ONE discontinuous operation of the network.  A ReLU whose argument is within rounding noise of
    zero can land on either side in two correct fp32 evaluations, and one flipped unit of an N-element activation changes the
    gradient by ~1/sqrt(N) of its norm.  With `masks` (bool tensors in call order: stem, then per block the inner and the
    outer ReLU) every ReLU becomes `x * mask`, i.e. the network is evaluated on the activation pattern of another evaluation;
    `pre` collects the arguments so that a test can list the flipped units and how close to zero they were."""

    def __init__(self, masks=None):
        self.masks = None if masks is None else list(masks)
        self.pre = []

    def relu(self, x):
        i = len(self.pre)
        self.pre.append(x.detach())
        if self.masks is None:
            return F.relu(x)
        return x * self.masks[i].to(device=x.device, dtype=x.dtype)


def _relu(x, trace):
    return F.relu(x) if trace is None else trace.relu(x)


def _basic_block(sd, prefix, x, stride, trace=None):
    """ref: ibrnet/feature_network.py:38-78."""
    y = _relu(_inorm(sd, prefix + '.bn1', _conv(sd, prefix + '.conv1', x, stride)), trace)
    y = _inorm(sd, prefix + '.bn2', _conv(sd, prefix + '.conv2', y))
    if (prefix + '.downsample.0.weight') in sd:
        x = _inorm(sd, prefix + '.downsample.1', _conv(sd, prefix + '.downsample.0', x, stride))
    return _relu(y + x, trace)


def _conv_in_elu(sd, prefix, x):
    """ref: ibrnet/feature_network.py:127-140 (`conv`: conv + InstanceNorm + ELU)."""
    return F.elu(_inorm(sd, prefix + '.bn', _conv(sd, prefix + '.conv', x)))


def _up(sd, prefix, x):
    """ref: ibrnet/feature_network.py:143-151."""
    x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
    return _conv_in_elu(sd, prefix + '.conv', x)


def _skip(small, big):
    """ref: ibrnet/feature_network.py:231-243: zero-pad the encoder tensor up to the decoder size, decoder first."""
    dy = big.shape[2] - small.shape[2]
    dx = big.shape[3] - small.shape[3]
    small = F.pad(small, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
    return torch.cat([big, small], dim=1)


def resunet_forward(sd, x, coarse_out_ch=32, fine_out_ch=32, trace=None):
    """x [V,3,H,W] -> (coarse [V,32,Hf,Wf], fine [V,32,Hf,Wf]).  ref: ibrnet/feature_network.py:245-268.
    trace: optional ReluTrace (checker hook, see there)."""
    x = _relu(_inorm(sd, 'bn1', _conv(sd, 'conv1', x, stride=2)), trace)
    feats = []
    for name, n_blocks, _ in _STAGES:
        for b in range(n_blocks):
            x = _basic_block(sd, '%s.%d' % (name, b), x, stride=2 if b == 0 else 1, trace=trace)
        feats.append(x)
    x1, x2, x3 = feats
    y = _conv_in_elu(sd, 'iconv3', _skip(x2, _up(sd, 'upconv3', x3)))
    y = _conv_in_elu(sd, 'iconv2', _skip(x1, _up(sd, 'upconv2', y)))
    out = _conv(sd, 'out_conv', y)
    return out[:, :coarse_out_ch], out[:, -fine_out_ch:]


def random_resunet_state(seed, coarse_out_ch=32, fine_out_ch=32, scale=1.0):
    """Seed-reproducible fixture weights (35.7 MB is too big for a fixture file): He-normal convolutions,
    InstanceNorm gamma ~ 1, beta ~ 0, keys in reference module order."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()

    def conv(name, cout, cin, k, bias):
        sd[name + '.weight'] = torch.randn(cout, cin, k, k, generator=g) * scale * (2.0 / (cin * k * k)) ** 0.5
        if bias:
            sd[name + '.bias'] = torch.randn(cout, generator=g) * 0.02

    def norm(name, c):
        sd[name + '.weight'] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[name + '.bias'] = 0.05 * torch.randn(c, generator=g)

    conv('conv1', 64, 3, 7, False)
    norm('bn1', 64)
    cin = 64
    for name, n_blocks, planes in _STAGES:
        for b in range(n_blocks):
            pre = '%s.%d' % (name, b)
            conv(pre + '.conv1', planes, cin if b == 0 else planes, 3, False)
            norm(pre + '.bn1', planes)
            conv(pre + '.conv2', planes, planes, 3, False)
            norm(pre + '.bn2', planes)
            if b == 0:
                conv(pre + '.downsample.0', planes, cin, 1, False)
                norm(pre + '.downsample.1', planes)
        cin = planes
    out_ch = coarse_out_ch + fine_out_ch
    conv('upconv3.conv.conv', 128, 256, 3, True)
    norm('upconv3.conv.bn', 128)
    conv('iconv3.conv', 128, 128 + 128, 3, True)
    norm('iconv3.bn', 128)
    conv('upconv2.conv.conv', 64, 128, 3, True)
    norm('upconv2.conv.bn', 64)
    conv('iconv2.conv', out_ch, 64 + 64, 3, True)
    norm('iconv2.bn', out_ch)
    conv('out_conv', out_ch, out_ch, 1, True)
    return sd
