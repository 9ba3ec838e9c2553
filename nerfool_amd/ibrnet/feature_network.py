"""ResUNet feature extractor on PyTorch-ROCm (MIOpen convolutions) -- SURVEY a14: delegated to the vendor library in
this round; the ray path behind it is the hand-written part.  Module paths equal those of ibrnet/feature_network.py
(conv1, bn1, layer{1,2,3}.N.{conv1,bn1,conv2,bn2,downsample.{0,1}}, upconv{3,2}.conv.{conv,bn}, iconv{3,2}.{conv,bn},
out_conv) so that reference checkpoints load by key.

The output is produced channels-last ([V, Hf, Wf, 64] in memory): each feature-map pixel is one 256-byte record whose
halves are the coarse / fine 32-channel maps, which is what the gather kernels want (one cache line per bilinear tap)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _c3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False, padding_mode='reflect')


def _c1(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 1, stride=stride, bias=False)


def _inorm(c):
    return nn.InstanceNorm2d(c, track_running_stats=False, affine=True)


class _ResBlock(nn.Module):
    """Two 3x3 convolutions with instance norm, identity (or 1x1-projected) shortcut."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1, self.bn1 = _c3(cin, cout, stride), _inorm(cout)
        self.conv2, self.bn2 = _c3(cout, cout), _inorm(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(_c1(cin, cout, stride), _inorm(cout))

    def forward(self, x):
        y = self.bn2(self.conv2(F.relu(self.bn1(self.conv1(x)))))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class _ConvNormELU(nn.Module):
    def __init__(self, cin, cout, k, stride=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=(k - 1) // 2, padding_mode='reflect')
        self.bn = _inorm(cout)

    def forward(self, x):
        return F.elu(self.bn(self.conv(x)))


class _Up(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _ConvNormELU(cin, cout, 3)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True))


def _stage(cin, cout, n):
    return nn.Sequential(*[_ResBlock(cin if i == 0 else cout, cout, 2 if i == 0 else 1) for i in range(n)])


def _join(enc, dec):
    """Zero-pad the encoder tensor to the decoder's size and stack it behind the decoder channels."""
    dy, dx = dec.shape[2] - enc.shape[2], dec.shape[3] - enc.shape[3]
    return torch.cat([dec, F.pad(enc, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))], dim=1)


class ResUNet(nn.Module):
    def __init__(self, encoder='resnet34', coarse_out_ch=32, fine_out_ch=32, norm_layer=None, coarse_only=False):
        super().__init__()
        if encoder != 'resnet34':
            raise ValueError('only the resnet34 encoder of the released IBRNet checkpoints is supported')
        self.coarse_only = coarse_only
        self.coarse_out_ch = coarse_out_ch
        self.fine_out_ch = 0 if coarse_only else fine_out_ch
        out_ch = self.coarse_out_ch + self.fine_out_ch
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False, padding_mode='reflect')
        self.bn1 = _inorm(64)
        self.layer1, self.layer2, self.layer3 = _stage(64, 64, 3), _stage(64, 128, 4), _stage(128, 256, 6)
        self.upconv3 = _Up(256, 128)
        self.iconv3 = _ConvNormELU(128 + 128, 128, 3)
        self.upconv2 = _Up(128, 64)
        self.iconv2 = _ConvNormELU(64 + 64, out_ch, 3)
        self.out_conv = nn.Conv2d(out_ch, out_ch, 1, 1)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x1 = self.layer1(x)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        y = self.iconv3(_join(x2, self.upconv3(x3)))
        y = self.iconv2(_join(x1, self.upconv2(y)))
        out = self.out_conv(y).contiguous(memory_format=torch.channels_last)
        if self.coarse_only:
            return out, None
        return out[:, :self.coarse_out_ch], out[:, -self.fine_out_ch:]
