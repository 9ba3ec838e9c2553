"""ResUNet feature extractor as a fused executor over hand-written kernels: Winograd F(2x2,3x3) for the stride-1
3x3 convolutions (csrc/nf_wino.hip), direct matrix-core convolutions for the four stride-2 ones (csrc/nf_conv_s2.hip), MFMA
GEMMs for the 1x1 ones (csrc/nf_conv1x1.hip), fused InstanceNorm / activation / padding / upsampling glue (csrc/nf_cnn.hip).
Module paths equal those of ibrnet/feature_network.py
(conv1, bn1, layer{1,2,3}.N.{conv1,bn1,conv2,bn2,downsample.{0,1}}, upconv{3,2}.conv.{conv,bn}, iconv{3,2}.{conv,bn},
out_conv) so that reference checkpoints load by key.

The output is produced channels-last ([V, Hf, Wf, 64] in memory): each feature-map pixel is one 256-byte record whose
halves are the coarse / fine 32-channel maps, which is what the gather kernels want (one cache line per bilinear tap)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops

# Test / diagnostic hook (tests/ and tools/ set it; nothing reads the environment).  'fused' (default): the executor below --
# hand-written convolutions with padding 0 + the InstanceNorm / activation / reflect-pad kernels of csrc/nf_cnn.hip in between,
# explicit backward-data only;  'torch': the plain nn.Module graph (ATen + autograd), what the fused executor is compared with
# and what the CPU tests of the host logic run (the stand-in emulates the fused glue slowly)
CNN_PATH = 'fused'


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
    def __init__(self, encoder='resnet34', coarse_out_ch=32, fine_out_ch=32, norm_layer=None, coarse_only=False,
                 single_net=False):
        super().__init__()
        if encoder != 'resnet34':
            raise ValueError('only the resnet34 encoder of the released IBRNet checkpoints is supported')
        # single_net (GNT flavour, gnt/feature_network.py:195-199,314-316): one map of coarse_out_ch channels returned twice
        self.single_net = single_net
        self.coarse_only = coarse_only or single_net
        self.coarse_out_ch = coarse_out_ch
        self.fine_out_ch = 0 if self.coarse_only else fine_out_ch
        out_ch = self.coarse_out_ch + self.fine_out_ch
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False, padding_mode='reflect')
        self.bn1 = _inorm(64)
        self.layer1, self.layer2, self.layer3 = _stage(64, 64, 3), _stage(64, 128, 4), _stage(128, 256, 6)
        self.upconv3 = _Up(256, 128)
        self.iconv3 = _ConvNormELU(128 + 128, 128, 3)
        self.upconv2 = _Up(128, 64)
        self.iconv2 = _ConvNormELU(64 + 64, out_ch, 3)
        self.out_conv = nn.Conv2d(out_ch, out_ch, 1, 1)
        # operand form of the stride-1 3x3 convolutions' Winograd products: None = the module default WINO_OPERANDS (fp32-grade).
        # DIAGNOSTIC ONLY: nothing in the package sets it -- plain 'bf16' operands were measured for BASELINE config 5 and REJECTED
        # for the CNN (d loss / d delta cosine 0.78 to the reference's, profiles/r04_plain_bf16_cnn_rejected.txt); config 5's bf16
        # path is the IBRNet row network only.  tools/diag_bf16_cnn.py sets it to reproduce that measurement.
        self.conv_precision = None

    def describe_output(self, H, W):
        """(channels of the distinct maps, same-map-twice, second-is-None, Hf, Wf) of forward() on [*, 3, H, W]: every
        stride-2 stage gives ceil(n / 2), the decoder doubles twice (feature_network.py:245-268)."""
        def size(n):
            for _ in range(4):
                n = (n + 1) // 2
            return 4 * n
        if self.single_net:
            return [self.coarse_out_ch], True, False, size(H), size(W)
        if self.coarse_only:
            return [self.coarse_out_ch], False, True, size(H), size(W)
        return [self.coarse_out_ch, self.fine_out_ch], False, False, size(H), size(W)

    def _fused(self, x):
        # the fused executor computes no weight gradients: frozen parameters only (checked on every call: ~70 us of host time
        # next to a multi-millisecond network, and a partly unfrozen network must not pass silently)
        self._frozen = not any(p.requires_grad for p in self.parameters())
        if CNN_PATH != 'fused':
            return False                # the plain nn.Module graph, an explicit choice (tests compare against it)
        if not (x.is_cuda or ops._lib.emulated()):
            raise RuntimeError('ResUNet: input on %s -- the feature CNN runs on the GPU only (no CPU fallback)' % x.device)
        if not self._frozen:
            raise NotImplementedError('ResUNet: the fused executor differentiates w.r.t. its input only; freeze the weights '
                                      '(requires_grad_(False), as IBRNetModel does)')
        return True

    def _module_graph(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x1 = self.layer1(x)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        y = self.iconv3(_join(x2, self.upconv3(x3)))
        y = self.iconv2(_join(x1, self.upconv2(y)))
        return self.out_conv(y).contiguous(memory_format=torch.channels_last)

    def forward_full(self, x):
        """out_conv's whole output [N, coarse + fine channels, Hf, Wf] (channels-last), i.e. forward() before the channel
        split -- what the view-sharded attack step exchanges between the ranks (eval_adv.RayShard)."""
        if self._fused(x):
            return _FusedResUNet.apply(x, self, (self.coarse_out_ch + self.fine_out_ch,), _taped(x))[0]
        return self._module_graph(x)

    def forward(self, x):
        if self._fused(x):
            if self.single_net or self.coarse_only:
                out, = _FusedResUNet.apply(x, self, (self.coarse_out_ch,), _taped(x))
                return (out, out) if self.single_net else (out, None)
            return _FusedResUNet.apply(x, self, (self.coarse_out_ch, self.fine_out_ch), _taped(x))
        out = self._module_graph(x)
        if self.single_net:
            return out, out
        if self.coarse_only:
            return out, None
        return out[:, :self.coarse_out_ch], out[:, -self.fine_out_ch:]


# ----------------------------------------------------------------------------------------------------------------------
# Fused executor: same network, same parameters, explicit forward tape + backward-data pass.
# Per convolution the module graph launches reflection_pad2d, conv, batch_norm (instance norm), add, relu (and their five
# backward kernels); here it is ONE hand-written convolution on a pre-padded input (padding 0: csrc/nf_wino_bf.hip, nf_conv_s2.hip,
# nf_conv1x1.hip -- no vendor library on this path) + ONE fused kernel per direction (csrc/nf_cnn.hip).
# Convolution biases in front of an InstanceNorm are dropped: the norm subtracts the plane mean, so they cancel exactly.
# ----------------------------------------------------------------------------------------------------------------------

class _Slot:
    """a plain tensor with a gradient accumulator"""
    __slots__ = ('v', 'g')

    def __init__(self, v):
        self.v, self.g = v, None

    def add(self, g):
        self.g = g if self.g is None else self.g + g


class _Act:
    """an activation written by the fused kernel: padded storage, gradient w.r.t. the padded tensor (from the next
    convolution) and w.r.t. its interior (residual / skip / upsample consumers)"""
    __slots__ = ('yp', 'pad', 'gp', 'gi', 'gs')

    def __init__(self, yp, pad):
        self.yp, self.pad, self.gp, self.gi, self.gs = yp, pad, None, None, None

    def interior(self):
        p = self.pad
        return self.yp if p == 0 else self.yp[:, :, p:-p, p:-p]

    def add_p(self, g):
        self.gp = g if self.gp is None else self.gp + g

    def add_i(self, g):
        self.gi = g if self.gi is None else self.gi + g

    def add_s(self, g):          # gradient of a stride-2 consumer of the interior (1x1 downsample): even rows / columns
        self.gs = g if self.gs is None else self.gs + g


# 3x3 stride-1 convolutions: the Winograd matrix-core kernel (csrc/nf_wino.hip) runs with 64 or with 32 output channels per
# workgroup -- both forms compute every output with the same arithmetic in the same order, so the width does not change a single bit
# of the result.  'auto' picks it by a RULE ON THE SHAPE (round 4; until round 3 a one-off timing per process, which made the trace,
# counter and bench runs of one commit use different configurations): 32 exactly when the 64-wide grid would not fill one round of
# the chip's 512 resident workgroups (two per CU) -- there the narrower form doubles the workgroups (256 -> 256 at 48 x 63: 79 -> 69
# us); everywhere else the 64-wide form is faster or equal (tools/bench_conv3x3.py: 128 -> 128 at 95 x 126 67 vs 71 us, 256 -> 128
# at 96 x 126 119 vs 136 us).  'wino' / 'wino32' force one (test / diagnostic hook).
CONV3X3 = 'auto'
_CONV_CHOICE = {}          # (direction, c_in, c_out, input shape) -> chosen form: what bench.py reports


def _wino_width(c_out, n_img, h_out, w_out, n_split=0):
    """output channels per workgroup for a Winograd launch producing [n_img, c_out, h_out, w_out] (8 x 16 output blocks).  The
    bf16-split kernels always take 64: their 32-wide form holds too many registers for a third workgroup per CU, which is what
    made 32 pay on thin grids (256 -> 256 at 48 x 63, bf16x3: 63 us at 64 against 75 us at 32)."""
    if c_out <= 64 or n_split:
        return ops.wino_group(c_out)
    blocks64 = n_img * (-(-h_out // 8)) * (-(-w_out // 16)) * (c_out // 64)
    return 32 if blocks64 < 512 else 64


def _pick(key, candidates, rule):
    """which implementation runs this convolution (candidates: name -> thunk); rule = the name the shape rule selects"""
    if CONV3X3 in candidates:
        name = CONV3X3
    elif len(candidates) == 1:
        name = next(iter(candidates))
    else:
        name = rule
    _CONV_CHOICE[key] = name
    return name


# Operand form of the Winograd products (csrc/nf_wino.hip / nf_wino_bf.hip): 'fp32' = v_mfma_f32_32x32x2_f32; 'bf16x3' = every fp32 operand
# as three bf16 parts on v_mfma_f32_32x32x16_bf16, the six cross terms of order <= 2^-16 (dropped: <= 2^-24 of a product, fp32 rounding
# level -- same parity bars as 'fp32'); 'bf16' = plain bf16 operands (BASELINE config 5's opt-in precision, never the default).
# WINO_OPERANDS is the default for networks that do not ask for 'bf16' (ResUNet.conv_precision); test / diagnostic hook.
WINO_OPERANDS = 'bf16x3'
# Operand form of the BACKWARD-DATA passes when the forward runs 'bf16x3' (round 5): 'bf16x2' = two bf16 parts per operand (16
# significant bits), the three cross terms of order <= 2^-8 -- half the matrix instructions of 'bf16x3'.  The gradient has no ReLU
# decision to flip (the masks are the forward's) and d loss / d delta stays inside its 1e-3 bar against float64 (measured in
# profiles/r05_parity_numbers.txt); 'bf16x3' keeps the backward at the forward's precision.  Test / diagnostic hook like WINO_OPERANDS.
WINO_BWD_OPERANDS = 'bf16x2'
_N_SPLIT = {'fp32': 0, 'bf16x3': 3, 'bf16x2': 2, 'bf16': 1}


def _wino_records(conv_w, k_per_group, n_split=0, backward=False):
    """forward (or backward-data) records of a weight for one workgroup width and operand form; packed on first use.  The records hang
    on the weight tensor itself (like conv._nf_records of the 1x1 convolutions), so they die with it and can never be taken for
    another weight's; they are re-packed when the weight's storage, version or device changed."""
    key = (conv_w.data_ptr(), conv_w._version, str(conv_w.device))
    store = getattr(conv_w, '_nf_wino', None)
    if store is None:
        store = conv_w._nf_wino = {}
    slot = (k_per_group, n_split, bool(backward))
    cache = store.get(slot)
    if cache is None or cache[0] != key:
        cache = (key, ops.wino_pack(conv_w, bool(backward), conv_w.device, k_per_group, n_split))
        store[slot] = cache
    return cache[1]


def _wino_ring_records(conv_w):
    """1-D border weights of the backward-data pass (nf_conv3x3_bwd_ring), kept on the weight tensor"""
    key = (conv_w.data_ptr(), conv_w._version, str(conv_w.device))
    cache = getattr(conv_w, '_nf_wino_ring', None)
    if cache is None or cache[0] != key:
        cache = (key, ops.wino_ring_pack(conv_w, conv_w.device))
        conv_w._nf_wino_ring = cache
    return cache[1]


def _conv3x3(tape, inp, w, sink, operands=None):
    """3x3 stride-1 convolution on a pre-padded activation (padding 0) and its backward-data pass; operands: 'fp32' / 'bf16x3' / 'bf16'
    (None = WINO_OPERANDS)"""
    c_out, c_in = w.shape[0], w.shape[1]
    N, Hi, Wi = inp.shape[0], inp.shape[2], inp.shape[3]
    ns = _N_SPLIT[operands or WINO_OPERANDS]
    ns_b = _N_SPLIT[WINO_BWD_OPERANDS] if ns == 3 else ns         # the backward-data pass of a 'bf16x3' forward
    fwd = {'wino': lambda: ops.conv3x3_wino(_wino_records(w, ops.wino_group(c_out), ns), inp, c_out, 0, k_per_group=ops.wino_group(c_out),
                                            n_split=ns)}
    if c_out > 64:                      # narrower workgroups only matter when they add workgroups to a thin grid
        fwd['wino32'] = lambda: ops.conv3x3_wino(_wino_records(w, 32, ns), inp, c_out, 0, k_per_group=32, n_split=ns)
    rule = 'wino32' if _wino_width(c_out, N, Hi - 2, Wi - 2, ns) == 32 and c_out > 64 else 'wino'
    out = _Slot(fwd[_pick(('f', c_in, c_out) + tuple(inp.shape), fwd, rule)]())

    def bwd():
        g_out = out.g
        # planes whose (H + 2) x (W + 2) gradient needs a round of 8 x 16 output blocks more than the H x W plane itself (48 x 63 at 4
        # images: 560 workgroups against 384): the Winograd kernel on the interior-aligned region + the 1-D border ring kernel (a rule on
        # the launch's shape, ops.wino_bwd_split_pays, not a timing: the two forms round differently)
        plan = None
        # (the rounds estimate uses the workgroup width the shape rule gives the one-launch form of this pass)
        kpg_est = _wino_width(c_in, N, g_out.shape[2] + 2, g_out.shape[3] + 2, ns_b) if c_in > 64 else ops.wino_group(c_in)
        if g_out.shape[1] <= ops.WINO_RING_MAX_CHANNELS and ops.wino_bwd_split_pays(g_out.shape[2], g_out.shape[3], g_out.shape[0],
                                                                                   g_out.shape[1], c_in, k_per_group=kpg_est):
            plan = ops.wino_bwd_split_plan(g_out.shape[2], g_out.shape[3])

        def bwd_data(kpg):
            rec = _wino_records(w, kpg, ns_b, backward=True)
            if plan is None:
                return ops.conv3x3_wino(rec, g_out, c_in, 2, k_per_group=kpg, n_split=ns_b)
            return ops.conv3x3_wino_bwd_split(rec, _wino_ring_records(w), g_out, c_in, plan, k_per_group=kpg, n_split=ns_b)
        cand = {'wino': lambda: bwd_data(ops.wino_group(c_in))}
        if c_in > 64:
            cand['wino32'] = lambda: bwd_data(32)
        ho, wo = (g_out.shape[2] + 2, g_out.shape[3] + 2) if plan is None else (plan[0], plan[1])
        rule_b = 'wino32' if _wino_width(c_in, N, ho, wo, ns_b) == 32 and c_in > 64 else 'wino'
        sink(cand[_pick(('b', c_in, c_out) + tuple(inp.shape), cand, rule_b)]())
        out.g = None
    tape.append(bwd)
    return out


def _s2_records(conv_w):
    """(forward, backward-data) records of a stride-2 convolution weight for csrc/nf_conv_s2.hip, kept on the weight tensor"""
    key = (conv_w.data_ptr(), conv_w._version, str(conv_w.device))
    cache = getattr(conv_w, '_nf_s2', None)
    if cache is None or cache[0] != key:
        cache = (key, ops.conv_s2_pack(conv_w, False, conv_w.device), ops.conv_s2_pack(conv_w, True, conv_w.device))
        conv_w._nf_s2 = cache
    return cache[1], cache[2]


# stride-2 convolutions (7x7 stem, first 3x3 of layer1-3): csrc/nf_conv_s2.hip


def _s2_records_x3(conv_w):
    """bf16x3 (forward, backward-data) records of a stride-2 3x3 weight (csrc/nf_conv_s2.hip: k_conv_s2_fwd3_x3 / k_conv_s2_bwd3_x3),
    kept on the weight tensor"""
    key = (conv_w.data_ptr(), conv_w._version, str(conv_w.device))
    cache = getattr(conv_w, '_nf_s2_x3', None)
    if cache is None or cache[0] != key:
        cache = (key, ops.conv_s2_pack_x3(conv_w, False, conv_w.device), ops.conv_s2_pack_x3(conv_w, True, conv_w.device))
        conv_w._nf_s2_x3 = cache
    return cache[1], cache[2]


def _stem_records_x3(conv_w):
    """bf16x3 forward records of the 7x7 stem weight (csrc/nf_conv_s2.hip: k_conv_s2_stem_fwd_x3), kept on the weight tensor"""
    key = (conv_w.data_ptr(), conv_w._version, str(conv_w.device))
    cache = getattr(conv_w, '_nf_stem_x3', None)
    if cache is None or cache[0] != key:
        cache = (key, ops.conv_s2_stem_pack_x3(conv_w, conv_w.device))
        conv_w._nf_stem_x3 = cache
    return cache[1]


def _conv_s2(tape, inp, w, sink, operands=None):
    c_out, c_in, ks = w.shape[0], w.shape[1], w.shape[2]
    x3 = ks == 3 and (operands or WINO_OPERANDS) == 'bf16x3'      # the 3x3 passes on the split operands (fp32-grade)
    stem_x3 = ks == 7 and c_in <= 3 and (operands or WINO_OPERANDS) == 'bf16x3'      # the stem's forward likewise (backward: fp32 operands)
    if x3:
        rf, rb = _s2_records_x3(w)
        out = _Slot(ops.conv_s2_fwd_x3(rf, inp, c_out))
    elif stem_x3:
        rb = _s2_records(w)[1]
        out = _Slot(ops.conv_s2_stem_fwd_x3(_stem_records_x3(w), inp, c_out))
    else:
        rf, rb = _s2_records(w)
        out = _Slot(ops.conv_s2_fwd(rf, inp, c_out, ks))
    Hi, Wi = inp.shape[2], inp.shape[3]

    def bwd():
        if x3:
            sink(ops.conv_s2_bwd_x3(rb, out.g, c_in, Hi, Wi))
        else:
            sink(ops.conv_s2_bwd(rb, out.g, c_in, ks, Hi, Wi))
        out.g = None
    tape.append(bwd)
    return out


def _conv(tape, inp, w, stride, sink, bias=None, operands=None):
    if stride == 1 and bias is None and tuple(w.shape[2:]) == (3, 3) and w.shape[0] % 32 == 0:
        return _conv3x3(tape, inp, w, sink, operands)
    if stride == 2 and bias is None and tuple(w.shape[2:]) in ((3, 3), (7, 7)) and (w.shape[2] == 3 or w.shape[1] <= 3):
        return _conv_s2(tape, inp, w, sink, operands)
    # no vendor-library fallback on the product path: the reference network (and every checkpoint that loads into it) has only the
    # shapes above -- anything else is a different network and must say so
    raise NotImplementedError('ResUNet fused executor: no hand-written kernel for a %dx%d stride-%d convolution %d -> %d%s '
                              '(3x3 stride-1 needs C_out %% 32 == 0; stride-2 is 3x3, or 7x7 on <= 3 input channels); '
                              "feature_network.CNN_PATH = 'torch' runs the plain nn.Module graph for comparison"
                              % (w.shape[2], w.shape[3], stride, w.shape[1], w.shape[0], ' with bias' if bias is not None else ''))


def _conv1x1(tape, inp, conv, sink, channels_last_out=False):
    """1x1 convolution (downsample branch, out_conv) as an MFMA GEMM over the pixels (csrc/nf_conv1x1.hip): `inp` may be any
    strided view -- the stride-2 subsampling is read in place -- and out_conv writes the channels-last maps directly."""
    w = conv.weight
    key = (w.data_ptr(), w._version, str(w.device))
    cache = getattr(conv, '_nf_records', None)
    if cache is None or cache[0] != key:
        cache = (key, ops.conv1x1_pack(w, False, w.device), ops.conv1x1_pack(w, True, w.device))
        conv._nf_records = cache
    c_out, c_in = w.shape[0], w.shape[1]
    out = _Slot(ops.conv1x1(cache[1], conv.bias, inp, c_out, channels_last_out))

    def bwd():
        if isinstance(out.g, tuple):          # out_conv: one gradient per feature map, read where they lie
            g0, g1 = out.g
            sink(ops.conv1x1(cache[2], None, g0, c_in, x2=g1))
        else:
            sink(ops.conv1x1(cache[2], None, out.g, c_in))
        out.g = None
    tape.append(bwd)
    return out


# Checker hook (tests only): a list here collects the interior of every ReLU output in network order, from which a test
# reads the activation pattern this evaluation used (the backward takes ReLU' from the same stored output / the same
# arithmetic).  The ReLU is the network's one discontinuity: see oracle/feature_net_ref.ReluTrace.
TRACE_RELU = None


def _fuse(tape, xs, norm, res, act, pad):
    gamma, beta = (norm.weight, norm.bias) if norm is not None else (None, None)
    yp, mean, rstd = ops.in_act_pad_fwd(xs.v, gamma, beta, None if res is None else res.interior(), act, pad,
                                        eps=norm.eps if norm is not None else 0.0)
    a = _Act(yp, pad)
    if TRACE_RELU is not None and act == ops.ACT_RELU:
        TRACE_RELU.append(a.interior())

    def bwd():
        dx, d_res = ops.in_act_pad_bwd(a.gp, a.gi, yp, xs.v if norm is not None else None, gamma, mean, rstd, act, pad,
                                       res is not None, beta=beta, d_extra_sub=a.gs)
        xs.add(dx)
        if res is not None:
            res.add_i(d_res)
        a.gp = a.gi = a.gs = None
    tape.append(bwd)
    return a


def _resblock(tape, blk, xin, operands=None):
    """xin: _Act padded by 1.  conv3x3(s) -> IN -> ReLU -> conv3x3 -> IN -> (+ identity | 1x1 conv + IN) -> ReLU."""
    stride = blk.conv1.stride[0]
    t1 = _conv(tape, xin.yp, blk.conv1.weight, stride, xin.add_p, operands=operands)
    a1 = _fuse(tape, t1, blk.bn1, None, ops.ACT_RELU, 1)
    t2 = _conv(tape, a1.yp, blk.conv2.weight, 1, a1.add_p, operands=operands)
    if blk.downsample is not None:
        # 1x1 stride-s convolution == 1x1 stride-1 convolution of the subsampled activation: a quarter of the bytes to gather,
        # and the gradient comes back subsampled (the fused backward adds it at the even positions)
        if stride == 2:
            d = _conv1x1(tape, xin.interior()[:, :, ::2, ::2], blk.downsample[0], xin.add_s)
        else:
            d = _conv1x1(tape, xin.interior(), blk.downsample[0], xin.add_i)
        res = _fuse(tape, d, blk.downsample[1], None, ops.ACT_NONE, 0)
    else:
        res = xin
    return _fuse(tape, t2, blk.bn2, res, ops.ACT_RELU, 1)


def _upsample_pad(tape, a, pad):
    """F.interpolate(x2, bilinear, align_corners) + the reflect padding of the following convolution in one kernel; the
    backward folds the padded gradient and applies the transposed interpolation in one kernel (nf_upsample2x_pad_bwd)."""
    src = a.interior()
    up = _Act(ops.upsample2x_pad_fwd(src, pad), pad)
    in_size = list(src.shape)

    def bwd():
        # the only consumer of the upsampled tensor is the convolution (feature_network.py:143-151): fold + transposed
        # interpolation in one kernel; an interior consumer would be another network
        assert up.gi is None, 'upsampled activation has a second consumer: not the ResUNet graph'
        a.add_i(ops.upsample2x_pad_bwd(up.gp, in_size[2], in_size[3], pad))
        up.gp = None
    tape.append(bwd)
    return up


def _join_pad(tape, enc, t, norm, pad):
    """reflect_pad(cat([ELU(IN(t)), zero-pad(enc)], dim=1), pad) written slice by slice into one buffer (no torch.cat, no copy
    of the decoder half: its norm + activation kernel writes straight into its channel slice), and the backward read slice by
    slice out of the convolution's input gradient (the decoder half's fold happens inside its norm backward).
    t: _Slot of the decoder convolution's output, enc: _Act."""
    e = enc.interior()
    N, cd, H, W = t.v.shape
    ce, eh, ew = e.shape[1], e.shape[2], e.shape[3]
    top, left = (H - eh) // 2, (W - ew) // 2
    buf = torch.empty(N, cd + ce, H + 2 * pad, W + 2 * pad, dtype=t.v.dtype, device=t.v.device)
    _, mean, rstd = ops.in_act_pad_fwd(t.v, norm.weight, norm.bias, None, ops.ACT_ELU, pad, eps=norm.eps, out=buf, c_off=0)
    # the encoder tensor is read where it lies (interior view of its padded activation) and lands zero-extended to H x W at
    # (top, left), reflect-padded, in its channel slice of the buffer: no F.pad copy, no separate padding pass
    ops.pad_gather_fwd(e, H, W, pad, top, left, out=buf[:, cd:])
    out = _Act(buf, pad)

    def bwd():
        g = out.gp
        dx, _ = ops.in_act_pad_bwd(g[:, :cd], None, None, t.v, norm.weight, mean, rstd, ops.ACT_ELU, pad, False, beta=norm.bias,
                                   shape=(N, cd, H, W))
        t.add(dx)
        enc.add_i(ops.pad_gather_bwd(g[:, cd:], H, W, pad, eh, ew, top, left))
        out.gp = None
    tape.append(bwd)
    return out


class _NoTape:
    """tape of an evaluation nobody will differentiate (torch.no_grad(), or an input without requires_grad): the backward
    closures are dropped as they are made, so no activation outlives its consumer (full-image evaluation)"""

    def append(self, step):
        pass


def fused_forward(net, x, need_grad=True):
    """x [V,3,H,W] -> (out [V,64,Hf,Wf] NCHW, tape, input slot)."""
    tape = [] if need_grad else _NoTape()
    operands = getattr(net, 'conv_precision', None)      # None = WINO_OPERANDS; 'bf16' = BASELINE config 5's opt-in
    # the input image is read in whatever layout it comes (the attack hands over a permuted view of the channels-last
    # src + delta) and leaves reflect-padded for the 7x7 stem; its gradient is written back in the same layout
    xin = _Slot(x)
    H0, W0 = x.shape[2], x.shape[3]
    a_in = a = _Act(ops.pad_gather_fwd(x, H0, W0, 3), 3)

    def input_bwd():
        xin.g = ops.pad_gather_bwd(a_in.gp, H0, W0, 3, H0, W0, like=x)
        a_in.gp = None
    tape.append(input_bwd)
    t = _conv(tape, a.yp, net.conv1.weight, 2, a.add_p)
    a = _fuse(tape, t, net.bn1, None, ops.ACT_RELU, 1)
    feats = []
    for layer in (net.layer1, net.layer2, net.layer3):
        for blk in layer:
            a = _resblock(tape, blk, a, operands)
        feats.append(a)
    x1, x2, x3 = feats

    def decoder_stage(src, up, iconv, enc):
        up_p = _upsample_pad(tape, src, 1)
        t = _conv(tape, up_p.yp, up.conv.conv.weight, 1, up_p.add_p, operands=operands)
        jp = _join_pad(tape, enc, t, up.conv.bn, 1)
        t = _conv(tape, jp.yp, iconv.conv.weight, 1, jp.add_p, operands=operands)
        return _fuse(tape, t, iconv.bn, None, ops.ACT_ELU, 0)

    y = decoder_stage(x3, net.upconv3, net.iconv3, x2)
    y = decoder_stage(y, net.upconv2, net.iconv2, x1)
    out = _conv1x1(tape, y.yp, net.out_conv, y.add_p, channels_last_out=True)
    return out, tape, xin


def _taped(x):
    """will anybody differentiate this evaluation?  (ctx.needs_input_grad ignores the grad mode, and inside Function.forward the
    grad mode is always off: the caller decides)"""
    return torch.is_grad_enabled() and x.requires_grad


class _FusedResUNet(torch.autograd.Function):
    """outputs: the feature maps as channel slices of out_conv's ONE channels-last buffer, split INSIDE the function -- the
    gradients then arrive one per map and out_conv's backward reads them in place (nf_conv1x1's second source) instead of
    autograd assembling them with zero fills, two strided copies and an add (0.19 ms per step at BASELINE config 2)."""

    @staticmethod
    def forward(ctx, x, net, split, need_grad):
        out, tape, xin = fused_forward(net, x, need_grad=need_grad)
        ctx.tape, ctx.out, ctx.xin = tape, out, xin
        full = out.v.contiguous(memory_format=torch.channels_last)
        if len(split) == 1:
            return (full,)
        return tuple(full.split(list(split), dim=1))

    @staticmethod
    def backward(ctx, *d_outs):
        if ctx.tape is None:
            raise RuntimeError('ResUNet (fused executor): the backward tape is consumed by the first backward pass and its '
                               'activations are released; call the network again instead of backward(retain_graph=True) twice')
        ref = next(g for g in d_outs if g is not None)
        d_outs = [torch.zeros_like(ref) if g is None else g for g in d_outs]
        ctx.out.g = d_outs[0] if len(d_outs) == 1 else tuple(d_outs)     # any layout: the 1x1 backward addresses it by strides
        for step in reversed(ctx.tape):
            step()
        g = ctx.xin.g
        ctx.tape = ctx.out = ctx.xin = None
        return g, None, None, None
