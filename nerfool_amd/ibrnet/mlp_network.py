"""IBRNet per-sample network as an nn.Module whose forward AND backward are the HIP kernels.

The module tree reproduces the parameter names of ibrnet/mlp_network.py:152-208 so that the public IBRNet
checkpoints (`net_coarse` / `net_fine` state-dicts) load by key; `pos_encoding` is rebuilt from n_samples because
checkpoints may lack it (ibrnet/model.py:148-150).  The parameters are treated as constants of the attack: the
backward produces d/d(rgb_feat) only (the reference accumulates weight gradients and never reads them)."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops


def _mlp(dims, final_act=True):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2 or final_act:
            layers.append(nn.ELU())        # placeholders keep the Sequential indices of the reference (0, 2, 4)
    return nn.Sequential(*layers)


class _RayAttentionParams(nn.Module):
    """Parameter holder named like MultiHeadAttention (ibrnet/mlp_network.py:69-88)."""

    def __init__(self, n_head=4, d_model=16, d_k=4, d_v=4):
        super().__init__()
        self.w_qs = nn.Linear(d_model, n_head * d_k, bias=False)
        self.w_ks = nn.Linear(d_model, n_head * d_k, bias=False)
        self.w_vs = nn.Linear(d_model, n_head * d_v, bias=False)
        self.fc = nn.Linear(n_head * d_v, d_model, bias=False)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)


def sinusoid_table(n_samples, d_hid=16):
    pos = np.arange(n_samples, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    ang = pos / np.power(10000.0, 2 * (j // 2) / d_hid)
    return torch.from_numpy(np.where(j % 2 == 0, np.sin(ang), np.cos(ang))).float().unsqueeze(0)


# Test / diagnostic hook (tests/ and tools/ set it; nothing reads the environment): 'auto' = matrix-core kernels whenever the shape
# allows (any V <= 32, S <= 256), 'generic' = shape-generic kernels only
KERNEL_PATH = 'auto'


class _IBRNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb_feat, ray_diff, mask, blob, mfma_blob, pos_enc, anti_alias, bf16_blob=None):
        S, V = rgb_feat.shape[1], rgb_feat.shape[2]
        ctx.mfma = KERNEL_PATH != 'generic' and mfma_blob is not None and ops.ibrnet_mfma_supported(S, V)
        if bf16_blob is not None and not ctx.mfma:
            raise RuntimeError('the bf16 IBRNet path needs the matrix-core kernels (V <= 32; got V=%d)' % V)
        ctx.bf16 = bf16_blob is not None
        if ctx.mfma:
            raw, smp = ops.ibrnet_fwd_mfma(mfma_blob, blob, pos_enc, rgb_feat, ray_diff, mask, anti_alias, bf16_blob=bf16_blob)
            ctx.save_for_backward(rgb_feat, ray_diff, mask, blob, pos_enc, mfma_blob, smp, *((bf16_blob,) if ctx.bf16 else ()))
        else:
            raw = ops.ibrnet_fwd(blob, pos_enc, rgb_feat, ray_diff, mask, anti_alias)
            ctx.save_for_backward(rgb_feat, ray_diff, mask, blob, pos_enc)
        ctx.anti_alias = anti_alias
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        if ctx.mfma:
            rgb_feat, ray_diff, mask, blob, pos_enc, mfma_blob, smp = ctx.saved_tensors[:7]
            d_rgb_feat = ops.ibrnet_bwd_mfma(mfma_blob, blob, pos_enc, rgb_feat, ray_diff, mask, smp, d_raw, ctx.anti_alias,
                                             bf16_blob=ctx.saved_tensors[7] if ctx.bf16 else None)
        else:
            rgb_feat, ray_diff, mask, blob, pos_enc = ctx.saved_tensors
            d_rgb_feat = ops.ibrnet_bwd(blob, pos_enc, rgb_feat, ray_diff, mask, d_raw, ctx.anti_alias)
        return d_rgb_feat, None, None, None, None, None, None, None


class _IBRNetGatherFunction(torch.autograd.Function):
    """_IBRNetFunction differentiated straight through to the feature maps: the backward scatters d rgb_feat into d featmaps
    inside the row kernel (ops.ibrnet_bwd_mfma_scatter).  rgb_feat comes in detached -- Projector.compute produced it from
    `featmaps` with the sample points / cameras handed over next to it."""

    @staticmethod
    def forward(ctx, featmaps, rgb_feat, ray_diff, mask, blob, mfma_blob, pos_enc, anti_alias, pts, cam_ws, bf16_blob=None):
        raw, smp = ops.ibrnet_fwd_mfma(mfma_blob, blob, pos_enc, rgb_feat, ray_diff, mask, anti_alias, bf16_blob=bf16_blob)
        ctx.bf16 = bf16_blob is not None
        ctx.save_for_backward(rgb_feat, ray_diff, mask, blob, pos_enc, mfma_blob, smp, pts, cam_ws, *((bf16_blob,) if ctx.bf16 else ()))
        ctx.anti_alias, ctx.feat_shape = anti_alias, tuple(featmaps.shape)
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        rgb_feat, ray_diff, mask, blob, pos_enc, mfma_blob, smp, pts, cam_ws = ctx.saved_tensors[:9]
        d_feat = ops.ibrnet_bwd_mfma_scatter(mfma_blob, blob, pos_enc, rgb_feat, ray_diff, mask, smp, d_raw, ctx.anti_alias, pts, cam_ws,
                                             ctx.feat_shape, bf16_blob=ctx.saved_tensors[9] if ctx.bf16 else None)
        return (d_feat,) + (None,) * 10


# Test hook: 'fused' (default) -- rendering gathers inside the row kernel (no rgb_feat at all); the attack's forward runs the
# stand-alone gather and its backward scatters d rgb_feat from inside the row kernel (a backward that gathers AGAIN for its
# recompute was built and measured 3 % slower per step: not shipped).  'separate': nf_project_gather_fwd /
# nf_ibrnet_* / nf_project_gather_bwd as three stages (always so with the generic kernels, the bf16 rows and the deterministic
# scatter)
GATHER_BWD_FUSION = 'fused'


class IBRNet(nn.Module):
    def __init__(self, args, in_feat_ch=32, n_samples=64, **kwargs):
        super().__init__()
        if in_feat_ch != 32:
            raise ValueError('the HIP IBRNet kernels are built for 32 feature channels (got %d)' % in_feat_ch)
        self.args = args
        self.anti_alias_pooling = args.anti_alias_pooling
        if self.anti_alias_pooling:
            self.s = nn.Parameter(torch.tensor(0.2), requires_grad=True)
        self.n_samples = n_samples
        self.ray_dir_fc = _mlp([4, 16, in_feat_ch + 3])
        self.base_fc = _mlp([(in_feat_ch + 3) * 3, 64, 32])
        self.vis_fc = _mlp([32, 32, 33])
        self.vis_fc2 = nn.Sequential(nn.Linear(32, 32), nn.ELU(), nn.Linear(32, 1), nn.Sigmoid())
        self.geometry_fc = _mlp([32 * 2 + 1, 64, 16])
        self.ray_attention = _RayAttentionParams(4, 16, 4, 4)
        self.out_geometry_fc = nn.Sequential(nn.Linear(16, 16), nn.ELU(), nn.Linear(16, 1), nn.ReLU())
        self.rgb_fc = _mlp([32 + 1 + 4, 16, 8, 1], final_act=False)
        self.register_buffer('pos_encoding', sinusoid_table(n_samples))
        for seq in (self.base_fc, self.vis_fc2, self.vis_fc, self.geometry_fc, self.rgb_fc):
            for m in seq:
                if isinstance(m, nn.Linear):
                    nn.init.kaiming_normal_(m.weight.data)
                    nn.init.zeros_(m.bias.data)
        self._blob = None
        self._mfma_blob = None
        self._bf16_blob = None
        self._blob_key = None
        # args.ibrnet_precision: 'fp32' (default: exact fp32 matrix-core arithmetic, the parity path) or 'bf16' (BASELINE config 5:
        # bf16 operands with fp32 accumulation in the per-(sample, view) row network; ~1e-2 accuracy, never chosen silently)
        self.precision = getattr(args, 'ibrnet_precision', None) or 'fp32'
        if self.precision not in ('fp32', 'bf16'):
            raise ValueError("ibrnet_precision must be 'fp32' or 'bf16' (got %r)" % (self.precision,))

    def _packed(self, device):
        """Flat parameter blob in the kernels' layout, re-packed only when a parameter changed."""
        key = (str(device), self.precision) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._blob is None or key != self._blob_key:
            self._blob = ops.pack_ibrnet_blob(self.state_dict(), device)
            self._mfma_blob = ops.pack_ibrnet_mfma_blob(self._blob)
            self._bf16_blob = ops.pack_ibrnet_bf16_blob(self._mfma_blob) if self.precision == 'bf16' else None
            self._blob_key = key
        return self._blob, self._mfma_blob

    def can_gather(self, featmaps, n_samples, n_views):
        """may forward_gathered take this level?  Matrix-core kernels, channels-last 32-channel maps and nothing to differentiate
        (the attack's forward keeps the stand-alone gather: its backward re-reads rgb_feat for the recompute, which is faster than
        gathering again -- N_rand 4096: 14.7 against 15.2 ms per step)."""
        if not (GATHER_BWD_FUSION == 'fused' and KERNEL_PATH != 'generic' and ops.ibrnet_mfma_supported(n_samples, n_views)
                and ops.ibrnet_gather_layout_ok(featmaps)):
            return False
        return not (torch.is_grad_enabled() and featmaps.requires_grad)

    def forward_gathered(self, xyz, cam_ws, src_rgbs, featmaps):
        """Projector.compute + forward in the network's own kernels: the row kernel projects the samples and takes its bilinear
        taps from the feature maps / source images itself (ops.ibrnet_fwd_mfma_gather); no gradient (rendering, pseudo-GT).
        xyz [n_rays, n_samples, 3], cam_ws from ops.camera_setup, src_rgbs [n_views, h, w, 3]
        -> raw [n_rays, n_samples, 4], mask [n_rays, n_samples, n_views]."""
        blob, mfma_blob = self._packed(xyz.device)
        if torch.is_grad_enabled() and featmaps.requires_grad:
            raise RuntimeError('IBRNet.forward_gathered carries no gradient: differentiate through Projector.compute + forward()')
        raw, mask, _ = ops.ibrnet_fwd_mfma_gather(mfma_blob, blob, self.pos_encoding, xyz.detach(), cam_ws, src_rgbs.detach(),
                                                  featmaps.detach(), bool(self.anti_alias_pooling),
                                                  bf16_blob=self._bf16_blob if self.precision == 'bf16' else None)
        return raw, mask

    def forward(self, rgb_feat, ray_diff, mask):
        """
        :param rgb_feat: [n_rays, n_samples, n_views, 35]
        :param ray_diff: [n_rays, n_samples, n_views, 4]
        :param mask: [n_rays, n_samples, n_views, 1]
        :return: [n_rays, n_samples, 4]  (rgb, sigma)
        """
        blob, mfma_blob = self._packed(rgb_feat.device)
        gather = getattr(rgb_feat, '_nf_gather', None)
        if gather is not None and gather[3] != rgb_feat._version:
            gather = None           # edited in place since Projector.compute: it is no longer the gather of those maps
        if (gather is not None and GATHER_BWD_FUSION == 'fused' and KERNEL_PATH != 'generic'
                and ops.GATHER_BWD != 'deterministic' and torch.is_grad_enabled() and gather[2].requires_grad
                and gather[2].shape[1] == 32 and ops.ibrnet_mfma_supported(rgb_feat.shape[1], rgb_feat.shape[2])):
            pts, cam_ws, featmaps = gather[:3]
            return _IBRNetGatherFunction.apply(featmaps, rgb_feat.detach(), ray_diff, mask[..., 0], blob, mfma_blob, self.pos_encoding,
                                               bool(self.anti_alias_pooling), pts, cam_ws, self._bf16_blob if self.precision == 'bf16' else None)
        return _IBRNetFunction.apply(rgb_feat, ray_diff, mask[..., 0], blob, mfma_blob, self.pos_encoding,
                                     bool(self.anti_alias_pooling), self._bf16_blob if self.precision == 'bf16' else None)
