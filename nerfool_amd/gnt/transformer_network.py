"""GNT as an nn.Module whose forward and backward are HIP kernels.  The module tree reproduces the parameter names of
gnt/transformer_network.py:205-268 (rgbfeat_fc, view_crosstrans.N.{attn_norm, ff_norm, ff.fc1/fc2, attn.{q_fc,k_fc,v_fc,
pos_fc.0/2, attn_fc.0/2, out_fc}}, view_selftrans.N.{...}, q_fcs.N.0/2 on even N, norm, rgb_fc) so that the public GNT
checkpoints load by key.  The parameters are constants of the attack.

Eval mode (`.eval()` / `model.switch_to_eval()`): Dropout = identity, matrix-core kernels where the shape allows -- the view-specific
attack, rendering.  TRAINING mode (round 5): the reference's universal GNT loop runs before `model.switch_to_eval()`
(eval/gnt/eval_adv.py:739-878 vs :959), i.e. with the `Dropout(0.1)` of every attention / feed-forward block live
(gnt/transformer_network.py:45-48, :85-88, :162-166) -- a stochastic forward.  torch's generator cannot be reproduced on a GPU, so a
module in training mode runs its kernels (round 6: the matrix-core pair too -- nf_gnt_fwd_train_mfma / nf_gnt_bwd_train_mfma -- so config 4's
universal loop runs on the same kernels as its view-specific one and can be captured into a hipGraph) with masks from a counter-based generator (csrc/nf_gnt.h: gnt_keep; pinned
against the reference with the same masks injected into its modules, and statistically against the reference's own Dropout --
tests/golden/make_golden_gnt_train.py): every forward call takes the next seed of `self.dropout_seed` (initialised from
`torch.initial_seed()`, so `torch.manual_seed` makes an attack reproducible), its backward regenerates the same masks."""
import torch
import torch.nn as nn

from .. import ops

# Test / diagnostic hook: 'mfma' = matrix-core kernels where the shape allows (S in {32, 64, 96, 128}); 'generic': shape-generic
# kernels only
KERNEL_PATH = 'mfma'


class _FF(nn.Module):
    def __init__(self, dim, hid):
        super().__init__()
        self.fc1 = nn.Linear(dim, hid)
        self.fc2 = nn.Linear(hid, dim)


class _ViewAttnParams(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.q_fc = nn.Linear(dim, dim, bias=False)
        self.k_fc = nn.Linear(dim, dim, bias=False)
        self.v_fc = nn.Linear(dim, dim, bias=False)
        self.pos_fc = nn.Sequential(nn.Linear(4, dim // 8), nn.ReLU(), nn.Linear(dim // 8, dim))
        self.attn_fc = nn.Sequential(nn.Linear(dim, dim // 8), nn.ReLU(), nn.Linear(dim // 8, dim))
        self.out_fc = nn.Linear(dim, dim)


class _RayAttnParams(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.q_fc = nn.Linear(dim, dim, bias=False)
        self.k_fc = nn.Linear(dim, dim, bias=False)
        self.v_fc = nn.Linear(dim, dim, bias=False)
        self.out_fc = nn.Linear(dim, dim)


class _TransformerParams(nn.Module):
    def __init__(self, dim, attn):
        super().__init__()
        self.attn_norm = nn.LayerNorm(dim, eps=1e-6)
        self.ff_norm = nn.LayerNorm(dim, eps=1e-6)
        self.ff = _FF(dim, dim * 4)
        self.attn = attn


class _GNTFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb_feat, ray_diff, mask, pts, ray_d, blob, mfma_blob, depth, ret_alpha, dropout=None):
        need_grad = rgb_feat.requires_grad
        ctx.dropout = dropout
        use_mfma = mfma_blob is not None and ops.gnt_mfma_supported(rgb_feat.shape[1], rgb_feat.shape[2])
        if use_mfma:
            # (each backward reads what ITS forward saved: ctx.use_mfma picks the matching one below)
            out = ops.gnt_fwd_mfma(mfma_blob, rgb_feat, ray_diff, mask, pts, ray_d, depth, save=need_grad, want_alpha=ret_alpha, dropout=dropout)
        else:       # shape-generic kernels (training mode: dropout = (seed, p) by value only -- no seed word, not capturable)
            if dropout is not None and len(dropout) > 2:
                raise RuntimeError('a captured training-mode GNT forward needs the matrix-core kernels (S in {32, 64, 96, 128})')
            out = ops.gnt_fwd(blob, rgb_feat, ray_diff, mask, pts, ray_d, depth, save=need_grad, want_alpha=ret_alpha, dropout=dropout)
        rgb, ws = out[0], out[1]
        alpha = out[2] if ret_alpha else rgb.new_zeros(0)
        ctx.depth = depth
        ctx.shape = tuple(rgb_feat.shape[:3])
        ctx.have_ws = ws is not None
        ctx.use_mfma = use_mfma
        if ws is not None:
            ctx.save_for_backward(ray_diff, mask, blob, ws, mfma_blob if ctx.use_mfma else None)
        # the attention-derived weights feed depth maps and the fine resampling only (both detached in the reference's rgb-loss
        # path, gnt/render_ray.py:256): no gradient is propagated through them
        ctx.mark_non_differentiable(alpha)
        ctx.set_materialize_grads(False)
        return rgb, alpha

    @staticmethod
    def backward(ctx, d_rgb, _d_alpha=None):
        if not ctx.have_ws:
            raise RuntimeError('GNT forward ran without saved activations (input did not require grad)')
        ray_diff, mask, blob, ws, mfma_blob = ctx.saved_tensors
        if ctx.use_mfma:
            d_rgb_feat = ops.gnt_bwd_mfma(mfma_blob, mask, d_rgb, ws, ctx.shape, ctx.depth, dropout=ctx.dropout)
        else:
            d_rgb_feat = ops.gnt_bwd(blob, ray_diff, mask, d_rgb, ws, ctx.shape, ctx.depth, dropout=ctx.dropout)
        return d_rgb_feat, None, None, None, None, None, None, None, None, None


class GNT(nn.Module):
    def __init__(self, args, in_feat_ch=32, posenc_dim=3, viewenc_dim=3, ret_alpha=False):
        super().__init__()
        if args.netwidth != 64 or in_feat_ch != 32:
            raise ValueError('the HIP GNT kernels are built for netwidth 64 and 32 feature channels')
        w = args.netwidth
        self.trans_depth = args.trans_depth
        self.rgbfeat_fc = nn.Sequential(nn.Linear(in_feat_ch + 3, w), nn.ReLU(), nn.Linear(w, w))
        self.view_selftrans = nn.ModuleList([_TransformerParams(w, _RayAttnParams(w)) for _ in range(args.trans_depth)])
        self.view_crosstrans = nn.ModuleList([_TransformerParams(w, _ViewAttnParams(w)) for _ in range(args.trans_depth)])
        self.q_fcs = nn.ModuleList([
            nn.Sequential(nn.Linear(w + posenc_dim + viewenc_dim, w), nn.ReLU(), nn.Linear(w, w)) if i % 2 == 0 else nn.Identity()
            for i in range(args.trans_depth)])
        if posenc_dim != 63 or viewenc_dim != 63:
            raise ValueError('positional encodings of 3 + 3*2*10 = 63 dims are built into the kernels')
        self.posenc_dim, self.viewenc_dim, self.ret_alpha = posenc_dim, viewenc_dim, ret_alpha
        self.norm = nn.LayerNorm(w)
        self.rgb_fc = nn.Linear(w, 3)
        self._blob = None
        self._mfma_blob = None
        self._blob_key = None
        # training mode: rate of the reference's nn.Dropout instances (transformer_network.py:222-233: ff_dp_rate = attn_dp_rate = 0.1)
        # and the seed of the NEXT training-mode forward (None: taken from torch.initial_seed() at first use)
        self.dropout_p = 0.1
        self.dropout_seed = None
        # a hipGraph-captured PGD step (eval_adv.PGDAttack) must not freeze the seeds: while a capture is open, the k-th training-mode
        # forward reads its seed from word k of this device buffer, which the attack refreshes before every replay (stage_replay_seeds)
        self._seed_words = None
        self._capture_calls = None

    def next_dropout_seed(self):
        """the seed of this training-mode forward; the following call gets the next one"""
        if self.dropout_seed is None:
            self.dropout_seed = torch.initial_seed() & 0xffffffff
        seed = self.dropout_seed
        self.dropout_seed = (seed + 1) & 0xffffffff
        return seed

    MAX_CAPTURED_FORWARDS = 16

    def begin_seed_capture(self, device):
        """called (outside the capture) before a step that runs this module in training mode is captured"""
        # ONE buffer per module for its lifetime on a device: every captured graph holds its address
        dev = torch.device(device)
        if dev.type == 'cuda' and dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        if self._seed_words is None:
            self._seed_words = torch.zeros(self.MAX_CAPTURED_FORWARDS, dtype=torch.int32, device=dev)
        elif self._seed_words.device != dev:
            raise RuntimeError('training-mode GNT steps were captured on %s; capturing on %s too needs another module instance'
                               % (self._seed_words.device, dev))
        self._capture_calls = 0

    def end_seed_capture(self):
        """-> number of training-mode forwards the captured step contains (the seeds one replay consumes)"""
        n, self._capture_calls = self._capture_calls, None
        return n

    def stage_replay_seeds(self, n):
        """before a replay: the next n seeds of the module's sequence -> the words the captured forwards read (stream-ordered copy), so
        that replayed steps consume the same seed sequence as steps enqueued launch by launch"""
        import numpy as np
        if n:
            seeds = np.array([self.next_dropout_seed() for _ in range(n)], dtype=np.uint32).view(np.int32)
            self._seed_words[:n].copy_(torch.from_numpy(seeds).pin_memory(), non_blocking=True)

    def _packed(self, device):
        key = (str(device), KERNEL_PATH) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._blob is None or key != self._blob_key:
            self._blob = ops.pack_gnt_blob(self.state_dict(), self.trans_depth, device)
            self._mfma_blob = ops.pack_gnt_mfma_blob(self._blob, self.trans_depth) if KERNEL_PATH == 'mfma' else None
            self._blob_key = key
        return self._blob, self._mfma_blob

    def forward(self, rgb_feat, ray_diff, mask, pts, ray_d):
        """rgb_feat [R,S,V,35], ray_diff [R,S,V,4], mask [R,S,V,1], pts [R,S,3], ray_d [R,3] -> rgb [R,3] (or [R,3+S])"""
        blob, mfma_blob = self._packed(rgb_feat.device)
        # training mode = the reference's Dropout-active forward (its universal loop, eval/gnt/eval_adv.py:739-878)
        dropout = None
        if self.training:
            if self._capture_calls is not None and rgb_feat.is_cuda and torch.cuda.is_current_stream_capturing():
                k = self._capture_calls
                if k >= self.MAX_CAPTURED_FORWARDS:
                    raise RuntimeError('more than %d training-mode GNT forwards in one captured step' % self.MAX_CAPTURED_FORWARDS)
                self._capture_calls = k + 1
                dropout = (0, self.dropout_p, self._seed_words[k:k + 1])      # the seed of each replay arrives in that word
            else:
                dropout = (self.next_dropout_seed(), self.dropout_p)
        rgb, alpha = _GNTFunction.apply(rgb_feat, ray_diff, mask[..., 0], pts.detach(), ray_d.detach(), blob, mfma_blob,
                                        self.trans_depth, bool(self.ret_alpha), dropout)
        # ret_alpha: [R, 3 + S] = colour | attention of the first sample in the last ray transformer, mean over heads (:303-309)
        return torch.cat([rgb, alpha], dim=1) if self.ret_alpha else rgb
