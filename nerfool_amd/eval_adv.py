"""Attack half of the reference's eval/ibrnet/eval_adv.py with the same call surface -- `clamp`, `init_adv_perturb`,
`optimize_adv_perturb` (rgb-loss path) -- plus `PGDAttack`, the view-specific / universal loop of eval_adv.py:609-740,
762-843 with the Adam-ascent / sign-PGD update and both projections fused into one HIP kernel, and an optional
ray-sharded multi-GPU mode (`RayShard`: one RCCL all-reduce of d(delta) per step, or the feature CNN sharded by source
view with one collective per exchange).

Everything outside the rgb-loss attack (depth / density / camera losses, PCGrad, camera perturbation, purification) is
out of scope and raises NotImplementedError when requested."""
import torch

from . import ops
from .ibrnet.criterion import Criterion
from .ibrnet.render_ray import render_rays
from .ibrnet.sample_ray import RaySamplerSingleImage

criterion = Criterion()

_UNSUPPORTED_FLAGS = ('gt_depth_path', 'use_patch_sampling', 'density_loss', 'depth_var_loss',
                      'depth_diff_loss', 'depth_consistency_loss', 'depth_smooth_loss', 'camera_consistency_loss',
                      'perturb_camera', 'use_pcgrad')


def _reject_out_of_scope(args):
    for flag in _UNSUPPORTED_FLAGS:
        if getattr(args, flag, None):
            raise NotImplementedError('--%s belongs to an auxiliary experiment outside the rgb-loss attack path' % flag)


def clamp(X, lower_limit, upper_limit):
    """max(min(X, upper), lower) with tensor or scalar bounds (eval_adv.py:28-29).  Convenience only: the attack loop
    itself uses the fused projection inside the update kernels."""
    hi = upper_limit if torch.is_tensor(upper_limit) else torch.tensor(upper_limit, dtype=X.dtype, device=X.device)
    lo = lower_limit if torch.is_tensor(lower_limit) else torch.tensor(lower_limit, dtype=X.dtype, device=X.device)
    return torch.max(torch.min(X, hi), lo)


def init_adv_perturb(args, src_ray_batch, epsilon, upper_limit, lower_limit):
    """delta ~ U(-eps, eps) from torch's generator on the source device, projected so that src+delta stays in
    [lower, upper] (eval_adv.py:248-254)."""
    src = src_ray_batch['src_rgbs']
    delta = torch.zeros_like(src)
    delta.uniform_(-float(epsilon), float(epsilon))
    ops.project_perturb_(delta, src, -1.0, float(lower_limit), float(upper_limit))
    delta.requires_grad = True
    return delta


def _is_gnt(model):
    from .gnt.transformer_network import GNT
    return isinstance(getattr(model, 'net_coarse', None), GNT)


def _render(gnt, train_ray_batch, model, projector, featmaps, args, src_ray_batch, det):
    if gnt:         # eval/gnt/eval_adv.py:319-333
        from .gnt.render_ray import render_rays as gnt_render_rays
        return gnt_render_rays(ray_batch=train_ray_batch, model=model, projector=projector, featmaps=featmaps,
                               N_samples=args.N_samples, inv_uniform=args.inv_uniform, N_importance=args.N_importance,
                               det=det, white_bkgd=args.white_bkgd, ret_alpha=getattr(args, 'ret_alpha', False),
                               single_net=getattr(args, 'single_net', True), args=args, src_ray_batch=src_ray_batch)
    return render_rays(ray_batch=train_ray_batch, model=model, projector=projector, featmaps=featmaps,
                       N_samples=args.N_samples, inv_uniform=args.inv_uniform, N_importance=args.N_importance,
                       det=det, white_bkgd=args.white_bkgd, args=args, src_ray_batch=src_ray_batch)


def clean_featmaps(model, src_ray_batch):
    """feature maps of the UNPERTURBED source images (no gradient): the pseudo-ground-truth render reads them.  They do not
    depend on delta, so a PGD loop computes them once (PGDAttack) where the reference re-runs the CNN every iteration."""
    with torch.no_grad():
        return model.feature_net(src_ray_batch['src_rgbs'].squeeze(0).permute(0, 3, 1, 2))


def optimize_adv_perturb(args, delta, model, projector, src_ray_batch, data, return_loss=True, select_inds=None,
                         shard=None, criterion=None, lookahead=False, featmaps_clean=None, two_phase=False):
    """One loss evaluation of the attack (eval_adv.py:258-310,512-519): draw N_rand rays of the target view `data`,
    features from the PERTURBED source images, colours from the CLEAN ones, masked MSE on coarse + fine.

    args.use_pseudo_gt (the universal loop sets it under args.use_unseen_views and puts the interpolated target camera into
    data['camera'], eval_adv.py:652-691 -- `PGDAttack.run_universal`; the flag `use_unseen_views` itself is not read here,
    as in the reference): the target colours are the model's own render from the clean source images
    (eval_adv.py:271-290 -- outputs_fine; the GNT flavour takes outputs_coarse, eval/gnt/eval_adv.py:296-315), always with
    det=True.  featmaps_clean: those clean feature maps if the caller already has them.

    select_inds: optional explicit pixel indices (otherwise drawn from the reference's RandomState(234) stream).
    shard: optional `RayShard` -- this rank renders its slice of the drawn rays and the loss denominators are the
    all-reduced mask counts.
    lookahead: the caller is a PGD loop that will draw again -- the next iteration's pixel pick is prepared on a helper
    thread (same RandomState stream, see ibrnet/sample_ray.py).
    two_phase: (shard with the CNN sharded by view) the caller finishes the backward pass itself with
    `shard.finish_view_sharded_backward(loss)` instead of `loss.backward()` -- see RayShard.view_sharded_featmaps."""
    _reject_out_of_scope(args)
    device = delta.device
    sampler = RaySamplerSingleImage.cached(data, device)
    if select_inds is None:
        n_draw = args.N_rand if shard is None else shard.pixels_to_draw(args.N_rand)
        select_inds = sampler.sample_random_pixel(n_draw, getattr(args, 'sample_mode', 'uniform'),
                                                  getattr(args, 'center_ratio', 0.8), lookahead=lookahead)
        if shard is not None:
            select_inds = select_inds[shard.rank::shard.world]
    train_ray_batch = sampler.select(select_inds)
    if shard is not None and shard.exchanges_views:
        featmaps = shard.view_sharded_featmaps(model.feature_net, src_ray_batch['src_rgbs'], delta, two_phase=two_phase)
    else:
        featmaps = model.feature_net((src_ray_batch['src_rgbs'] + delta).squeeze(0).permute(0, 3, 1, 2))
    gnt = _is_gnt(model)
    if getattr(args, 'use_pseudo_gt', False):
        with torch.no_grad():
            if featmaps_clean is None:
                featmaps_clean = clean_featmaps(model, src_ray_batch)
            ret_gt = _render(gnt, train_ray_batch, model, projector, featmaps_clean, args, src_ray_batch, True)
            level = ret_gt['outputs_coarse'] if gnt else ret_gt['outputs_fine']
            train_ray_batch['rgb'] = level['rgb']
            train_ray_batch['depth'] = level['depth']
    if gnt:         # criterion passed in (unmasked MSE)
        from .gnt.criterion import Criterion as GntCriterion
        crit = criterion if criterion is not None else GntCriterion()
    else:
        crit = criterion if criterion is not None else globals()['criterion']
    ret = _render(gnt, train_ray_batch, model, projector, featmaps, args, src_ray_batch, getattr(args, 'det', True))
    counts, global_loss = None, None
    if shard is not None:
        counts, global_loss = shard.global_counts_and_loss(ret, train_ray_batch)
    loss_rgb, _ = crit(ret['outputs_coarse'], train_ray_batch, None, None if counts is None else counts[0:1])
    if ret['outputs_fine'] is not None:
        fine_loss, _ = crit(ret['outputs_fine'], train_ray_batch, None, None if counts is None else counts[1:2])
        loss_rgb = loss_rgb + fine_loss
    # sharded: `loss` is this rank's partial sum over the global denominators (what backward() needs: the partial losses
    # add up to the global loss); the all-reduced value is reported next to it
    total_loss = {'rgb': loss_rgb if global_loss is None else global_loss}
    loss = loss_rgb
    if return_loss:
        return loss, total_loss
    grad = torch.autograd.grad(loss, delta)[0].detach()
    return grad if shard is None else shard.all_reduce_grad(grad)


class _GatherViewFeatures(torch.autograd.Function):
    """forward: each rank contributes out_conv's output for the source views it owns and every rank receives all V of them
    (ONE collective: all-gather, or an all-reduce of disjointly filled buffers when the views do not divide evenly);
    backward: d featmaps summed over the ranks (each rank rendered other rays), every rank keeps the slice of its own views
    (ONE collective: reduce-scatter / all-reduce).  Runs on every rank in both directions, also on ranks that own no view."""

    @staticmethod
    def forward(ctx, local, shard, lo, hi, n_views, split):
        full = shard.gather_views_nhwc(local, lo, hi, n_views)
        ctx.shard, ctx.lo, ctx.hi, ctx.n_views = shard, lo, hi, n_views
        ctx.set_materialize_grads(False)
        # the per-map channel split happens HERE, so each map's gradient comes back as its own tensor (sliced outside,
        # autograd would assemble them with zero fills, strided copies and an add over all V views)
        return tuple(full.split(list(split), dim=1)) if len(split) > 1 else (full,)

    @staticmethod
    def backward(ctx, *gs):
        ref = next(g for g in gs if g is not None)
        gs = [torch.zeros_like(ref, memory_format=torch.channels_last) if g is None else g for g in gs]
        # one buffer for all maps (a fresh tensor: never reduce into autograd's own buffers), one collective
        g = gs[0].clone(memory_format=torch.channels_last) if len(gs) == 1 else \
            torch.cat(gs, dim=1).contiguous(memory_format=torch.channels_last)
        return ctx.shard.scatter_views_nhwc(g, ctx.lo, ctx.hi), None, None, None, None, None


class RayShard:
    """Data-parallel sharding of one PGD step over the ranks of a torch.distributed group (backend 'nccl' = RCCL over
    xGMI on the GPU box, 'gloo' in the CPU tests).

    rays: every rank renders its slice of the step's rays.  With `split_n_rand=False` (weak scaling) `args.N_rand` is the
    PER-RANK ray count: the step draws N_rand * world pixels from the RandomState(234) stream and the batch semantics become
    N_rand * world rays per step.  With `split_n_rand=True` (strong scaling) `args.N_rand` is the GLOBAL batch: the step
    draws exactly the pixels the single-GPU / reference run draws, so the trajectory equals the single-GPU one up to
    summation order.

    shard_views=False -- the north-star form: the feature CNN is replicated; collectives per step = ONE 16-byte all-reduce
    (mask counts = loss denominators, utils.py:58, and loss numerators for the reported loss) + ONE all-reduce of d(delta)
    (valid because the CNN backward is linear in the upstream gradient).

    shard_views=True (SURVEY 8e "shard the CNN by view") -- the V source images are independent samples of the feature CNN
    (InstanceNorm is per sample, feature_network.py:137,180), so rank r runs the CNN forward/backward only for its
    contiguous block of views: 4 collectives per step -- the 16-byte one, an all-gather of the feature maps, a
    reduce-scatter of their gradients, an all-gather of the per-view slices of d(delta) (a rank's d(delta) is non-zero only
    on its own views).  When the views do not divide evenly over the ranks (world does not divide V, or world > V: the ranks
    beyond V own no view and only render) each of the three is an all-reduce of disjointly filled / full buffers instead --
    still one collective per exchange, twice the bytes.

    Every rank then applies the identical deterministic update, so delta stays replicated without a broadcast."""

    def __init__(self, group=None, shard_views=True, split_n_rand=False, exchange_when_alone=False):
        """exchange_when_alone: test hook -- a ONE-rank group still runs the view-sharded flow with its four collectives (the only way
        to drive those RCCL calls, and the graph segments around them, on a single-GPU box)."""
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.shard_views = bool(shard_views)
        self.split_n_rand = bool(split_n_rand)
        self._exchange_when_alone = bool(exchange_when_alone)
        self.collectives = 0            # issued so far (bench.py reports collectives per step)
        self.bytes = 0                  # payload bytes of those collectives (size of the reduced / gathered buffer)
        self.gather_render_to = 0       # group rank that receives a sharded render_single_image (None: every rank)
        self._segmenter = None          # set while PGDAttack captures a step: collectives cut the capture into graph segments
        self._pending = None            # two-phase backward of the view-sharded CNN (view_sharded_featmaps(two_phase=True))

    @property
    def exchanges_views(self):
        return self.shard_views and (self.world > 1 or self._exchange_when_alone)

    def global_rank(self, r):
        return self.dist.get_global_rank(self.group, r) if self.group is not None else r

    def pixels_to_draw(self, n_rand):
        return n_rand if self.split_n_rand else n_rand * self.world

    def view_range(self, n_views):
        """contiguous block of source views owned by this rank (ragged when world does not divide V; empty for the
        ranks beyond V)."""
        base, extra = divmod(n_views, self.world)
        lo = self.rank * base + min(self.rank, extra)
        return lo, lo + base + (1 if self.rank < extra else 0)

    def even(self, n_views):
        return n_views % self.world == 0

    def _count(self, t):
        self.collectives += 1
        self.bytes += t.numel() * t.element_size()

    def _run(self, collective, payload):
        """Issue one collective of the step (`collective()`: a closure over STATIC buffers, nothing allocated inside).  While a step is
        being captured (PGDAttack._capture) the collective is not issued: it ends the current graph segment, is remembered, and the next
        segment begins -- a replayed step is then segment, collective, segment, ... with the collectives enqueued eagerly between the
        graph launches (host-driven backends like gloo cannot be captured; RCCL sees exactly the calls of an eager step)."""
        def issue():
            collective()
            self._count(payload)
        if self._segmenter is not None:
            self._segmenter.cut(issue)
        else:
            issue()

    def gather_views_nhwc(self, local, lo, hi, n_views):
        """local [hi-lo, C, H, W] channels-last (the views this rank owns) -> [V, C, H, W] channels-last on every rank"""
        full = torch.empty((n_views,) + tuple(local.shape[1:]), dtype=local.dtype,
                           device=local.device).contiguous(memory_format=torch.channels_last)
        flat = full.permute(0, 2, 3, 1)            # channels-last storage seen as a plain contiguous [V,H,W,C] tensor
        assert flat.is_contiguous()
        if self.even(n_views):
            mine = local.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
            self._run(lambda: self.dist.all_gather_into_tensor(flat, mine, group=self.group), flat)
        else:
            flat.zero_()
            if hi > lo:
                full[lo:hi] = local
            self._run(lambda: self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group), flat)
        return full

    def scatter_views_nhwc(self, g, lo, hi):
        """g [V, C, H, W] channels-last, this rank's contribution (consumed) -> sum over the ranks of the own views
        [hi-lo, C, H, W] channels-last"""
        flat = g.permute(0, 2, 3, 1)
        assert flat.is_contiguous()
        if self.even(g.shape[0]):
            out = torch.empty((hi - lo,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device).contiguous(memory_format=torch.channels_last)
            out_flat = out.permute(0, 2, 3, 1)
            self._run(lambda: self.dist.reduce_scatter_tensor(out_flat, flat, op=self.dist.ReduceOp.SUM, group=self.group), flat)
            return out
        self._run(lambda: self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group), flat)
        return g[lo:hi]

    def gather_rows(self, buf, dst=0):
        """buf [rows, F] (same shape on every rank) -> [world, rows, F] on group rank `dst`, None on the others; dst None: on
        every rank (all-gather).  ONE collective: the image assembly of a sharded render_single_image."""
        buf = buf.contiguous()
        if dst is not None and buf.is_cuda and self.dist.get_backend(self.group) == 'gloo':
            # functional single-GPU debugging set-up only (bench.py NERFOOL_DIST_BACKEND=gloo): gloo gathers host tensors only
            full = self.gather_rows(buf, None)
            return full if self.rank == dst else None
        if dst is None:
            full = torch.empty((self.world,) + tuple(buf.shape), dtype=buf.dtype, device=buf.device)
            # output = the ranks' buffers concatenated along dim 0 (the one form both RCCL and gloo accept)
            self.dist.all_gather_into_tensor(full.view((self.world * buf.shape[0],) + tuple(buf.shape[1:])), buf, group=self.group)
        else:
            full = torch.empty((self.world,) + tuple(buf.shape), dtype=buf.dtype, device=buf.device) if self.rank == dst else None
            self.dist.gather(buf, list(full.unbind(0)) if full is not None else None, dst=self.global_rank(dst), group=self.group)
        self.collectives += 1
        self.bytes += self.world * buf.numel() * buf.element_size()
        return full

    def view_sharded_featmaps(self, feature_net, src_rgbs, delta, two_phase=False):
        """feature_net(src + delta) with the views split over the ranks -> the same tuple the network returns.

        two_phase (PGDAttack): the gathered maps are autograd LEAVES and the caller finishes the backward pass with
        `finish_view_sharded_backward(loss)` -- d maps, the reduce-scatter, then the CNN backward of the own views -- so that every
        collective is issued from the calling thread (a captured step is cut into graph segments there; autograd's device thread
        cannot end a stream capture).  Otherwise one autograd.Function carries both exchanges and `loss.backward()` just works."""
        n_views, H, W = src_rgbs.shape[1], src_rgbs.shape[2], src_rgbs.shape[3]
        lo, hi = self.view_range(n_views)
        channels, twice, second_none, Hf, Wf = feature_net.describe_output(H, W)
        if hi > lo:
            local = feature_net.forward_full((src_rgbs[:, lo:hi] + delta[:, lo:hi]).squeeze(0).permute(0, 3, 1, 2))
        else:       # nothing to compute here, but this rank still takes part in both exchanges
            local = delta.new_zeros((0, sum(channels), Hf, Wf)) + delta[:, 0:0].sum()
        if two_phase:
            full = self.gather_views_nhwc(local.detach(), lo, hi, n_views)
            maps = tuple(m.requires_grad_() for m in (full.split(list(channels), dim=1) if len(channels) > 1 else (full,)))
            self._pending = (local, maps, lo, hi)
        else:
            maps = _GatherViewFeatures.apply(local, self, lo, hi, n_views, tuple(channels))
        if twice:
            return maps[0], maps[0]
        if second_none:
            return maps[0], None
        return maps[0], maps[1]

    def finish_view_sharded_backward(self, loss):
        """second half of view_sharded_featmaps(two_phase=True): d loss / d maps on this rank's rays -> ONE collective (sum over the ranks,
        each keeps its own views) -> backward through the feature CNN of the own views into delta.grad"""
        local, maps, lo, hi = self._pending
        self._pending = None
        gs = torch.autograd.grad(loss, maps, allow_unused=True)
        ref = next(g for g in gs if g is not None)
        gs = [torch.zeros_like(ref, memory_format=torch.channels_last) if g is None else g for g in gs]
        g = gs[0].clone(memory_format=torch.channels_last) if len(gs) == 1 else \
            torch.cat(gs, dim=1).contiguous(memory_format=torch.channels_last)
        out = self.scatter_views_nhwc(g, lo, hi)
        local.backward(out)

    def global_counts_and_loss(self, ret, train_ray_batch):
        """ONE small all-reduce per step: [valid rays coarse, valid rays fine, sum of masked squared errors coarse, fine]
        -> (counts [2] = the loss denominators every rank divides by, the GLOBAL loss value for reporting)"""
        def stats(o):       # masked MSE counts the valid rays, the unmasked one (GNT) every ray
            err = (o['rgb'].detach() - train_ray_batch['rgb']) ** 2
            if 'mask' in o and o['mask'] is not None:
                m = o['mask'].to(torch.float32)
                return m.sum().reshape(1), (err * m[:, None]).sum().reshape(1)
            return torch.full((1,), float(o['rgb'].shape[0]), dtype=torch.float32, device=err.device), err.sum().reshape(1)
        c, nc = stats(ret['outputs_coarse'])
        f, nf = stats(ret['outputs_fine']) if ret['outputs_fine'] is not None else (c, torch.zeros_like(nc))
        buf = torch.cat([c, f, nc, nf])
        self._run(lambda: self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group), buf)
        loss = buf[2] / (buf[0] * 3 + 1e-6)
        if ret['outputs_fine'] is not None:
            loss = loss + buf[3] / (buf[1] * 3 + 1e-6)
        return buf[:2], loss

    def all_reduce_grad(self, grad):
        """the full d(delta) [1,V,H,W,3] on every rank.  Views sharded: a rank's gradient is non-zero only on its own views,
        so the sum over the ranks IS the concatenation of the owners' slices -- one all-gather, half the bytes of an
        all-reduce (an all-reduce when the views do not divide evenly)."""
        V = grad.shape[1] if grad.dim() == 5 else 0
        if self.exchanges_views and grad.dim() == 5 and grad.is_contiguous() and self.even(V):
            lo, hi = self.view_range(V)
            whole, mine = grad[0], grad[0, lo:hi].clone()
            self._run(lambda: self.dist.all_gather_into_tensor(whole, mine, group=self.group), grad)
        else:
            self._run(lambda: self.dist.all_reduce(grad, op=self.dist.ReduceOp.SUM, group=self.group), grad)
        return grad


class _SegmentedCapture:
    """A PGD step captured as hipGraph SEGMENTS split at its collectives: [graph 0] collective 0 [graph 1] collective 1 ... [graph n].
    Unsharded steps have no collective and are one graph.  All segments allocate from ONE private pool, so a tensor made in segment k
    (saved activations, the autograd graph's buffers, the static collective buffers) is still there when segment k + 1 reads it -- the
    arrangement of torch.cuda.make_graphed_callables (separate forward / backward graphs over one pool).  Nothing executes while
    capturing, so the collectives are NOT issued at capture time (every rank captures at the same step): `replay()` runs the step."""

    def __init__(self, pool, error_mode):
        self.pool, self.error_mode = pool, error_mode
        self.graphs, self.collectives = [], []
        self._ctx = None

    def begin(self):
        g = torch.cuda.CUDAGraph()
        self._ctx = torch.cuda.graph(g, pool=self.pool, capture_error_mode=self.error_mode)
        self._ctx.__enter__()
        self.graphs.append(g)

    def cut(self, collective):
        self._ctx.__exit__(None, None, None)
        self.collectives.append(collective)
        self.begin()

    def end(self, failed=False):
        ctx, self._ctx = self._ctx, None
        if ctx is not None:
            ctx.__exit__(None, None, None)

    def replay(self):
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.collectives):
                self.collectives[i]()


class PGDAttack:
    """The perturbation loop of eval_adv.py (view-specific :783-843, universal :634-740).

    state: delta [1,V,H,W,3] (requires_grad), Adam moments, iteration counter.  `step(data)` = loss forward, backward to
    delta, optional gradient all-reduce, fused update + eps-ball + [0,1]-box projection."""

    def __init__(self, args, model, projector, src_ray_batch, shard=None, delta=None, graph=None):
        """graph: None (default) -- on a GPU the steps of one target view are captured after two eager steps and replayed
        (`_graph_step`): ONE hipGraph without a shard, graph segments split at the collectives with one (`_SegmentedCapture`);
        False -- every step is enqueued launch by launch."""
        _reject_out_of_scope(args)
        self.args, self.model, self.projector, self.src = args, model, projector, src_ray_batch
        self.shard = shard
        self.epsilon = args.epsilon / 255.0
        self.alpha = args.adv_lr / 255.0
        self.delta = delta if delta is not None else init_adv_perturb(args, src_ray_batch, self.epsilon, 1, 0)
        if shard is not None and shard.world > 1:
            # delta must start replicated (every rank then applies the identical deterministic update): rank 0's draw wins
            shard.dist.broadcast(self.delta.data, src=shard.global_rank(0), group=shard.group)
        self.use_adam = bool(getattr(args, 'use_adam', False))
        if self.use_adam:
            self.exp_avg = torch.zeros_like(self.delta.data)
            self.exp_avg_sq = torch.zeros_like(self.delta.data)
        self.iters = 0
        self.last_loss = None
        self._featmaps_clean = None
        self.use_graph = (graph is None or bool(graph)) and self.delta.is_cuda
        self._graphs = {}           # (target view, draw) -> (CUDAGraph, static picks, static {neg_step_size, bc2_sqrt}, loss, gradient, sampler)
        self._g_warm = {}           # key -> eager steps taken on it so far
        self._g_pool = None         # one activation pool for all captured views (replays never overlap; the loss is copied out)
        self.graph_replays = 0

    def lr(self):
        step_size = getattr(self.args, 'lr_step_size', 100)
        gamma = getattr(self.args, 'lr_gamma', 0.5)
        return self.args.adam_lr * gamma ** (self.iters // step_size)     # StepLR stepped after every opt.step()

    def gradient(self, data, select_inds=None, lookahead=True):
        self.delta.grad = None
        if self._featmaps_clean is None and getattr(self.args, 'use_pseudo_gt', False):
            self._featmaps_clean = clean_featmaps(self.model, self.src)
        # (view-sharded CNN: the backward pass is finished in two phases around its reduce-scatter -- every collective of a step is
        #  issued from this thread, which is what lets `_capture` cut the step into graph segments there)
        two_phase = self.shard is not None and self.shard.exchanges_views
        loss, total = optimize_adv_perturb(self.args, self.delta, self.model, self.projector, self.src, data,
                                           return_loss=True, select_inds=select_inds, shard=self.shard, lookahead=lookahead,
                                           featmaps_clean=self._featmaps_clean, two_phase=two_phase)
        if two_phase:
            self.shard.finish_view_sharded_backward(loss)
        else:
            loss.backward()
        grad = self.delta.grad
        if self.shard is not None:
            self.shard.all_reduce_grad(grad)
        self.last_loss = total['rgb'].detach()          # the global loss on every rank
        return grad

    def apply(self, grad):
        src = self.src['src_rgbs']
        if self.use_adam:
            lr = self.lr()
            self.iters += 1
            ops.pgd_adam_step_(self.delta.data, grad, self.exp_avg, self.exp_avg_sq, src, lr, self.iters, self.epsilon)
        else:
            self.iters += 1
            ops.pgd_sign_step_(self.delta.data, grad, src, self.alpha, self.epsilon)

    def step(self, data, select_inds=None, lookahead=True):
        """lookahead: prepare the next step's pixel pick off-thread (harmless if no next step follows: the RandomState(234)
        stream only advances when a pick is consumed)."""
        if self._graph_eligible(select_inds):
            return self._graph_step(data, lookahead)
        self.apply(self.gradient(data, select_inds, lookahead))
        return self.last_loss

    # ---- the step as ONE hipGraph launch (sharded: as graph segments between its collectives) ---------------------------------
    # A PGD step is ~250 kernel launches of fixed shapes on fixed buffers; enqueued one by one from Python they cost the host
    # 4.5 ms against 8 ms of GPU time (BENCH_r04 host_issue_ms_per_step).  The iteration-dependent inputs are exactly two: the pixel
    # picks (host RNG stream -> a static int64 device buffer, refreshed by a stream-ordered copy before each replay) and Adam's
    # {neg_step_size, bc2_sqrt} (nf_pgd_adam_step_dev reads them from a 2-float device buffer).  Everything else -- feature CNN on
    # src + delta, render, loss, backward, fused update on the persistent delta / moment tensors -- is captured once per target view
    # (torch.cuda.CUDAGraph = hipStreamBeginCapture / hipGraphLaunch, activations in a pool shared by the views of a universal loop,
    # at most MAX_GRAPHS views, further ones run eagerly) after two eager steps on that view have warmed every lazy initialisation,
    # and replayed: the same kernels on the same arguments in the same order, so an eager step and a replayed one are
    # interchangeable (bench.py interleaves them: HIP-event brackets need eager launches).
    # Sharded (RayShard): the same capture, cut at every collective of the step (RayShard._run -> _SegmentedCapture.cut): the north-star
    # form is [CNN fwd, render fwd, local loss sums] 16-byte all-reduce [loss, render bwd, CNN bwd] all-reduce of d delta [update]; the
    # view-sharded form has five segments around its four collectives.  Each rank draws the step's picks (the same stream on every
    # rank), keeps its rank::world slice and copies it into its static index buffer.  All ranks take the same eager / capture / replay
    # decisions (they depend on step counts only), so the collectives stay matched.
    MAX_GRAPHS = 64

    def _graph_eligible(self, select_inds):
        if not self.use_graph or select_inds is not None or getattr(self.args, 'use_pseudo_gt', False):
            return False
        from . import prof
        from .ibrnet import feature_network
        if self._training_gnt_nets():
            # GNT in training mode (the reference's universal GNT loop): every forward takes a new Dropout seed.  The matrix-core kernels
            # read it from a device word the attack refreshes before each replay; the shape-generic pair takes it by value -- a kernel
            # argument a replay would freeze -- so other shapes step eagerly
            from .gnt import transformer_network as tn
            a, V = self.args, self.src['src_rgbs'].shape[1]
            shapes = [a.N_samples] + ([a.N_samples + a.N_importance] if a.N_importance > 0 else [])
            if tn.KERNEL_PATH != 'mfma' or not all(ops.gnt_mfma_supported(S, V) for S in shapes):
                return False
        return prof._active is None and feature_network.TRACE_RELU is None and torch.is_grad_enabled()

    def _training_gnt_nets(self):
        if not _is_gnt(self.model):
            return []
        nets = []
        for net in (getattr(self.model, 'net_coarse', None), getattr(self.model, 'net_fine', None)):
            if net is not None and getattr(net, 'training', False) and all(net is not n for n in nets):
                nets.append(net)
        return nets

    def _graph_step(self, data, lookahead):
        import numpy as np
        device = self.delta.device
        sampler = RaySamplerSingleImage.cached(data, device)
        shard = self.shard
        n_rand = self.args.N_rand if shard is None else shard.pixels_to_draw(self.args.N_rand)
        mode, ratio = getattr(self.args, 'sample_mode', 'uniform'), getattr(self.args, 'center_ratio', 0.8)
        # everything a replay would freeze: the target view (its sampler: rays, colours, cameras), the draw, the render settings, the
        # projection radii, the buffers the captured kernels address (delta, moments, sources) and the state of the model's weights
        # (parameter versions: an in-place edit re-packs the kernels' weight images at new addresses)
        a = self.args
        key = (id(sampler), n_rand, mode, ratio, self.use_adam, a.N_samples, a.N_importance, bool(a.inv_uniform), bool(getattr(a, 'det', True)),
               bool(a.white_bkgd), bool(getattr(a, 'ret_alpha', False)), self.epsilon, self.alpha, self.delta.data_ptr(),
               self.src['src_rgbs'].data_ptr(), self.exp_avg.data_ptr() if self.use_adam else 0, self._weights_version(),
               None if shard is None else (shard.rank, shard.world, shard.exchanges_views, shard.split_n_rand))
        picks = sampler.sample_random_pixel(n_rand, mode, ratio, lookahead=lookahead)
        if shard is not None:
            picks = picks[shard.rank::shard.world]
        if key not in self._graphs:
            if self._g_warm.get(key, 0) < 2 or len(self._graphs) >= self.MAX_GRAPHS:
                # eager: warms allocator, record packing, per-device kernel attributes
                self._g_warm[key] = self._g_warm.get(key, 0) + 1
                self.apply(self.gradient(data, picks, False))
                return self.last_loss
            self._capture(key, data, sampler, len(picks))
        graph, g_idx, g_hyper, g_loss = self._graphs[key][:4]
        for net, n_seeds in self._graphs[key][6]:          # training-mode GNT: this replay's Dropout seeds
            net.stage_replay_seeds(n_seeds)
        g_idx.copy_(torch.from_numpy(np.ascontiguousarray(picks, dtype=np.int64)).pin_memory(), non_blocking=True)
        if self.use_adam:
            lr = self.lr()
            self.iters += 1
            g_hyper.copy_(torch.tensor(ops.adam_hyper(lr, self.iters), dtype=torch.float32).pin_memory(), non_blocking=True)
        else:
            self.iters += 1
        graph.replay()
        self.graph_replays += 1
        self.last_loss = g_loss.clone()          # the graph's loss buffer is overwritten by the next replay
        return self.last_loss

    def _weights_version(self):
        v = 0
        for net in (getattr(self.model, 'net_coarse', None), getattr(self.model, 'net_fine', None), getattr(self.model, 'feature_net', None)):
            if net is not None:
                for p in net.parameters():
                    v = (v * 31 + p._version + p.data_ptr()) & 0xffffffffffff
        return v

    def _capture(self, key, data, sampler, n_rand):
        device = self.delta.device
        g_idx = torch.zeros(n_rand, dtype=torch.int64, device=device)
        g_hyper = torch.ones(2, dtype=torch.float32, device=device)
        self.delta.grad = None
        torch.cuda.synchronize(device)
        if self._g_pool is None:
            self._g_pool = torch.cuda.graph_pool_handle()
        # (a process group's watchdog thread may poll its events while this thread captures: with a shard, capture errors are scoped
        #  to the capturing thread)
        graph = _SegmentedCapture(self._g_pool, 'global' if self.shard is None else 'thread_local')
        if self.shard is not None:
            self.shard._segmenter = graph
        train_nets = self._training_gnt_nets()
        for net in train_nets:
            net.begin_seed_capture(device)
        try:
            graph.begin()
            grad = self.gradient(data, select_inds=g_idx, lookahead=False)        # (sharded: its collectives cut the capture)
            if self.use_adam:
                ops.pgd_adam_step_dev_(self.delta.data, grad, self.exp_avg, self.exp_avg_sq, self.src['src_rgbs'], g_hyper, self.epsilon)
            else:
                ops.pgd_sign_step_(self.delta.data, grad, self.src['src_rgbs'], self.alpha, self.epsilon)
            g_loss = self.last_loss
        finally:
            if self.shard is not None:
                self.shard._segmenter = None
            graph.end()
        self.delta.grad = None
        seeds = [(net, net.end_seed_capture()) for net in train_nets]
        # (the sampler is held so that its ray tensors -- and its id(), part of the key -- outlive the graph that reads them)
        self._graphs[key] = (graph, g_idx, g_hyper, g_loss, grad, sampler, seeds)

    def run_view_specific(self, data, n_iters=None):
        """eval_adv.py:796-843: adv_iters steps on one target view."""
        for _ in range(self.args.adv_iters if n_iters is None else n_iters):
            self.step(data)
        return self.delta

    def run_universal(self, train_loader, n_iters=None, render_poses=None):
        """eval_adv.py:646-740: cycles over the training views; the reference's `iters > adv_iters` test makes it run
        adv_iters + 1 steps.

        args.use_unseen_views (eval_adv.py:652-691): every step replaces the batch's target camera by an interpolation of three
        of the scene's render poses drawn from numpy's global generator (`geo_interp.unseen_camera`) and renders the pseudo
        ground truth there.  render_poses: that list of camera-to-world matrices (default: `train_loader.dataset.render_poses`,
        where the reference's loaders keep it); without one the flag raises instead of silently attacking the seen cameras.
        Sharded: rank 0's draw is broadcast so that all ranks step on the same camera."""
        unseen = bool(getattr(self.args, 'use_unseen_views', False))
        if unseen:
            if render_poses is None:
                render_poses = getattr(getattr(train_loader, 'dataset', None), 'render_poses', None)
            if render_poses is None or len(render_poses) < 3:
                raise NotImplementedError('--use_unseen_views needs at least three render poses of the scene: pass render_poses= or a '
                                          'loader whose dataset has .render_poses (the data loaders themselves are out of scope)')
            from .geo_interp import unseen_camera
            self.args.use_pseudo_gt = True      # as GT rgb/depth may not be available for unseen views (eval_adv.py:653)
        total = (self.args.adv_iters if n_iters is None else n_iters) + 1
        done = 0
        while done < total:
            for data in train_loader:
                if unseen:
                    camera = unseen_camera(self.args, render_poses, data['camera'].to(self.delta.device))
                    if self.shard is not None and self.shard.world > 1:
                        self.shard.dist.broadcast(camera, src=self.shard.global_rank(0), group=self.shard.group)
                    data = dict(data, camera=camera)
                self.step(data)
                done += 1
                if done >= total:
                    break
        return self.delta
