"""Attack half of the reference's eval/ibrnet/eval_adv.py with the same call surface -- `clamp`, `init_adv_perturb`,
`optimize_adv_perturb` (rgb-loss path) -- plus `PGDAttack`, the view-specific / universal loop of eval_adv.py:609-740,
762-843 with the Adam-ascent / sign-PGD update and both projections fused into one HIP kernel, and an optional
ray-sharded multi-GPU mode (one RCCL all-reduce of d(delta) per step).

Everything outside the rgb-loss attack (pseudo ground truth, depth / density / camera losses, PCGrad, camera
perturbation, purification) is out of scope and raises NotImplementedError when requested."""
import torch

from . import ops
from .ibrnet.criterion import Criterion
from .ibrnet.render_ray import render_rays
from .ibrnet.sample_ray import RaySamplerSingleImage

criterion = Criterion()

_UNSUPPORTED_FLAGS = ('gt_depth_path', 'use_patch_sampling', 'use_pseudo_gt', 'density_loss', 'depth_var_loss',
                      'depth_diff_loss', 'depth_consistency_loss', 'depth_smooth_loss', 'camera_consistency_loss',
                      'perturb_camera', 'use_pcgrad', 'use_unseen_views')


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


def optimize_adv_perturb(args, delta, model, projector, src_ray_batch, data, return_loss=True, select_inds=None,
                         shard=None, criterion=None, lookahead=False):
    """One loss evaluation of the attack (eval_adv.py:258-310,512-519): draw N_rand rays of the target view `data`,
    features from the PERTURBED source images, colours from the CLEAN ones, masked MSE on coarse + fine.

    select_inds: optional explicit pixel indices (otherwise drawn from the reference's RandomState(234) stream).
    shard: optional `RayShard` -- this rank renders its slice of the drawn rays and the loss denominators are the
    all-reduced mask counts.
    lookahead: the caller is a PGD loop that will draw again -- the next iteration's pixel pick is prepared on a helper
    thread (same RandomState stream, see ibrnet/sample_ray.py)."""
    _reject_out_of_scope(args)
    device = delta.device
    sampler = RaySamplerSingleImage.cached(data, device)
    if select_inds is None:
        n_draw = args.N_rand * (shard.world if shard is not None else 1)
        select_inds = sampler.sample_random_pixel(n_draw, getattr(args, 'sample_mode', 'uniform'),
                                                  getattr(args, 'center_ratio', 0.8), lookahead=lookahead)
        if shard is not None:
            select_inds = select_inds[shard.rank::shard.world]
    train_ray_batch = sampler.select(select_inds)
    if shard is not None and shard.shard_views and shard.world > 1:
        featmaps = shard.view_sharded_featmaps(model.feature_net, src_ray_batch['src_rgbs'], delta)
    else:
        featmaps = model.feature_net((src_ray_batch['src_rgbs'] + delta).squeeze(0).permute(0, 3, 1, 2))
    if _is_gnt(model):      # eval/gnt/eval_adv.py:319-333: GNT renderer, criterion passed in (unmasked MSE)
        from .gnt.criterion import Criterion as GntCriterion
        from .gnt.render_ray import render_rays as gnt_render_rays
        crit = criterion if criterion is not None else GntCriterion()
        ret = gnt_render_rays(ray_batch=train_ray_batch, model=model, projector=projector, featmaps=featmaps,
                              N_samples=args.N_samples, inv_uniform=args.inv_uniform, N_importance=args.N_importance,
                              det=getattr(args, 'det', True), white_bkgd=args.white_bkgd,
                              ret_alpha=getattr(args, 'ret_alpha', False), single_net=getattr(args, 'single_net', True),
                              args=args, src_ray_batch=src_ray_batch)
    else:
        crit = criterion if criterion is not None else globals()['criterion']
        ret = render_rays(ray_batch=train_ray_batch, model=model, projector=projector, featmaps=featmaps,
                          N_samples=args.N_samples, inv_uniform=args.inv_uniform, N_importance=args.N_importance,
                          det=getattr(args, 'det', True), white_bkgd=args.white_bkgd, args=args, src_ray_batch=src_ray_batch)
    counts = None
    if shard is not None:
        counts = shard.global_mask_counts(ret)
    loss_rgb, _ = crit(ret['outputs_coarse'], train_ray_batch, None, None if counts is None else counts[0:1])
    if ret['outputs_fine'] is not None:
        fine_loss, _ = crit(ret['outputs_fine'], train_ray_batch, None, None if counts is None else counts[1:2])
        loss_rgb = loss_rgb + fine_loss
    total_loss = {'rgb': loss_rgb}
    loss = loss_rgb
    if return_loss:
        return loss, total_loss
    return torch.autograd.grad(loss, delta)[0].detach()


class _GatherViewFeatures(torch.autograd.Function):
    """forward: each rank contributes the feature maps of the source views it owns and every rank receives all V of them
    (sum of disjointly filled buffers); backward: d featmaps summed over the ranks (each rank rendered other rays), every
    rank keeps the slice of its own views.  Runs on every rank in both directions, also on ranks that own no view."""

    @staticmethod
    def forward(ctx, local, shard, lo, hi, n_views, split):
        full = torch.empty((n_views,) + tuple(local.shape[1:]), dtype=local.dtype,
                           device=local.device).contiguous(memory_format=torch.channels_last)
        if hi > lo:
            full[lo:hi] = local
        shard.broadcast_views_nhwc_(full)
        ctx.shard, ctx.lo, ctx.hi = shard, lo, hi
        # the per-map channel split happens HERE, so each map's gradient comes back as its own tensor (sliced outside,
        # autograd would assemble them with zero fills, strided copies and an add over all V views)
        return tuple(full.split(list(split), dim=1)) if len(split) > 1 else (full,)

    @staticmethod
    def backward(ctx, *gs):
        ref = next(g for g in gs if g is not None)
        mine = []
        for g in gs:
            g = torch.zeros_like(ref, memory_format=torch.channels_last) if g is None \
                else g.clone(memory_format=torch.channels_last)          # never reduce into autograd's own buffer
            ctx.shard.reduce_views_nhwc_(g)
            mine.append(g[ctx.lo:ctx.hi])
        local = mine[0] if len(mine) == 1 else torch.cat(mine, dim=1).contiguous(memory_format=torch.channels_last)
        return local, None, None, None, None, None


class RayShard:
    """Data-parallel sharding of one PGD step over the ranks of a torch.distributed group (backend 'nccl' = RCCL over
    xGMI on the GPU box, 'gloo' in the CPU tests).

    rays: every rank renders its slice of the step's rays.  Collectives: a 2-float all-reduce of the mask counts (they
    set the loss denominators, utils.py:58) and ONE all-reduce of d(delta) (valid because the CNN backward is linear in the
    upstream gradient).  Every rank then applies the identical deterministic update, so delta stays replicated.

    source views (shard_views=True, SURVEY 8e "shard the CNN by view"): the V source images are independent samples of
    the feature CNN (InstanceNorm is per sample, feature_network.py:137,180), so rank r runs the CNN forward/backward only
    for its contiguous block of views.  Exchanges are per view, to / from the rank that owns it, so that only the bytes that
    are needed travel (half of what all-reducing zero-filled buffers moves) and ragged or empty blocks need no special case:
    forward a broadcast of each view's feature maps from its owner, backward a reduce (sum) of each view's d featmaps to
    its owner; d(delta) of a rank is non-zero only on its own views, so the full gradient is assembled by a broadcast of
    each view's slice from its owner instead of the all-reduce."""

    def __init__(self, group=None, shard_views=True):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.shard_views = bool(shard_views)

    def view_range(self, n_views):
        """contiguous block of source views owned by this rank (ragged when world does not divide V; empty for the
        ranks beyond V)."""
        base, extra = divmod(n_views, self.world)
        lo = self.rank * base + min(self.rank, extra)
        return lo, lo + base + (1 if self.rank < extra else 0)

    def owner(self, v, n_views):
        """group rank that owns source view v (inverse of view_range)"""
        base, extra = divmod(n_views, self.world)
        cut = extra * (base + 1)
        return v // (base + 1) if v < cut else extra + (v - cut) // base

    def _global(self, r):
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def _per_view(self, t, op):
        """one collective per view on its contiguous block of t (views on the leading axis), enqueued back to back in view order
        on every rank, waited for together"""
        n_views = t.shape[0]
        work = [op(t[v], self._global(self.owner(v, n_views))) for v in range(n_views)]
        for w in work:
            w.wait()

    def broadcast_views_nhwc_(self, t):
        """t [V,C,H,W] channels-last, view v valid on owner(v): afterwards every rank holds every view"""
        flat = t.permute(0, 2, 3, 1)            # channels-last storage seen as a plain contiguous [V,H,W,C] tensor
        assert flat.is_contiguous()
        self._per_view(flat, lambda x, r: self.dist.broadcast(x, src=r, group=self.group, async_op=True))
        return t

    def reduce_views_nhwc_(self, t):
        """t [V,C,H,W] channels-last: afterwards view v on owner(v) holds the sum over the ranks (elsewhere: unspecified)"""
        flat = t.permute(0, 2, 3, 1)
        assert flat.is_contiguous()
        self._per_view(flat, lambda x, r: self.dist.reduce(x, dst=r, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
        return t

    def view_sharded_featmaps(self, feature_net, src_rgbs, delta):
        """feature_net(src + delta) with the views split over the ranks -> the same tuple the network returns."""
        n_views, H, W = src_rgbs.shape[1], src_rgbs.shape[2], src_rgbs.shape[3]
        lo, hi = self.view_range(n_views)
        channels, twice, second_none, Hf, Wf = feature_net.describe_output(H, W)
        if hi > lo:
            local = feature_net.forward_full((src_rgbs[:, lo:hi] + delta[:, lo:hi]).squeeze(0).permute(0, 3, 1, 2))
        else:       # nothing to compute here, but this rank still takes part in both exchanges
            local = delta.new_zeros((0, sum(channels), Hf, Wf)) + delta[:, 0:0].sum()
        maps = _GatherViewFeatures.apply(local, self, lo, hi, n_views, tuple(channels))
        if twice:
            return maps[0], maps[0]
        if second_none:
            return maps[0], None
        return maps[0], maps[1]

    def global_mask_counts(self, ret):
        def count(o):       # masked MSE counts the valid rays, the unmasked one (GNT) every ray
            if 'mask' in o and o['mask'] is not None:
                return o['mask'].sum(dtype=torch.float32).reshape(1)
            return torch.full((1,), float(o['rgb'].shape[0]), dtype=torch.float32, device=o['rgb'].device)
        c = count(ret['outputs_coarse'])
        f = count(ret['outputs_fine']) if ret['outputs_fine'] is not None else c
        counts = torch.cat([c, f])
        self.dist.all_reduce(counts, op=self.dist.ReduceOp.SUM, group=self.group)
        return counts

    def all_reduce_grad(self, grad):
        """the full d(delta) [1,V,H,W,3] on every rank.  Views sharded: a rank's gradient is non-zero only on its own views,
        so the sum over the ranks IS each view's slice from its owner -- V broadcasts, half the bytes of an all-reduce."""
        if self.shard_views and self.world > 1 and grad.dim() == 5 and grad.is_contiguous():
            self._per_view(grad[0], lambda x, r: self.dist.broadcast(x, src=r, group=self.group, async_op=True))
        else:
            self.dist.all_reduce(grad, op=self.dist.ReduceOp.SUM, group=self.group)
        return grad


class PGDAttack:
    """The perturbation loop of eval_adv.py (view-specific :783-843, universal :634-740).

    state: delta [1,V,H,W,3] (requires_grad), Adam moments, iteration counter.  `step(data)` = loss forward, backward to
    delta, optional gradient all-reduce, fused update + eps-ball + [0,1]-box projection."""

    def __init__(self, args, model, projector, src_ray_batch, shard=None, delta=None):
        _reject_out_of_scope(args)
        self.args, self.model, self.projector, self.src = args, model, projector, src_ray_batch
        self.shard = shard
        self.epsilon = args.epsilon / 255.0
        self.alpha = args.adv_lr / 255.0
        self.delta = delta if delta is not None else init_adv_perturb(args, src_ray_batch, self.epsilon, 1, 0)
        self.use_adam = bool(getattr(args, 'use_adam', False))
        if self.use_adam:
            self.exp_avg = torch.zeros_like(self.delta.data)
            self.exp_avg_sq = torch.zeros_like(self.delta.data)
        self.iters = 0
        self.last_loss = None

    def lr(self):
        step_size = getattr(self.args, 'lr_step_size', 100)
        gamma = getattr(self.args, 'lr_gamma', 0.5)
        return self.args.adam_lr * gamma ** (self.iters // step_size)     # StepLR stepped after every opt.step()

    def gradient(self, data, select_inds=None, lookahead=True):
        self.delta.grad = None
        loss, _ = optimize_adv_perturb(self.args, self.delta, self.model, self.projector, self.src, data,
                                       return_loss=True, select_inds=select_inds, shard=self.shard, lookahead=lookahead)
        loss.backward()
        grad = self.delta.grad
        if self.shard is not None:
            self.shard.all_reduce_grad(grad)
        self.last_loss = loss.detach()
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
        self.apply(self.gradient(data, select_inds, lookahead))
        return self.last_loss

    def run_view_specific(self, data, n_iters=None):
        """eval_adv.py:796-843: adv_iters steps on one target view."""
        for _ in range(self.args.adv_iters if n_iters is None else n_iters):
            self.step(data)
        return self.delta

    def run_universal(self, train_loader, n_iters=None):
        """eval_adv.py:646-740: cycles over the training views; the reference's `iters > adv_iters` test makes it run
        adv_iters + 1 steps."""
        total = (self.args.adv_iters if n_iters is None else n_iters) + 1
        done = 0
        while done < total:
            for data in train_loader:
                self.step(data)
                done += 1
                if done >= total:
                    break
        return self.delta
