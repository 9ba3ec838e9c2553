"""Full-image rendering: chunk loop over render_rays (ibrnet/render_image.py:21-123 call surface and return schema).

Unlike the reference (a synchronising `.cpu()` per key per chunk, render_image.py:98-102), a chunk's outputs leave for the
host on a second HIP stream while the next chunk renders: 2.3 KB per ray at 64 + 64 samples, 1.8 GB for a 756 x 1008 image --
a third of the image time when moved at the end through pageable memory.  The host tensors are page-locked (torch's caching
host allocator: pinned once, reused by later calls) up to `PINNED_BUDGET_BYTES` of live results, pageable beyond that; the
returned values are identical.

Multi-GPU (`shard=RayShard`, SURVEY 8e: "contiguous ray ranges per rank, gather to rank 0"; the reference's only
ray-parallel precedent is the DistributedDataParallel wrapping of ibrnet/model.py:98-110): the image's chunks are dealt to
the ranks as contiguous blocks, every rank renders its block into HBM, and ONE collective per image (a gather of the packed
per-ray records to rank 0, or an all-gather with `shard.gather_render_to=None`) assembles the image.  No collective sits in
the data path of a chunk."""
import weakref
from collections import OrderedDict

import torch

from .projection import Projector
from .render_ray import camera_workspace, render_rays, render_rays_hybrid

_WHOLE = ('camera', 'depth_range', 'src_rgbs', 'src_cameras')
_LEVELS = ('outputs_coarse', 'outputs_fine')

PINNED_BUDGET_BYTES = 6 << 30       # live page-locked result tensors (three 756 x 1008 images at 64 + 64 samples); beyond: pageable
_pinned_live = [0]


def _release(nbytes):
    _pinned_live[0] -= nbytes


def _host_tensor(shape, dtype, want_pinned):
    """page-locked while the live pinned results stay under the budget (accounted until the tensor's storage dies), else
    pageable: retained renders cannot exhaust lockable memory"""
    nbytes = torch.empty((), dtype=dtype).element_size()
    for s in shape:
        nbytes *= int(s)
    if want_pinned and _pinned_live[0] + nbytes <= PINNED_BUDGET_BYTES:
        t = torch.empty(shape, dtype=dtype, pin_memory=True)
        _pinned_live[0] += nbytes
        weakref.finalize(t.untyped_storage(), _release, nbytes)
        return t
    return torch.empty(shape, dtype=dtype)


class HostCollector:
    """chunk outputs -> host tensors [n_rays, ...] per level and key, copied on a second stream as they appear"""

    def __init__(self, n_rays, device):
        self.n_rays, self.device = n_rays, device
        self.on_gpu = device.type == 'cuda'
        self.host = {'outputs_coarse': OrderedDict(), 'outputs_fine': OrderedDict()}
        self.copy_stream = torch.cuda.Stream(device) if self.on_gpu else None
        self.in_flight = []          # chunk outputs stay referenced until their copies are done

    def add(self, i, ret):
        """outputs of the chunk that starts at ray i (a level may be None, and so may a key of the GNT flavour)"""
        if self.on_gpu:
            self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        for level in _LEVELS:
            if ret[level] is None:
                self.host[level] = None
                continue
            for k, v in ret[level].items():
                if v is None:
                    self.host[level].setdefault(k, None)
                    continue
                if self.host[level].get(k) is None:
                    self.host[level][k] = _host_tensor((self.n_rays,) + tuple(v.shape[1:]), v.dtype, self.on_gpu)
                dst = self.host[level][k][i:i + v.shape[0]]
                if self.on_gpu:
                    with torch.cuda.stream(self.copy_stream):
                        dst.copy_(v, non_blocking=True)
                    self.in_flight.append(v)
                else:
                    dst.copy_(v)

    def finish(self, hs, ws):
        """[hs, ws, ...] views of the host tensors, the reference's return schema"""
        if self.on_gpu:
            self.copy_stream.synchronize()
        self.in_flight = []
        all_ret = OrderedDict([('outputs_coarse', OrderedDict()), ('outputs_fine', OrderedDict())])
        for level in _LEVELS:
            if self.host[level] is None:
                all_ret[level] = None
                continue
            for k, t in self.host[level].items():
                all_ret[level][k] = None if t is None else t.reshape(hs, ws, -1).squeeze()
        return all_ret


def chunk_block(n_chunks, rank, world):
    """contiguous block [lo, hi) of the image's chunks rendered by `rank` (sizes differ by at most one; the first
    `n_chunks % world` ranks take the longer blocks; empty for the ranks beyond n_chunks)"""
    base, extra = divmod(n_chunks, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardCollector:
    """This rank's chunk outputs stay in HBM as ONE packed fp32 record per ray (all fields of both levels side by side, in
    schema order; bool / integer fields travel as 0.0 / 1.0 resp. their exact float value), padded to the longest block, so
    that a single collective assembles the image.  The schema -- (level, key, trailing shape, dtype) of every field --
    follows from the first chunk; a rank without any chunk (fewer chunks than ranks) learns it by rendering the image's first ray
    itself (`run_chunks`), so the ONE gather stays the only collective of an image.  Fields must be float32 or bool: the records
    travel as fp32 (an integer field above 2^24 or a float64 field would be rounded silently -- rejected instead)."""

    def __init__(self, n_rays, chunk_size, shard, device):
        self.n_rays, self.chunk_size, self.shard, self.device = n_rays, chunk_size, shard, device
        self.n_chunks = -(-n_rays // chunk_size)
        self.lo, self.hi = chunk_block(self.n_chunks, shard.rank, shard.world)
        self.rows = chunk_block(self.n_chunks, 0, shard.world)[1] * chunk_size     # rank 0 owns a longest block
        self.schema = None
        self.buf = None

    def starts(self):
        return range(self.lo * self.chunk_size, min(self.hi * self.chunk_size, self.n_rays), self.chunk_size)

    @staticmethod
    def _schema_of(ret):
        schema = []
        for level in _LEVELS:
            if ret[level] is None:
                schema.append((level, None, None, None))
                continue
            for k, v in ret[level].items():
                if v is not None and v.dtype not in (torch.float32, torch.bool):
                    raise TypeError('sharded render: field %s/%s is %s; the packed per-ray records carry float32 and bool fields only'
                                    % (level, k, v.dtype))
                schema.append((level, k, None, None) if v is None else (level, k, tuple(v.shape[1:]), str(v.dtype)))
        return schema

    def learn_schema(self, ret):
        """record layout from the outputs of any chunk (their leading dimension is not part of it)"""
        if self.schema is None:
            self.schema = self._schema_of(ret)
            self._alloc()

    @staticmethod
    def _width(shape):
        w = 1
        for s in shape:
            w *= s
        return w

    def _alloc(self):
        width = sum(self._width(s) for _, k, s, _ in self.schema if k is not None and s is not None)
        # (no fill: every row of this rank's block is written by `add`, the padding rows behind it are zeroed in `finish`)
        self.buf = torch.empty(self.rows, width, dtype=torch.float32, device=self.device)

    def add(self, i, ret):
        self.learn_schema(ret)
        r0 = i - self.lo * self.chunk_size
        col = 0
        for level, k, shape, _ in self.schema:
            if k is None or shape is None:
                continue
            v = ret[level][k]
            w = self._width(shape)
            self.buf[r0:r0 + v.shape[0], col:col + w] = v.reshape(v.shape[0], w)
            col += w

    def finish(self, hs, ws):
        """-> the reference's return schema on the gathering rank (host tensors), None on the others"""
        shard = self.shard
        dist = shard.dist
        if self.schema is None:               # an image without rays (on every rank alike): nothing to assemble, no collective
            return OrderedDict([('outputs_coarse', OrderedDict()), ('outputs_fine', OrderedDict())])
        n_mine = max(0, min(self.hi * self.chunk_size, self.n_rays) - self.lo * self.chunk_size)
        if n_mine < self.rows:
            self.buf[n_mine:].zero_()         # the padding rows (dropped again on the gathering rank)
        dst = shard.gather_render_to
        full = shard.gather_rows(self.buf, dst)
        if full is None:
            return None
        # the blocks are contiguous in image order: drop every rank's padding rows
        parts = []
        for r in range(shard.world):
            lo, hi = chunk_block(self.n_chunks, r, shard.world)
            n = min(hi * self.chunk_size, self.n_rays) - lo * self.chunk_size
            if n > 0:
                parts.append(full[r, :n])
        rows = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
        all_ret = OrderedDict([('outputs_coarse', OrderedDict()), ('outputs_fine', OrderedDict())])
        on_gpu = self.device.type == 'cuda'
        col = 0
        pending = []
        for level, k, shape, dtype in self.schema:
            if k is None:
                all_ret[level] = None
                continue
            if shape is None:
                all_ret[level][k] = None
                continue
            w = self._width(shape)
            field = rows[:, col:col + w].reshape((self.n_rays,) + shape).to(getattr(torch, dtype.split('.')[-1]))
            col += w
            host = _host_tensor(tuple(field.shape), field.dtype, on_gpu)
            host.copy_(field, non_blocking=on_gpu)
            pending.append(field)
            all_ret[level][k] = host.reshape(hs, ws, -1).squeeze()
        if on_gpu:
            torch.cuda.current_stream(self.device).synchronize()
        return all_ret


def run_chunks(ray_batch, chunk_size, render_chunk, hs, ws, shard=None):
    """the chunk loop shared by both flavours: `render_chunk(chunk ray batch) -> ret`; with `shard` (world > 1) this rank
    renders its contiguous block of chunks and the image is assembled by one collective"""
    n_rays = ray_batch['ray_o'].shape[0]
    device = ray_batch['ray_o'].device
    if shard is not None and shard.world > 1:
        out = ShardCollector(n_rays, chunk_size, shard, device)
        starts = out.starts()
    else:
        out = HostCollector(n_rays, device)
        starts = range(0, n_rays, chunk_size)
    with torch.no_grad():
        for i in starts:
            chunk = OrderedDict()
            for k, v in ray_batch.items():
                chunk[k] = v if (k in _WHOLE or v is None) else v[i:i + chunk_size]
            out.add(i, render_chunk(chunk))
        if n_rays > 0 and getattr(out, 'schema', True) is None:
            # a rank without a chunk of its own: the record layout from the image's first ray (deterministic on every rank, no
            # exchange -- the image's one collective stays the gather)
            out.learn_schema(render_chunk(OrderedDict((k, v if (k in _WHOLE or v is None) else v[0:1]) for k, v in ray_batch.items())))
    return out.finish(hs, ws)


def render_single_image(ray_sampler, ray_batch, model, projector, chunk_size, N_samples, inv_uniform=False,
                        N_importance=0, det=False, white_bkgd=False, render_stride=1, featmaps=None, args=None,
                        featmaps_clean=None, src_ray_batch=None, shard=None):
    """shard: optional `eval_adv.RayShard` -- the chunks are rendered by all ranks of its group; the result is returned on
    rank `shard.gather_render_to` (default 0; None = on every rank) and is None elsewhere."""
    hybrid = args is not None and (getattr(args, 'use_clean_color', False) or getattr(args, 'use_clean_density', False))
    if hybrid:
        assert featmaps_clean is not None
    kw = dict(projector=projector, N_samples=N_samples, inv_uniform=inv_uniform, N_importance=N_importance, det=det,
              white_bkgd=white_bkgd, args=args, src_ray_batch=src_ray_batch)
    if hybrid:
        render_chunk = lambda chunk: render_rays_hybrid(chunk, model, featmaps, featmaps_clean=featmaps_clean, **kw)
    else:
        # every chunk sees the same cameras: their projection workspace is built once per image, owned by this call
        if isinstance(projector, Projector) and ray_batch['ray_o'].shape[0] > 0:
            kw['cam_ws'] = camera_workspace(ray_batch, src_ray_batch)
        render_chunk = lambda chunk: render_rays(chunk, model, featmaps, **kw)
    all_ret = run_chunks(ray_batch, chunk_size, render_chunk, len(range(0, ray_sampler.H, render_stride)),
                         len(range(0, ray_sampler.W, render_stride)), shard)
    if all_ret is None:
        return None
    coarse = all_ret['outputs_coarse']
    if coarse:                                    # (an image without rays has no fields)
        coarse['rgb'][coarse['mask'] == 0] = 1.   # coarse level only (render_image.py:113)
    return all_ret
