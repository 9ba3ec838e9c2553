"""Tensor-level wrappers of the C ABI: validate, borrow torch storage (raw data_ptr), allocate outputs through torch,
launch on torch's current stream.  No arithmetic happens here; torch is used for memory and streams only."""
import torch

from . import _lib
from . import prof


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


_current_device = getattr(torch._C, '_cuda_getDevice', None) or torch.cuda.current_device


def _launch(fn, name, t, *args):
    """One C-ABI launch: `fn(*args, hipStream_t)` on torch's current stream of t's device, status checked.  Kernel launches, the
    per-device LDS opt-ins and nf_device_cu_count all refer to the CURRENT HIP device, so a tensor living on another GPU than the
    caller's current one is launched under a device guard (restored on every exit path, per thread); the common case -- the tensor
    is on the current device -- takes the raw-handle query only (a step makes ~250 launches)."""
    if t.is_cuda:
        idx = t.device.index
        cur = _current_device()
        if idx is None:
            idx = cur
        if idx != cur:
            with torch.cuda.device(idx):
                rc = fn(*args, _raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(t.device).cuda_stream)
        else:
            rc = fn(*args, _raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(t.device).cuda_stream)
    else:
        rc = fn(*args, 0)
    _lib.check(rc, name)


def _f32(t, name):
    if t.dtype != torch.float32:
        raise TypeError('%s must be float32 (got %s)' % (name, t.dtype))
    if not t.is_cuda and not _lib.emulated():
        raise RuntimeError('%s is on %s: the nerfool_amd kernels run on the GPU only (no CPU fallback)' % (name, t.device))
    return t


def _c(t, name):
    return _f32(t, name).contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


def sample_along_ray(ray_o, ray_d, depth_range, n_samples, inv_uniform, t_rand=None):
    ray_o, ray_d = _c(ray_o, 'ray_o'), _c(ray_d, 'ray_d')
    dr = _c(depth_range.reshape(-1), 'depth_range')
    R = ray_o.shape[0]
    pts = torch.empty(R, n_samples, 3, dtype=torch.float32, device=ray_o.device)
    z = torch.empty(R, n_samples, dtype=torch.float32, device=ray_o.device)
    if t_rand is not None:
        t_rand = _c(t_rand, 't_rand')
    _launch(_lib.lib().nf_sample_along_ray, 'nf_sample_along_ray', ray_o, _ptr(ray_o), _ptr(ray_d), _ptr(dr), R, n_samples, int(bool(inv_uniform)),
                                              _ptr(t_rand), _ptr(pts), _ptr(z))
    return pts, z


def points_from_depths(ray_o, ray_d, z_vals):
    ray_o, ray_d, z_vals = _c(ray_o, 'ray_o'), _c(ray_d, 'ray_d'), _c(z_vals, 'z_vals')
    R, S = z_vals.shape
    pts = torch.empty(R, S, 3, dtype=torch.float32, device=z_vals.device)
    _launch(_lib.lib().nf_points_from_depths, 'nf_points_from_depths', pts, _ptr(ray_o), _ptr(ray_d), _ptr(z_vals), R, S, _ptr(pts))
    return pts


def camera_setup(query_camera, src_cameras):
    q = _c(query_camera.reshape(-1), 'query_camera')
    s = _c(src_cameras.reshape(-1, 34), 'src_cameras')
    V = s.shape[0]
    ws = torch.empty((V + 1) * 16, dtype=torch.float32, device=s.device)
    _launch(_lib.lib().nf_camera_setup, 'nf_camera_setup', s, _ptr(q), _ptr(s), V, _ptr(ws))
    return ws


def project_gather_fwd(xyz, cam_ws, src_rgbs, featmaps, want_pix=False):
    """xyz [N,3]; src_rgbs [V,H,W,3]; featmaps [V,C,Hf,Wf] (any strides)."""
    xyz = _c(xyz, 'xyz')
    src_rgbs = _c(src_rgbs, 'src_rgbs')
    _f32(featmaps, 'featmaps')
    N = xyz.shape[0]
    V, H, W, _ = src_rgbs.shape
    _, C, Hf, Wf = featmaps.shape
    dev = xyz.device
    rgb_feat = torch.empty(N, V, 3 + C, dtype=torch.float32, device=dev)
    ray_diff = torch.empty(N, V, 4, dtype=torch.float32, device=dev)
    mask = torch.empty(N, V, dtype=torch.float32, device=dev)
    pix = torch.empty(V, N, 2, dtype=torch.float32, device=dev) if want_pix else None
    sv, sc, sh, sw = featmaps.stride()
    with prof.launch('nf_project_gather_fwd', xyz, n_pts=N, V=V, C=C):
        _launch(_lib.lib().nf_project_gather_fwd, 'nf_project_gather_fwd', xyz, _ptr(xyz), N, _ptr(cam_ws), V, _ptr(src_rgbs), H, W, _ptr(featmaps), C,
                                                    Hf, Wf, sv, sc, sh, sw, _ptr(rgb_feat), _ptr(ray_diff), _ptr(mask),
                                                    _ptr(pix))
    return rgb_feat, ray_diff, mask, pix


# 'atomic' (default): float atomics, summation order varies run to run (last-bit differences in d featmaps);
# 'deterministic': keys -> stable sort by feature-map pixel -> segmented sum in sorted order (bitwise reproducible, ~3x the time)
GATHER_BWD = 'atomic'


def project_gather_bwd(xyz, cam_ws, V, H, W, d_rgb_feat, feat_shape):
    """Returns d_featmaps [V,C,Hf,Wf] stored channels-last (one 128-byte record per pixel for C=32)."""
    xyz = _c(xyz, 'xyz')
    d_rgb_feat = _c(d_rgb_feat, 'd_rgb_feat')
    _, C, Hf, Wf = feat_shape
    N = xyz.shape[0]
    d_feat = torch.zeros(V, Hf, Wf, C, dtype=torch.float32, device=xyz.device).permute(0, 3, 1, 2)
    sv, sc, sh, sw = d_feat.stride()
    if GATHER_BWD == 'deterministic':
        L = _lib.lib()
        n_taps = N * V * 4
        keys = torch.empty(n_taps, dtype=torch.int32, device=xyz.device)
        wts = torch.empty(n_taps, dtype=torch.float32, device=xyz.device)
        _launch(L.nf_project_gather_keys, 'nf_project_gather_keys', xyz, _ptr(xyz), N, _ptr(cam_ws), V, Hf, Wf, _ptr(keys), _ptr(wts))
        skeys, perm = torch.sort(keys, stable=True)
        with prof.launch('nf_project_gather_bwd_sorted', xyz, n_pts=N, V=V, C=C):
            _launch(L.nf_project_gather_bwd_sorted, 'nf_project_gather_bwd_sorted', xyz, _ptr(skeys), _ptr(perm), _ptr(wts), n_taps, _ptr(d_rgb_feat), C, Hf, Wf, sv, sc, sh, sw,
                                                      _ptr(d_feat))
        return d_feat
    with prof.launch('nf_project_gather_bwd', xyz, n_pts=N, V=V, C=C):
        _launch(_lib.lib().nf_project_gather_bwd, 'nf_project_gather_bwd', xyz, _ptr(xyz), N, _ptr(cam_ws), V, H, W, _ptr(d_rgb_feat), C, Hf, Wf, sv, sc,
                                                    sh, sw, _ptr(d_feat))
    return d_feat


def pixel_mask(mask):
    """mask [..., V] float -> bool [...] : (sum_v mask) > 1."""
    m = _c(mask, 'mask')
    V = m.shape[-1]
    n = m.numel() // V
    out = torch.empty(m.shape[:-1], dtype=torch.bool, device=m.device)
    _launch(_lib.lib().nf_pixel_mask, 'nf_pixel_mask', m, _ptr(m), n, V, _ptr(out))
    return out


def ibrnet_blob_layout():
    """[(state-dict key, offset, rows, cols, transposed)] from the library's own table."""
    import ctypes
    L = _lib.lib()
    out, idx = [], 0
    name = ctypes.create_string_buffer(96)
    off, rows, cols, tr = ctypes.c_int64(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    while L.nf_ibrnet_blob_entry(idx, name, 96, ctypes.byref(off), ctypes.byref(rows), ctypes.byref(cols),
                                 ctypes.byref(tr)) == 0:
        out.append((name.value.decode(), off.value, rows.value, cols.value, bool(tr.value)))
        idx += 1
    return out


def pack_ibrnet_blob(state, device):
    """state: mapping key -> tensor (reference IBRNet state-dict layout) -> flat float32 blob on `device`."""
    blob = torch.zeros(_lib.lib().nf_ibrnet_blob_floats(), dtype=torch.float32)
    for key, off, rows, cols, transposed in ibrnet_blob_layout():
        if key == 's' and key not in state:
            continue                                    # anti_alias_pooling = 0 nets have no `s`
        t = state[key].detach().to('cpu', torch.float32).reshape(rows, cols)
        if transposed:
            t = t.t()
        blob[off:off + rows * cols] = t.contiguous().reshape(-1)
    return blob.to(device)


def pack_ibrnet_mfma_blob(natural_blob):
    """natural blob (any device) -> MFMA-operand-order blob on the same device (re-ordering done by the library on
    host memory)."""
    L = _lib.lib()
    nat = natural_blob.detach().to('cpu', torch.float32).contiguous()
    out = torch.empty(L.nf_ibrnet_mfma_blob_floats(), dtype=torch.float32)
    _lib.check(L.nf_ibrnet_pack_mfma(nat.data_ptr(), out.data_ptr()), 'nf_ibrnet_pack_mfma')
    return out.to(natural_blob.device)


def pack_ibrnet_bf16_blob(mfma_blob):
    """MFMA-order fp32 blob -> bf16 group image of the row kernels (config 5), on the same device"""
    L = _lib.lib()
    src = mfma_blob.detach().to('cpu', torch.float32).contiguous()
    out = torch.empty(L.nf_ibrnet_mfma_bf16_blob_floats(), dtype=torch.float32)
    _lib.check(L.nf_ibrnet_pack_mfma_bf16(src.data_ptr(), out.data_ptr()), 'nf_ibrnet_pack_mfma_bf16')
    return out.to(mfma_blob.device)


def ibrnet_mfma_supported(S, V):
    return bool(_lib.lib().nf_ibrnet_mfma_supported(int(S), int(V)))


def ibrnet_rows_form(form):
    """TEST / DIAGNOSTIC hook (nf_ibrnet_rows_form): 'auto' = the sample-on-the-lane forward kernels where they exist (fp32-grade
    operands as three bf16 parts, 2 <= V <= 10), 'rows' = the row-form kernels always, 'sol_fp32' = sample-on-the-lane with fp32
    matrix-core operands (V <= 4).  Returns the previous setting."""
    names = ('auto', 'rows', 'sol_fp32')
    return names[_lib.lib().nf_ibrnet_rows_form(names.index(form))]


def ibrnet_sol_selected(n_views, bf16=False):
    return bool(_lib.lib().nf_ibrnet_sol_selected(int(n_views), int(bool(bf16))))


def ibrnet_fwd_mfma(mfma_blob, blob, pos_enc, rgb_feat, ray_diff, mask, anti_alias, bf16_blob=None):
    """bf16_blob: run the per-(sample, view) row network on bf16 matrix-core operands (fp32 accumulate)"""
    rgb_feat, ray_diff, mask = _c(rgb_feat, 'rgb_feat'), _c(ray_diff, 'ray_diff'), _c(mask, 'mask')
    R, S, V, F = rgb_feat.shape
    if F != 35:
        raise ValueError('IBRNet expects 3+32 channels per view (got %d)' % F)
    pe = _c(pos_enc.reshape(-1, 16), 'pos_encoding')
    if pe.shape[0] != S:
        raise ValueError('pos_encoding is built for %d samples, input has %d' % (pe.shape[0], S))
    L = _lib.lib()
    ws = torch.empty(L.nf_ibrnet_mfma_workspace_floats(R, S), dtype=torch.float32, device=rgb_feat.device)
    raw = torch.empty(R, S, 4, dtype=torch.float32, device=rgb_feat.device)
    if bf16_blob is not None:
        with prof.launch('nf_ibrnet_fwd_mfma_bf16', raw, R=R, S=S, V=V):
            _launch(L.nf_ibrnet_fwd_mfma_bf16, 'nf_ibrnet_fwd_mfma_bf16', raw, _ptr(bf16_blob), _ptr(mfma_blob), _ptr(blob), _ptr(pe), _ptr(rgb_feat), _ptr(ray_diff),
                                                 _ptr(mask), R, S, V, int(bool(anti_alias)), _ptr(raw), _ptr(ws))
        return raw, ws
    with prof.launch('nf_ibrnet_fwd_mfma', raw, R=R, S=S, V=V):
        _launch(L.nf_ibrnet_fwd_mfma, 'nf_ibrnet_fwd_mfma', raw, _ptr(mfma_blob), _ptr(blob), _ptr(pe), _ptr(rgb_feat), _ptr(ray_diff), _ptr(mask), R, S,
                                        V, int(bool(anti_alias)), _ptr(raw), _ptr(ws))
    return raw, ws


def ibrnet_gather_layout_ok(featmaps):
    """the layout the gather-fused forward reads: 32 channels, channels-last, 16-byte aligned pixel records"""
    sv, sc, sh, sw = featmaps.stride()
    return (featmaps.dim() == 4 and featmaps.shape[1] == 32 and featmaps.dtype == torch.float32 and sc == 1 and sv % 4 == 0 and sh % 4 == 0
            and sw % 4 == 0 and featmaps.data_ptr() % 16 == 0 and featmaps.shape[0] * sv < 2 ** 31)       # 32-bit element offsets


def ibrnet_fwd_mfma_gather(mfma_blob, blob, pos_enc, xyz, cam_ws, src_rgbs, featmaps, anti_alias, bf16_blob=None):
    """Projector.compute + IBRNet.forward in one pair of kernels (no-grad rendering): xyz [R,S,3], src_rgbs [V,H,W,3], featmaps
    [V,32,Hf,Wf] channels-last -> raw [R,S,4], mask [R,S,V]; rgb_feat and ray_diff are never written."""
    xyz, src_rgbs = _c(xyz, 'xyz'), _c(src_rgbs, 'src_rgbs')
    _f32(featmaps, 'featmaps')
    if not ibrnet_gather_layout_ok(featmaps):
        raise ValueError('ibrnet_fwd_mfma_gather: feature maps must be [V,32,Hf,Wf] float32, channels-last, 16-byte aligned')
    R, S, _ = xyz.shape
    V, H, W, _ = src_rgbs.shape
    _, _, Hf, Wf = featmaps.shape
    pe = _c(pos_enc.reshape(-1, 16), 'pos_encoding')
    if pe.shape[0] != S:
        raise ValueError('pos_encoding is built for %d samples, input has %d' % (pe.shape[0], S))
    L = _lib.lib()
    dev = xyz.device
    ws = torch.empty(L.nf_ibrnet_mfma_workspace_floats(R, S), dtype=torch.float32, device=dev)
    raw = torch.empty(R, S, 4, dtype=torch.float32, device=dev)
    mask = torch.empty(R, S, V, dtype=torch.float32, device=dev)
    sv, sc, sh, sw = featmaps.stride()
    with prof.launch('nf_ibrnet_fwd_mfma_bf16' if bf16_blob is not None else 'nf_ibrnet_fwd_mfma', raw, R=R, S=S, V=V):
        _launch(L.nf_ibrnet_fwd_mfma_gather, 'nf_ibrnet_fwd_mfma_gather', raw, _ptr(bf16_blob), _ptr(mfma_blob), _ptr(blob), _ptr(pe), _ptr(xyz), _ptr(cam_ws), _ptr(src_rgbs),
                                               H, W, _ptr(featmaps), Hf, Wf, sv, sc, sh, sw, R, S, V, int(bool(anti_alias)), _ptr(raw), _ptr(ws),
                                               _ptr(mask))
    return raw, mask, ws


def ibrnet_bwd_mfma(mfma_blob, blob, pos_enc, rgb_feat, ray_diff, mask, smp, d_raw, anti_alias, bf16_blob=None):
    rgb_feat, ray_diff, mask = _c(rgb_feat, 'rgb_feat'), _c(ray_diff, 'ray_diff'), _c(mask, 'mask')
    d_raw = _c(d_raw, 'd_raw')
    R, S, V, _ = rgb_feat.shape
    pe = _c(pos_enc.reshape(-1, 16), 'pos_encoding')
    d_ws = torch.empty_like(smp)
    d_rgb_feat = torch.empty_like(rgb_feat)
    if bf16_blob is not None:
        with prof.launch('nf_ibrnet_bwd_mfma_bf16', d_raw, R=R, S=S, V=V):
            _launch(_lib.lib().nf_ibrnet_bwd_mfma_bf16, 'nf_ibrnet_bwd_mfma_bf16', d_raw, _ptr(bf16_blob), _ptr(mfma_blob), _ptr(blob), _ptr(pe), _ptr(rgb_feat),
                                                          _ptr(ray_diff), _ptr(mask), _ptr(smp), _ptr(d_raw), R, S, V,
                                                          int(bool(anti_alias)), _ptr(d_rgb_feat), _ptr(d_ws))
        return d_rgb_feat
    with prof.launch('nf_ibrnet_bwd_mfma', d_raw, R=R, S=S, V=V):
        _launch(_lib.lib().nf_ibrnet_bwd_mfma, 'nf_ibrnet_bwd_mfma', d_raw, _ptr(mfma_blob), _ptr(blob), _ptr(pe), _ptr(rgb_feat), _ptr(ray_diff), _ptr(mask),
                                                 _ptr(smp), _ptr(d_raw), R, S, V, int(bool(anti_alias)), _ptr(d_rgb_feat),
                                                 _ptr(d_ws))
    return d_rgb_feat


def ibrnet_bwd_mfma_scatter(mfma_blob, blob, pos_enc, rgb_feat, ray_diff, mask, smp, d_raw, anti_alias, xyz, cam_ws, feat_shape,
                            bf16_blob=None):
    """ibrnet_bwd_mfma + project_gather_bwd in one: d_raw [R,S,4] -> d_featmaps [V,C,Hf,Wf] (channels-last storage); the
    gradient of rgb_feat never exists in memory.  bf16_blob: the row network on bf16 matrix-core operands."""
    rgb_feat, ray_diff, mask = _c(rgb_feat, 'rgb_feat'), _c(ray_diff, 'ray_diff'), _c(mask, 'mask')
    d_raw, xyz = _c(d_raw, 'd_raw'), _c(xyz, 'xyz')
    R, S, V, _ = rgb_feat.shape
    Vf, C, Hf, Wf = feat_shape
    if Vf != V or C != 32 or xyz.shape[0] != R * S:
        raise ValueError('feature maps %s / sample points %s do not match rgb_feat %s' % (tuple(feat_shape), tuple(xyz.shape), tuple(rgb_feat.shape)))
    pe = _c(pos_enc.reshape(-1, 16), 'pos_encoding')
    d_ws = torch.empty_like(smp)
    d_feat = torch.zeros(V, Hf, Wf, C, dtype=torch.float32, device=xyz.device).permute(0, 3, 1, 2)
    sv, sc, sh, sw = d_feat.stride()
    with prof.launch('nf_ibrnet_bwd_mfma_bf16' if bf16_blob is not None else 'nf_ibrnet_bwd_mfma', d_raw, R=R, S=S, V=V):
        _launch(_lib.lib().nf_ibrnet_bwd_mfma_scatter, 'nf_ibrnet_bwd_mfma_scatter', d_raw, _ptr(bf16_blob), _ptr(mfma_blob), _ptr(blob), _ptr(pe), _ptr(rgb_feat), _ptr(ray_diff), _ptr(mask),
                                                         _ptr(smp), _ptr(d_raw), R, S, V, int(bool(anti_alias)), _ptr(d_ws), _ptr(xyz),
                                                         _ptr(cam_ws), _ptr(d_feat), sv, sc, sh, sw, Hf, Wf)
    return d_feat


def debug_mfma32(a, b, c):
    d = torch.empty_like(c)
    _launch(_lib.lib().nf_debug_mfma32, 'nf_debug_mfma32', d, _ptr(_c(a, 'a')), _ptr(_c(b, 'b')), _ptr(_c(c, 'c')), _ptr(d))
    return d


def ibrnet_fwd(blob, pos_enc, rgb_feat, ray_diff, mask, anti_alias):
    rgb_feat, ray_diff, mask = _c(rgb_feat, 'rgb_feat'), _c(ray_diff, 'ray_diff'), _c(mask, 'mask')
    R, S, V, F = rgb_feat.shape
    if F != 35:
        raise ValueError('IBRNet expects 3+32 channels per view (got %d)' % F)
    pe = _c(pos_enc.reshape(-1, 16), 'pos_encoding')
    if pe.shape[0] != S:
        raise ValueError('pos_encoding is built for %d samples, input has %d' % (pe.shape[0], S))
    L = _lib.lib()
    ws = torch.empty(L.nf_ibrnet_workspace_floats(R, S, V, 0), dtype=torch.float32, device=rgb_feat.device)
    raw = torch.empty(R, S, 4, dtype=torch.float32, device=rgb_feat.device)
    with prof.launch('nf_ibrnet_fwd', raw, R=R, S=S, V=V):
        _launch(L.nf_ibrnet_fwd, 'nf_ibrnet_fwd', raw, _ptr(blob), _ptr(pe), _ptr(rgb_feat), _ptr(ray_diff), _ptr(mask), R, S, V,
                                   int(bool(anti_alias)), _ptr(raw), _ptr(ws))
    return raw


def ibrnet_bwd(blob, pos_enc, rgb_feat, ray_diff, mask, d_raw, anti_alias):
    rgb_feat, ray_diff, mask = _c(rgb_feat, 'rgb_feat'), _c(ray_diff, 'ray_diff'), _c(mask, 'mask')
    d_raw = _c(d_raw, 'd_raw')
    R, S, V, _ = rgb_feat.shape
    pe = _c(pos_enc.reshape(-1, 16), 'pos_encoding')
    L = _lib.lib()
    ws = torch.empty(L.nf_ibrnet_workspace_floats(R, S, V, 1), dtype=torch.float32, device=rgb_feat.device)
    d_rgb_feat = torch.empty_like(rgb_feat)
    with prof.launch('nf_ibrnet_bwd', d_raw, R=R, S=S, V=V):
        _launch(L.nf_ibrnet_bwd, 'nf_ibrnet_bwd', d_raw, _ptr(blob), _ptr(pe), _ptr(rgb_feat), _ptr(ray_diff), _ptr(mask), _ptr(d_raw), R, S, V,
                                   int(bool(anti_alias)), _ptr(d_rgb_feat), _ptr(ws))
    return d_rgb_feat


def composite_fwd(raw, z_vals, mask, white_bkgd):
    """mask: the per-sample pixel mask [R,S] (bool) or the per-view validity flags [R,S,V] (float: the kernel counts them itself)"""
    raw, z_vals = _c(raw, 'raw'), _c(z_vals, 'z_vals')
    R, S, _ = raw.shape
    if mask.dim() == 3:
        pm, vm, V = None, _c(mask, 'view mask'), mask.shape[2]
        if tuple(mask.shape[:2]) != (R, S):
            raise ValueError('view mask %s does not match raw %s' % (tuple(mask.shape), tuple(raw.shape)))
    else:
        pm, vm, V = mask.contiguous(), None, 0
        if pm.dtype != torch.bool:
            pm = pm != 0
    dev = raw.device
    rgb = torch.empty(R, 3, dtype=torch.float32, device=dev)
    depth = torch.empty(R, dtype=torch.float32, device=dev)
    weights = torch.empty(R, S, dtype=torch.float32, device=dev)
    alpha = torch.empty(R, S, dtype=torch.float32, device=dev)
    ray_mask = torch.empty(R, dtype=torch.bool, device=dev)
    _launch(_lib.lib().nf_composite_fwd, 'nf_composite_fwd', raw, _ptr(raw), _ptr(z_vals), _ptr(pm), _ptr(vm), V, R, S, int(bool(white_bkgd)), _ptr(rgb),
                                           _ptr(depth), _ptr(weights), _ptr(alpha), _ptr(ray_mask))
    return rgb, depth, weights, alpha, ray_mask


def composite_bwd(raw, z_vals, white_bkgd, d_rgb=None, d_depth=None, d_weights=None, d_alpha=None):
    raw, z_vals = _c(raw, 'raw'), _c(z_vals, 'z_vals')
    R, S, _ = raw.shape
    grads = [None if g is None else _c(g, 'upstream gradient') for g in (d_rgb, d_depth, d_weights, d_alpha)]
    d_raw = torch.empty_like(raw)
    _launch(_lib.lib().nf_composite_bwd, 'nf_composite_bwd', raw, _ptr(raw), _ptr(z_vals), R, S, int(bool(white_bkgd)), _ptr(grads[0]), _ptr(grads[1]),
                                           _ptr(grads[2]), _ptr(grads[3]), _ptr(d_raw))
    return d_raw


def sample_fine(z_vals, weights, n_importance, inv_uniform, u_rand=None):
    z_vals, weights = _c(z_vals, 'z_vals'), _c(weights, 'weights')
    R, S = z_vals.shape
    out = torch.empty(R, S + n_importance, dtype=torch.float32, device=z_vals.device)
    if u_rand is not None:
        u_rand = _c(u_rand, 'u_rand')
    _launch(_lib.lib().nf_sample_fine, 'nf_sample_fine', out, _ptr(z_vals), _ptr(weights), R, S, n_importance, int(bool(inv_uniform)), _ptr(u_rand),
                                         _ptr(out))
    return out


def sample_pdf(bins, weights, n_samples, u_rand=None):
    bins, weights = _c(bins, 'bins'), _c(weights, 'weights')
    R, M = weights.shape
    if bins.shape != (R, M + 1):
        raise ValueError('bins must be [N_rays, M+1] for weights [N_rays, M]')
    out = torch.empty(R, n_samples, dtype=torch.float32, device=bins.device)
    if u_rand is not None:
        u_rand = _c(u_rand, 'u_rand')
    _launch(_lib.lib().nf_sample_pdf, 'nf_sample_pdf', out, _ptr(bins), _ptr(weights), R, M, n_samples, _ptr(u_rand), _ptr(out))
    return out


def masked_mse_fwd(rgb, gt, mask_b=None, cnt_override=None):
    rgb, gt = _c(rgb, 'rgb'), _c(gt, 'gt')
    R = rgb.shape[0]
    pm = None
    if mask_b is not None:
        pm = mask_b.contiguous()
        if pm.dtype != torch.bool:
            pm = pm != 0
    out = torch.empty(3, dtype=torch.float32, device=rgb.device)
    _launch(_lib.lib().nf_masked_mse_fwd, 'nf_masked_mse_fwd', rgb, _ptr(rgb), _ptr(gt), _ptr(pm), R, _ptr(cnt_override), _ptr(out))
    return out, pm


def masked_mse_bwd(rgb, gt, pm, cnt, d_loss):
    R = rgb.shape[0]
    d_rgb = torch.empty_like(rgb)
    d_loss = _c(d_loss.reshape(1), 'd_loss')
    _launch(_lib.lib().nf_masked_mse_bwd, 'nf_masked_mse_bwd', rgb, _ptr(rgb), _ptr(gt), _ptr(pm), R, _ptr(cnt), _ptr(d_loss), _ptr(d_rgb))
    return d_rgb


def _flat_inplace(t, name):
    _f32(t, name)
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous (updated in place)' % name)
    return t


def project_perturb_(delta, src, epsilon, lower=0.0, upper=1.0):
    _flat_inplace(delta, 'delta')
    src = _c(src, 'src')
    _launch(_lib.lib().nf_project_perturb, 'nf_project_perturb', delta, _ptr(delta), _ptr(src), delta.numel(), float(epsilon), float(lower), float(upper))
    return delta


def pgd_adam_step_(delta, grad, exp_avg, exp_avg_sq, src, lr, step, epsilon, beta1=0.9, beta2=0.999, adam_eps=1e-8,
                   lower=0.0, upper=1.0):
    """`step` is 1-based.  Bias corrections are HOST doubles rounded to float exactly as torch.optim.Adam does."""
    for t, n in ((delta, 'delta'), (exp_avg, 'exp_avg'), (exp_avg_sq, 'exp_avg_sq')):
        _flat_inplace(t, n)
    grad, src = _c(grad, 'grad'), _c(src, 'src')
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    with prof.launch('nf_pgd_adam_step', delta, n=delta.numel()):
        _launch(_lib.lib().nf_pgd_adam_step, 'nf_pgd_adam_step', delta, _ptr(delta), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(src),
                                               delta.numel(), -(lr / bc1), 1.0 - beta1, beta2, 1.0 - beta2, bc2 ** 0.5,
                                               adam_eps, float(epsilon), float(lower), float(upper))
    return delta


def adam_hyper(lr, step, beta1=0.9, beta2=0.999):
    """(neg_step_size, bc2_sqrt) of torch.optim.Adam's single-tensor step `step` (1-based) at rate `lr`: HOST doubles, rounded to
    float by the consumer -- the two scalars of the fused update that change with the iteration count"""
    return -(lr / (1.0 - beta1 ** step)), (1.0 - beta2 ** step) ** 0.5


def pgd_adam_step_dev_(delta, grad, exp_avg, exp_avg_sq, src, hyper, epsilon, beta1=0.9, beta2=0.999, adam_eps=1e-8, lower=0.0, upper=1.0):
    """pgd_adam_step_ with the per-iteration scalars in a 2-float DEVICE tensor `hyper` = adam_hyper(lr, step): the launch a captured
    PGD step replays (eval_adv.PGDAttack)"""
    for t, n in ((delta, 'delta'), (exp_avg, 'exp_avg'), (exp_avg_sq, 'exp_avg_sq')):
        _flat_inplace(t, n)
    grad, src = _c(grad, 'grad'), _c(src, 'src')
    assert hyper.dtype == torch.float32 and hyper.numel() == 2 and hyper.device == delta.device and hyper.is_contiguous()
    with prof.launch('nf_pgd_adam_step', delta, n=delta.numel()):
        _launch(_lib.lib().nf_pgd_adam_step_dev, 'nf_pgd_adam_step_dev', delta, _ptr(delta), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq),
                _ptr(src), delta.numel(), _ptr(hyper), 1.0 - beta1, beta2, 1.0 - beta2, adam_eps, float(epsilon), float(lower), float(upper))
    return delta


def pgd_sign_step_(delta, grad, src, alpha, epsilon, lower=0.0, upper=1.0):
    _flat_inplace(delta, 'delta')
    grad, src = _c(grad, 'grad'), _c(src, 'src')
    _launch(_lib.lib().nf_pgd_sign_step, 'nf_pgd_sign_step', delta, _ptr(delta), _ptr(grad), _ptr(src), delta.numel(), float(alpha), float(epsilon),
                                           float(lower), float(upper))
    return delta


ACT_NONE, ACT_RELU, ACT_ELU = 0, 1, 2


def in_act_pad_fwd(x, gamma, beta, res, act, pad, eps=1e-5, out=None, c_off=0):
    """y_padded = reflect_pad(act(instance_norm(x) * gamma + beta + res), pad); res may be any strided [N,C,H,W] view.
    out / c_off: write the C planes into channels [c_off, c_off + C) of a preallocated contiguous [N,Ctot,H+2p,W+2p] tensor
    (concatenation without torch.cat)."""
    x = _c(x, 'x')
    N, C, H, W = x.shape
    Hp, Wp = H + 2 * pad, W + 2 * pad
    if out is None:
        yp = torch.empty(N, C, Hp, Wp, dtype=torch.float32, device=x.device)
        y_ptr, y_ns = yp.data_ptr(), 0
    else:
        if not out.is_contiguous() or tuple(out.shape[2:]) != (Hp, Wp) or out.shape[0] != N or c_off + C > out.shape[1]:
            raise ValueError('out must be a contiguous [N, Ctot, H+2p, W+2p] tensor holding channels c_off .. c_off+C')
        yp = out
        y_ptr, y_ns = out.data_ptr() + 4 * c_off * Hp * Wp, out.shape[1] * Hp * Wp
    mean = rstd = scratch = None
    if gamma is not None:
        mean = torch.empty(N * C, dtype=torch.float32, device=x.device)
        rstd = torch.empty(N * C, dtype=torch.float32, device=x.device)
        scratch = torch.empty(N * C * 64, dtype=torch.float64, device=x.device)
    rs = (0, 0, 0, 0)
    if res is not None:
        _f32(res, 'res')
        if tuple(res.shape) != (N, C, H, W):
            raise ValueError('residual shape %s does not match %s' % (tuple(res.shape), (N, C, H, W)))
        rs = res.stride()
    with prof.launch('nf_in_act_pad_fwd', x, n=x.numel()):
        _launch(_lib.lib().nf_in_act_pad_fwd, 'nf_in_act_pad_fwd', x, _ptr(x), N, C, H, W, _ptr(gamma), _ptr(beta), float(eps), _ptr(res), rs[0], rs[1],
                                                rs[2], rs[3], int(act), int(pad), y_ptr, y_ns, _ptr(mean), _ptr(rstd),
                                                _ptr(scratch))
    return yp, mean, rstd


def in_act_pad_bwd(dyp, d_extra, yp, x, gamma, mean, rstd, act, pad, want_d_res, beta=None, d_extra_sub=None, shape=None):
    """dyp may be a channel slice [:, c0:c1] of a contiguous padded tensor (folded in place, no copy); d_extra_sub is the
    gradient of a stride-2 consumer; shape = (N, C, H, W) of the unpadded output when yp is not given."""
    if shape is None:
        N, C, Hp, Wp = yp.shape
        H, W = Hp - 2 * pad, Wp - 2 * pad
    else:
        N, C, H, W = shape
        Hp, Wp = H + 2 * pad, W + 2 * pad
    dy_ns = 0
    if dyp is not None:
        _f32(dyp, 'dy_padded')
        if tuple(dyp.shape) != (N, C, Hp, Wp):
            raise ValueError('dy_padded shape %s does not match %s' % (tuple(dyp.shape), (N, C, Hp, Wp)))
        if dyp.stride()[1:] == (Hp * Wp, Wp, 1) and (N == 1 or dyp.stride(0) >= C * Hp * Wp):
            dy_ns = dyp.stride(0)
        else:
            dyp = dyp.contiguous()
    if d_extra is not None:
        d_extra = _c(d_extra, 'd_extra')
    if d_extra_sub is not None:
        d_extra_sub = _c(d_extra_sub, 'd_extra_sub')
        if tuple(d_extra_sub.shape) != (N, C, (H + 1) // 2, (W + 1) // 2):
            raise ValueError('d_extra_sub shape %s' % (tuple(d_extra_sub.shape),))
    ref = dyp if dyp is not None else (d_extra if d_extra is not None else d_extra_sub)
    dx = torch.empty(N, C, H, W, dtype=torch.float32, device=ref.device)
    d_res = torch.empty_like(dx) if want_d_res else None
    scratch = torch.empty(N * C * 64, dtype=torch.float64, device=ref.device) if gamma is not None else None
    with prof.launch('nf_in_act_pad_bwd', ref, n=dx.numel()):
        _launch(_lib.lib().nf_in_act_pad_bwd, 'nf_in_act_pad_bwd', ref, _ptr(dyp), _ptr(d_extra), _ptr(yp), _ptr(x), N, C, H, W, _ptr(gamma), _ptr(beta),
                                                _ptr(mean), _ptr(rstd), int(act), int(pad), _ptr(d_res), _ptr(dx), _ptr(scratch),
                                                dy_ns, _ptr(d_extra_sub))
    return dx, d_res


def conv1x1_pack(weight, transposed, device):
    """weight [c_out, c_in(, 1, 1)] -> MFMA records of nf_conv1x1 (transposed: the backward-data GEMM)"""
    L = _lib.lib()
    w = weight.detach().to('cpu', torch.float32).reshape(weight.shape[0], weight.shape[1]).contiguous()
    out = torch.empty(L.nf_conv1x1_pack_floats(w.shape[0], w.shape[1]), dtype=torch.float32)
    _lib.check(L.nf_conv1x1_pack(w.data_ptr(), w.shape[0], w.shape[1], int(bool(transposed)), out.data_ptr()), 'nf_conv1x1_pack')
    return out.to(device)


def conv1x1(records, bias, x, c_out, channels_last_out=False, x2=None):
    """1x1 convolution of x [N, c_in, H, W] (ANY strides: NCHW, channels-last, subsampled views) -> [N, c_out, H, W]
    contiguous, or channels-last when channels_last_out.  x2: the input channels continue in a second tensor
    (x = channels 0 .. c1 - 1 with c1 a multiple of 32, x2 the rest, same strides)."""
    _f32(x, 'x')
    N, c_in, H, W = x.shape
    c_split = 0
    if x2 is not None:
        _f32(x2, 'x2')
        if x2.stride() != x.stride() or c_in % 32 != 0:
            x, x2 = torch.cat([x, x2], dim=1), None
            c_in = x.shape[1]
        else:
            c_split, c_in = c_in, c_in + x2.shape[1]
    if channels_last_out:
        y = torch.empty(N, H, W, c_out, dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
    else:
        y = torch.empty(N, c_out, H, W, dtype=torch.float32, device=x.device)
    xs, ys = x.stride(), y.stride()
    with prof.launch('nf_conv1x1', x, n=y.numel()):
        _launch(_lib.lib().nf_conv1x1, 'nf_conv1x1', x, _ptr(records), _ptr(bias), _ptr(x), xs[0], xs[1], xs[2], xs[3], _ptr(y), ys[0], ys[1], ys[2],
                                         ys[3], N, H, W, c_in, c_out, _ptr(x2), c_split)
    return y


def wino_group(c_out):
    """default output channels per workgroup of the Winograd convolution"""
    return 32 if c_out <= 32 else 64


def pad_gather_fwd(src, H, W, pad, top=0, left=0, out=None):
    """reflect_pad(Z, pad) with Z = the [N,C,H,W] tensor that holds src [N,C,eh,ew] (ANY strides: a channels-last image, the
    interior view of a padded activation) at (top, left) and zeros elsewhere -> [N,C,H+2pad,W+2pad] (or written into `out`, a
    channel slice of a larger contiguous padded buffer)"""
    _f32(src, 'src')
    N, C, eh, ew = src.shape
    Hp, Wp = H + 2 * pad, W + 2 * pad
    if out is None:
        out = torch.empty(N, C, Hp, Wp, dtype=torch.float32, device=src.device)
    assert tuple(out.shape) == (N, C, Hp, Wp) and out.stride(3) == 1 and out.stride(2) == Wp
    ss = src.stride()
    with prof.launch('nf_pad_gather_fwd', src, n=out.numel()):
        _launch(_lib.lib().nf_pad_gather_fwd, 'nf_pad_gather_fwd', src, _ptr(src), ss[0], ss[1], ss[2], ss[3], N, C, eh, ew, int(top), int(left), int(H), int(W), int(pad),
                                                _ptr(out), out.stride(0), out.stride(1))
    return out


def pad_gather_bwd(d_out, H, W, pad, eh, ew, top=0, left=0, like=None):
    """adjoint of pad_gather_fwd: d_out [N,C,H+2pad,W+2pad] (contiguous planes, any image / channel stride) -> d src [N,C,eh,ew],
    laid out like `like` (same strides) when given, else contiguous"""
    _f32(d_out, 'd_out')
    N, C, Hp, Wp = d_out.shape
    assert d_out.stride(3) == 1 and d_out.stride(2) == Wp and Hp == H + 2 * pad and Wp == W + 2 * pad
    if like is not None:
        din = torch.empty_strided((N, C, eh, ew), like.stride(), dtype=torch.float32, device=d_out.device)
    else:
        din = torch.empty(N, C, eh, ew, dtype=torch.float32, device=d_out.device)
    ds = din.stride()
    with prof.launch('nf_pad_gather_bwd', d_out, n=d_out.numel()):
        _launch(_lib.lib().nf_pad_gather_bwd, 'nf_pad_gather_bwd', d_out, _ptr(d_out), d_out.stride(0), d_out.stride(1), N, C, int(H), int(W), int(pad), int(eh), int(ew),
                                                int(top), int(left), _ptr(din), ds[0], ds[1], ds[2], ds[3])
    return din


def upsample2x_pad_bwd(d_yp, h, w, pad):
    """adjoint of upsample2x_pad_fwd: d_yp [N,C,2h+2pad,2w+2pad] -> [N,C,h,w]"""
    d_yp = _c(d_yp, 'd_yp')
    N, C = d_yp.shape[0], d_yp.shape[1]
    dx = torch.empty(N, C, h, w, dtype=torch.float32, device=d_yp.device)
    with prof.launch('nf_upsample2x_pad_bwd', d_yp, n=d_yp.numel()):
        _launch(_lib.lib().nf_upsample2x_pad_bwd, 'nf_upsample2x_pad_bwd', d_yp, _ptr(d_yp), N * C, int(h), int(w), int(pad), _ptr(dx), h * w, w)
    return dx


def conv_s2_pack(weight, backward, device):
    """weight [c_out, c_in, ks, ks] (ks 3 or 7) -> MFMA records of the stride-2 direct convolution (backward: its
    backward-data pass)"""
    L = _lib.lib()
    w = weight.detach().to('cpu', torch.float32).contiguous()
    c_out, c_in, ks = w.shape[0], w.shape[1], w.shape[2]
    n = L.nf_conv_s2_pack_floats(c_out, c_in, ks, int(bool(backward)))
    if n < 0:
        raise ValueError('stride-2 convolution kernels exist for 3x3 and 7x7 (got %dx%d)' % (ks, ks))
    out = torch.empty(n, dtype=torch.float32)
    _lib.check(L.nf_conv_s2_pack(w.data_ptr(), c_out, c_in, ks, int(bool(backward)), out.data_ptr()), 'nf_conv_s2_pack')
    return out.to(device)


def conv_s2_fwd(records, x, c_out, ks):
    """stride-2, padding-0 convolution of the pre-padded x [N, c_in, Hi, Wi] (unit column stride) -> [N, c_out, Ho, Wo]"""
    _f32(x, 'x')
    if x.stride(3) != 1:
        x = x.contiguous()
    N, c_in, Hi, Wi = x.shape
    Ho, Wo = (Hi - ks) // 2 + 1, (Wi - ks) // 2 + 1
    y = torch.empty(N, c_out, Ho, Wo, dtype=torch.float32, device=x.device)
    xs, ys = x.stride(), y.stride()
    with prof.launch('nf_conv_s2_fwd', x, n_img=N, c_in=c_in, c_out=c_out, ks=ks, Ho=Ho, Wo=Wo):
        _launch(_lib.lib().nf_conv_s2_fwd, 'nf_conv_s2_fwd', x, _ptr(records), int(ks), _ptr(x), xs[0], xs[1], xs[2], Hi, Wi, _ptr(y), ys[0], ys[1], ys[2],
                                             Ho, Wo, N, c_in, c_out)
    return y


def conv_s2_bwd(records, dy, c_in, ks, Hi, Wi):
    """backward-data of conv_s2_fwd: dy [N, c_out, Ho, Wo] -> dx [N, c_in, Hi, Wi] (all of it written)"""
    _f32(dy, 'dy')
    if dy.stride(3) != 1:
        dy = dy.contiguous()
    N, c_out, Ho, Wo = dy.shape
    dx = torch.empty(N, c_in, Hi, Wi, dtype=torch.float32, device=dy.device)
    ds, xs = dy.stride(), dx.stride()
    with prof.launch('nf_conv_s2_bwd', dy, n_img=N, c_in=c_in, c_out=c_out, ks=ks, Ho=Ho, Wo=Wo):
        _launch(_lib.lib().nf_conv_s2_bwd, 'nf_conv_s2_bwd', dy, _ptr(records), int(ks), _ptr(dy), ds[0], ds[1], ds[2], Ho, Wo, _ptr(dx), xs[0], xs[1], xs[2],
                                             Hi, Wi, N, c_in, c_out)
    return dx


def conv_s2_pack_x3(weight, backward, device):
    """weight [c_out, c_in, 3, 3] -> bf16x3 records of the stride-2 forward (nf_conv_s2_fwd_x3) or backward-data pass (nf_conv_s2_bwd_x3)"""
    L = _lib.lib()
    w = weight.detach().to('cpu', torch.float32).contiguous()
    assert tuple(w.shape[2:]) == (3, 3)
    out = torch.empty(L.nf_conv_s2_x3_pack_floats(w.shape[0], w.shape[1], int(bool(backward))), dtype=torch.float32)
    _lib.check(L.nf_conv_s2_x3_pack(w.data_ptr(), w.shape[0], w.shape[1], int(bool(backward)), out.data_ptr()), 'nf_conv_s2_x3_pack')
    return out.to(device)


def conv_s2_fwd_x3(records, x, c_out):
    """conv_s2_fwd for ks = 3 on the bf16 matrix cores with three-way split operands (fp32-grade)"""
    _f32(x, 'x')
    if x.stride(3) != 1:
        x = x.contiguous()
    N, c_in, Hi, Wi = x.shape
    Ho, Wo = (Hi - 3) // 2 + 1, (Wi - 3) // 2 + 1
    y = torch.empty(N, c_out, Ho, Wo, dtype=torch.float32, device=x.device)
    xs, ys = x.stride(), y.stride()
    with prof.launch('nf_conv_s2_fwd', x, n_img=N, c_in=c_in, c_out=c_out, ks=3, Ho=Ho, Wo=Wo, n_split=3):
        _launch(_lib.lib().nf_conv_s2_fwd_x3, 'nf_conv_s2_fwd_x3', x, _ptr(records), _ptr(x), xs[0], xs[1], xs[2], Hi, Wi, _ptr(y), ys[0], ys[1],
                ys[2], Ho, Wo, N, c_in, c_out)
    return y


def conv_s2_stem_pack_x3(weight, device):
    """weight [c_out, c_in <= 3, 7, 7] -> bf16x3 records of the stem's forward (nf_conv_s2_stem_fwd_x3)"""
    L = _lib.lib()
    w = weight.detach().to('cpu', torch.float32).contiguous()
    assert tuple(w.shape[2:]) == (7, 7) and w.shape[1] <= 3
    out = torch.empty(L.nf_conv_s2_stem_x3_pack_floats(w.shape[0]), dtype=torch.float32)
    _lib.check(L.nf_conv_s2_stem_x3_pack(w.data_ptr(), w.shape[0], w.shape[1], out.data_ptr()), 'nf_conv_s2_stem_x3_pack')
    return out.to(device)


def conv_s2_stem_fwd_x3(records, x, c_out):
    """conv_s2_fwd for the 7x7 stem on the bf16 matrix cores with three-way split operands (fp32-grade)"""
    _f32(x, 'x')
    if x.stride(3) != 1:
        x = x.contiguous()
    N, c_in, Hi, Wi = x.shape
    Ho, Wo = (Hi - 7) // 2 + 1, (Wi - 7) // 2 + 1
    y = torch.empty(N, c_out, Ho, Wo, dtype=torch.float32, device=x.device)
    xs, ys = x.stride(), y.stride()
    with prof.launch('nf_conv_s2_fwd', x, n_img=N, c_in=c_in, c_out=c_out, ks=7, Ho=Ho, Wo=Wo, n_split=3):
        _launch(_lib.lib().nf_conv_s2_stem_fwd_x3, 'nf_conv_s2_stem_fwd_x3', x, _ptr(records), _ptr(x), xs[0], xs[1], xs[2], Hi, Wi, _ptr(y), ys[0],
                ys[1], ys[2], Ho, Wo, N, c_in, c_out)
    return y


def conv_s2_bwd_x3(records, dy, c_in, Hi, Wi):
    """conv_s2_bwd for ks = 3 on the bf16 matrix cores with three-way split operands (fp32-grade)"""
    _f32(dy, 'dy')
    if dy.stride(3) != 1:
        dy = dy.contiguous()
    N, c_out, Ho, Wo = dy.shape
    dx = torch.empty(N, c_in, Hi, Wi, dtype=torch.float32, device=dy.device)
    ds, xs = dy.stride(), dx.stride()
    with prof.launch('nf_conv_s2_bwd', dy, n_img=N, c_in=c_in, c_out=c_out, ks=3, Ho=Ho, Wo=Wo, n_split=3):
        _launch(_lib.lib().nf_conv_s2_bwd_x3, 'nf_conv_s2_bwd_x3', dy, _ptr(records), _ptr(dy), ds[0], ds[1], ds[2], Ho, Wo, _ptr(dx), xs[0], xs[1],
                xs[2], Hi, Wi, N, c_in, c_out)
    return dx


def wino_pack(weight, backward, device, k_per_group=None, n_split=0):
    """weight [c_out, c_in, 3, 3] -> Winograd-domain MFMA records (backward: the backward-data convolution);
    k_per_group: output channels per workgroup, 64 or 32 (default wino_group); n_split: 0 = fp32 matrix-core operands (nf_wino_pack),
    3 / 1 = bf16 parts of the three-way split / plain bf16 operands (nf_wino_bf_pack)"""
    L = _lib.lib()
    w = weight.detach().to('cpu', torch.float32).contiguous()
    c_out, c_in = w.shape[0], w.shape[1]
    n_out = c_in if backward else c_out
    kg = wino_group(n_out) if k_per_group is None else int(k_per_group)
    if n_split:
        out = torch.empty(L.nf_wino_bf_pack_floats(n_out, c_out if backward else c_in, kg, int(n_split)), dtype=torch.float32)
        _lib.check(L.nf_wino_bf_pack(w.data_ptr(), c_out, c_in, int(bool(backward)), kg, int(n_split), out.data_ptr()), 'nf_wino_bf_pack')
        return out.to(device)
    out = torch.empty(L.nf_wino_pack_floats(n_out, c_out if backward else c_in, kg), dtype=torch.float32)
    _lib.check(L.nf_wino_pack(w.data_ptr(), c_out, c_in, int(bool(backward)), kg, out.data_ptr()), 'nf_wino_pack')
    return out.to(device)


def _wino_launch(records, kpg, n_split, x, xs, Hi, Wi, pad, y_ptr, ys, Ho, Wo, N, c_in, c_out):
    """one Winograd launch on raw strides (the entry point by operand form)"""
    L = _lib.lib()
    if n_split:
        _launch(L.nf_conv3x3_wino_bf, 'nf_conv3x3_wino_bf', x, _ptr(records), int(kpg), int(n_split), _ptr(x), xs[0], xs[1], xs[2], Hi, Wi,
                int(pad), y_ptr, ys[0], ys[1], ys[2], Ho, Wo, N, c_in, c_out)
    else:
        _launch(L.nf_conv3x3_wino, 'nf_conv3x3_wino', x, _ptr(records), int(kpg), _ptr(x), xs[0], xs[1], xs[2], Hi, Wi, int(pad), y_ptr,
                ys[0], ys[1], ys[2], Ho, Wo, N, c_in, c_out, 0)


def conv3x3_wino(records, x, c_out, pad, k_per_group=None, n_split=0):
    """3x3 stride-1 convolution of x [N, c_in, Hi, Wi] (unit column stride) with zero padding `pad` (0: the network's
    forward on pre-padded activations, 2: its backward-data on the gradient) -> [N, c_out, Hi - 2 + 2 pad, Wi - 2 + 2 pad].
    k_per_group / n_split must be the values the records were packed with."""
    _f32(x, 'x')
    if x.stride(3) != 1:
        x = x.contiguous()
    N, c_in, Hi, Wi = x.shape
    Ho, Wo = Hi - 2 + 2 * pad, Wi - 2 + 2 * pad
    y = torch.empty(N, c_out, Ho, Wo, dtype=torch.float32, device=x.device)
    xs, ys = x.stride(), y.stride()
    with prof.launch('nf_conv3x3_wino', x, n_img=N, c_in=c_in, c_out=c_out, Hi=Hi, Wi=Wi, Ho=Ho, Wo=Wo, n_split=int(n_split)):
        _wino_launch(records, wino_group(c_out) if k_per_group is None else k_per_group, n_split, x, xs, Hi, Wi, pad, _ptr(y), ys, Ho, Wo,
                     N, c_in, c_out)
    return y


def wino_ring_pack(weight, device):
    """weight [c_out, c_in, 3, 3] -> the 1-D border weights of its backward-data pass (nf_conv3x3_bwd_ring)"""
    L = _lib.lib()
    w = weight.detach().to('cpu', torch.float32).contiguous()
    out = torch.empty(L.nf_wino_ring_pack_floats(w.shape[0], w.shape[1]), dtype=torch.float32)
    _lib.check(L.nf_wino_ring_pack(w.data_ptr(), w.shape[0], w.shape[1], out.data_ptr()), 'nf_wino_ring_pack')
    return out.to(device)


WINO_RING_MAX_CHANNELS = 1365     # gradient channels whose line pieces fit the ring kernel's 64 KB of LDS (nf_conv3x3_bwd_ring)


def _wino_blocks(h, w):
    return -(-h // 8) * -(-w // 16)


def wino_bwd_split_plan(H, W):
    """(rows, columns, ring kinds) of the split backward-data pass of an H x W gradient, or None when the split does not save
    a tenth of the 8 x 16 output blocks: the Winograd kernel covers rows 1 .. rows, columns 1 .. columns of the (H + 2) x (W + 2)
    result -- the bottom row / right column ride along when the last block row / column has room -- the ring kernel the rest.
    A function of the shape alone, so the choice (and with it every rounding) is the same in every run."""
    rows = min(H + 1, -(-H // 8) * 8)
    cols = min(W + 1, -(-W // 16) * 16)
    if 10 * _wino_blocks(rows, cols) > 9 * _wino_blocks(H + 2, W + 2):
        return None
    return rows, cols, 1 | (0 if rows == H + 1 else 2) | 4 | (0 if cols == W + 1 else 8)


WINO_SLOTS = 2 * 256      # workgroups of the Winograd kernel the chip holds at once (two per CU, 256 CUs)


def wino_bwd_split_pays(H, W, n_img, c_dy, c_dx, k_per_group=64):
    """Whether the split backward-data pass is the faster form for this launch: the border-ring kernel costs a fixed ~18 us (latency-bound),
    so the split pays only where the one-launch form on the (H + 2) x (W + 2) output needs a ROUND of workgroups more than the
    interior-aligned region (the chip holds WINO_SLOTS at once) and a workgroup runs long enough (>= 12 chunks of 16 gradient channels,
    ~40 us) for that round to outweigh the ring.  Measured on the MI355X (tools/experimental/ab_split.py): config 2's 48 x 63 planes
    (4 images, 256 channels: 560 -> 384 workgroups) 8.31 -> 8.16 ms per step with the split; config 5's 128^2 / 64^2 / 32^2 planes
    (8 images: 1224 -> 1024, 720 -> 512, 480 -> 256 workgroups of 4 / 8 / 16 chunks) 7.35 -> 7.58 ms with it.  A function of the launch's
    shape alone: the choice -- and with it every rounding -- is the same in every run."""
    plan = wino_bwd_split_plan(H, W)
    if plan is None:
        return False
    groups = -(-c_dx // k_per_group)
    full = _wino_blocks(H + 2, W + 2) * n_img * groups
    part = _wino_blocks(plan[0], plan[1]) * n_img * groups
    return -(-full // WINO_SLOTS) > -(-part // WINO_SLOTS) and -(-c_dy // 16) >= 12


def conv3x3_wino_bwd_split(records, ring_records, dy, c_dx, plan, k_per_group=None, n_split=0):
    """backward-data of a 3x3 stride-1 convolution, dy [N, c_dy, H, W] -> d(padded input) [N, c_dx, H + 2, W + 2], as the Winograd
    kernel on the interior-aligned region + the 1-D ring kernel (plan = wino_bwd_split_plan(H, W)); every element of the result
    is written exactly once."""
    _f32(dy, 'dy')
    if dy.stride(3) != 1:
        dy = dy.contiguous()
    N, c_dy, H, W = dy.shape
    rows, cols, kinds = plan
    g = torch.empty(N, c_dx, H + 2, W + 2, dtype=torch.float32, device=dy.device)
    xs, gs = dy.stride(), g.stride()
    L = _lib.lib()
    with prof.launch('nf_conv3x3_wino', dy, n_img=N, c_in=c_dy, c_out=c_dx, Hi=H, Wi=W, Ho=rows, Wo=cols, n_split=int(n_split)):
        _wino_launch(records, wino_group(c_dx) if k_per_group is None else k_per_group, n_split, dy, xs, H, W, 1,
                     g.data_ptr() + 4 * (gs[2] + 1), gs, rows, cols, N, c_dy, c_dx)
    with prof.launch('nf_conv3x3_bwd_ring', dy, n=N * c_dx * (2 * (W + 2) + 2 * H)):
        _launch(L.nf_conv3x3_bwd_ring, 'nf_conv3x3_bwd_ring', dy, _ptr(ring_records), _ptr(dy), xs[0], xs[1], xs[2], H, W, _ptr(g), gs[0], gs[1], gs[2], N, c_dy, c_dx,
                                         int(kinds))
    return g


def upsample2x_pad_fwd(x, pad):
    """reflect_pad(F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True), pad); x [N,C,h,w] may be the
    interior view of a padded tensor (unit column stride, planes laid out like a contiguous [N,C] grid)."""
    _f32(x, 'x')
    N, C, h, w = x.shape
    sn, sc, sh, sw = x.stride()
    if sw != 1 or sn != C * sc:
        x = x.contiguous()
        sn, sc, sh, sw = x.stride()
    yp = torch.empty(N, C, 2 * h + 2 * pad, 2 * w + 2 * pad, dtype=torch.float32, device=x.device)
    with prof.launch('nf_upsample2x_pad_fwd', x, n=yp.numel()):
        _launch(_lib.lib().nf_upsample2x_pad_fwd, 'nf_upsample2x_pad_fwd', x, _ptr(x), N * C, sc, sh, h, w, int(pad), _ptr(yp))
    return yp


def pack_gnt_blob(state, depth, device):
    """reference GNT state-dict (gnt/transformer_network.py module paths) -> flat blob in the kernels' layout"""
    import ctypes
    L = _lib.lib()
    blob = torch.zeros(L.nf_gnt_blob_floats(depth), dtype=torch.float32)
    name = ctypes.create_string_buffer(128)
    off, rows, cols, tr = ctypes.c_int64(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    idx = 0
    while L.nf_gnt_blob_entry(depth, idx, name, 128, ctypes.byref(off), ctypes.byref(rows), ctypes.byref(cols),
                              ctypes.byref(tr)) == 0:
        key = name.value.decode()
        idx += 1
        if not key:
            continue
        t = state[key].detach().to('cpu', torch.float32).reshape(rows.value, cols.value)
        if tr.value:
            t = t.t()
        blob[off.value:off.value + rows.value * cols.value] = t.contiguous().reshape(-1)
    return blob.to(device)


def gnt_fwd(blob, rgb_feat, ray_diff, mask, pts, ray_d, depth, save, want_alpha=False, dropout=None):
    """dropout = (seed, p): TRAINING mode -- the eight Dropout sites of every layer live, masks from the counter-based generator
    (nf_gnt_fwd_train); the backward must be given the same pair"""
    rgb_feat, ray_diff, mask = _c(rgb_feat, 'rgb_feat'), _c(ray_diff, 'ray_diff'), _c(mask, 'mask')
    pts, ray_d = _c(pts, 'pts'), _c(ray_d, 'ray_d')
    R, S, V, F = rgb_feat.shape
    if F != 35:
        raise ValueError('GNT expects 3+32 channels per view (got %d)' % F)
    L = _lib.lib()
    ws = torch.empty(L.nf_gnt_workspace_floats(R, S, V, depth, int(bool(save))), dtype=torch.float32, device=rgb_feat.device)
    rgb = torch.empty(R, 3, dtype=torch.float32, device=rgb_feat.device)
    alpha = torch.empty(R, S, dtype=torch.float32, device=rgb_feat.device) if want_alpha else None
    with prof.launch('nf_gnt_fwd', rgb, R=R, S=S, V=V, depth=depth):
        if dropout is not None:
            _launch(L.nf_gnt_fwd_train, 'nf_gnt_fwd_train', rgb, _ptr(blob), _ptr(rgb_feat), _ptr(ray_diff), _ptr(mask), _ptr(pts), _ptr(ray_d), R, S, V,
                    depth, int(bool(save)), _ptr(rgb), _ptr(alpha), _ptr(ws), int(dropout[0]) & 0xffffffff, float(dropout[1]))
        else:
            _launch(L.nf_gnt_fwd, 'nf_gnt_fwd', rgb, _ptr(blob), _ptr(rgb_feat), _ptr(ray_diff), _ptr(mask), _ptr(pts), _ptr(ray_d), R, S, V, depth,
                    int(bool(save)), _ptr(rgb), _ptr(alpha), _ptr(ws))
    if want_alpha:
        return rgb, (ws if save else None), alpha
    return rgb, (ws if save else None)


def pack_gnt_mfma_blob(blob, depth):
    """natural GNT blob (pack_gnt_blob) -> record-ordered blob of the matrix-core forward, on the same device"""
    L = _lib.lib()
    nat = blob.detach().to('cpu', torch.float32).contiguous()
    out = torch.empty(L.nf_gnt_mfma_blob_floats(depth), dtype=torch.float32)
    _lib.check(L.nf_gnt_pack_mfma(depth, nat.data_ptr(), out.data_ptr()), 'nf_gnt_pack_mfma')
    return out.to(blob.device)


def gnt_mfma_supported(n_samples, n_views):
    return bool(_lib.lib().nf_gnt_mfma_supported(int(n_samples), int(n_views)))


def gnt_fwd_mfma(mfma_blob, rgb_feat, ray_diff, mask, pts, ray_d, depth, save, want_alpha=False, dropout=None):
    """dropout = (seed, p[, seed word]): TRAINING mode on the matrix-core kernels (nf_gnt_fwd_train_mfma); seed word: a 1-element int32
    device tensor the kernel reads the seed from instead (captured steps); the backward must be given the same triple"""
    rgb_feat, ray_diff, mask = _c(rgb_feat, 'rgb_feat'), _c(ray_diff, 'ray_diff'), _c(mask, 'mask')
    pts, ray_d = _c(pts, 'pts'), _c(ray_d, 'ray_d')
    R, S, V, F = rgb_feat.shape
    if F != 35:
        raise ValueError('GNT expects 3+32 channels per view (got %d)' % F)
    L = _lib.lib()
    ws = torch.empty(L.nf_gnt_workspace_floats(R, S, V, depth, int(bool(save))), dtype=torch.float32, device=rgb_feat.device)
    rgb = torch.empty(R, 3, dtype=torch.float32, device=rgb_feat.device)
    alpha = torch.empty(R, S, dtype=torch.float32, device=rgb_feat.device) if want_alpha else None
    with prof.launch('nf_gnt_fwd_mfma', rgb, R=R, S=S, V=V, depth=depth):
        if dropout is not None:
            word = dropout[2] if len(dropout) > 2 else None
            _launch(L.nf_gnt_fwd_train_mfma, 'nf_gnt_fwd_train_mfma', rgb, _ptr(mfma_blob), _ptr(rgb_feat), _ptr(ray_diff), _ptr(mask), _ptr(pts),
                    _ptr(ray_d), R, S, V, depth, int(bool(save)), _ptr(rgb), _ptr(alpha), _ptr(ws), int(dropout[0]) & 0xffffffff,
                    float(dropout[1]), _ptr(word))
        else:
            _launch(L.nf_gnt_fwd_mfma, 'nf_gnt_fwd_mfma', rgb, _ptr(mfma_blob), _ptr(rgb_feat), _ptr(ray_diff), _ptr(mask), _ptr(pts), _ptr(ray_d), R, S, V,
                    depth, int(bool(save)), _ptr(rgb), _ptr(alpha), _ptr(ws))
    if want_alpha:
        return rgb, (ws if save else None), alpha
    return rgb, (ws if save else None)


def gnt_bwd_mfma(mfma_blob, mask, d_rgb, ws, shape, depth, dropout=None):
    R, S, V = shape
    mask, d_rgb = _c(mask, 'mask'), _c(d_rgb, 'd_rgb')
    d_rgb_feat = torch.empty(R, S, V, 35, dtype=torch.float32, device=d_rgb.device)
    with prof.launch('nf_gnt_bwd_mfma', d_rgb, R=R, S=S, V=V, depth=depth):
        if dropout is not None:
            word = dropout[2] if len(dropout) > 2 else None
            _launch(_lib.lib().nf_gnt_bwd_train_mfma, 'nf_gnt_bwd_train_mfma', d_rgb, _ptr(mfma_blob), _ptr(mask), _ptr(d_rgb), R, S, V, depth,
                    _ptr(d_rgb_feat), _ptr(ws), int(dropout[0]) & 0xffffffff, float(dropout[1]), _ptr(word))
        else:
            _launch(_lib.lib().nf_gnt_bwd_mfma, 'nf_gnt_bwd_mfma', d_rgb, _ptr(mfma_blob), _ptr(mask), _ptr(d_rgb), R, S, V, depth, _ptr(d_rgb_feat), _ptr(ws))
    return d_rgb_feat


def gnt_bwd(blob, ray_diff, mask, d_rgb, ws, shape, depth, dropout=None):
    R, S, V = shape
    ray_diff, mask, d_rgb = _c(ray_diff, 'ray_diff'), _c(mask, 'mask'), _c(d_rgb, 'd_rgb')
    d_rgb_feat = torch.empty(R, S, V, 35, dtype=torch.float32, device=d_rgb.device)
    with prof.launch('nf_gnt_bwd', d_rgb, R=R, S=S, V=V, depth=depth):
        if dropout is not None:
            _launch(_lib.lib().nf_gnt_bwd_train, 'nf_gnt_bwd_train', d_rgb, _ptr(blob), _ptr(ray_diff), _ptr(mask), _ptr(d_rgb), R, S, V, depth,
                    _ptr(d_rgb_feat), _ptr(ws), int(dropout[0]) & 0xffffffff, float(dropout[1]))
        else:
            _launch(_lib.lib().nf_gnt_bwd, 'nf_gnt_bwd', d_rgb, _ptr(blob), _ptr(ray_diff), _ptr(mask), _ptr(d_rgb), R, S, V, depth, _ptr(d_rgb_feat),
                    _ptr(ws))
    return d_rgb_feat
