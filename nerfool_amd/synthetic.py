"""Seed-driven synthetic scenes with the reference loaders' batch schema.

There is no dataset or checkpoint on the build or GPU boxes, so bench.py, smoke() and the parity tests all
run on scenes made here: smooth random source images, pinhole cameras on an arc around the target camera
(OpenCV convention: x right, y down, z forward; `camera` = [H, W, K 4x4 row-major, c2w 4x4 row-major] as in
ibrnet/data_loaders/llff_test.py:116-117,160-161), depth_range [near, far].
"""
import math

import numpy as np
import torch


def _smooth_image(gen, H, W, channels=3, coarse=8, detail=0.15):
    """Low-pass noise (bilinear-upsampled coarse grid) plus a little fine detail, squashed into [0,1]."""
    h0, w0 = max(2, H // coarse), max(2, W // coarse)
    base = torch.rand(1, channels, h0, w0, generator=gen)
    img = torch.nn.functional.interpolate(base, size=(H, W), mode='bilinear', align_corners=True)[0]
    img = img + detail * (torch.rand(channels, H, W, generator=gen) - 0.5)
    return img.clamp_(0.0, 1.0).permute(1, 2, 0).contiguous()      # [H,W,C]


def _look_at(center, target, up=(0.0, -1.0, 0.0)):
    """c2w (4x4, float64) of an OpenCV-convention camera at `center` looking at `target`."""
    center = np.asarray(center, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - center
    fwd /= np.linalg.norm(fwd)
    upv = np.asarray(up, dtype=np.float64)
    right = np.cross(-upv, fwd)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, center
    return c2w


def pack_camera(H, W, K, c2w):
    return np.concatenate([[float(H), float(W)], np.asarray(K, dtype=np.float64).reshape(-1),
                           np.asarray(c2w, dtype=np.float64).reshape(-1)]).astype(np.float32)


def make_scene(H, W, V, seed=0, depth_range=(2.0, 6.0), baseline=0.3, focal=None, blender=False, tilt=0.0,
               push_forward=0.0):
    """Returns the loader-style batch dict (batch dim 1, CPU tensors).

    baseline: radius of the arc the V source cameras sit on (scene units); tilt: extra yaw (rad) applied to the
    LAST source camera so that part of the ray bundle leaves its frustum; push_forward: z offset of source camera 0
    (a value inside the depth range puts the near samples BEHIND that camera) -- both for edge-case coverage.
    """
    gen = torch.Generator().manual_seed(seed)
    rs = np.random.RandomState(seed)
    if focal is None:
        focal = 0.5 * W / math.tan(0.5 * 0.6911) if blender else 0.8 * W
    K = np.array([[focal, 0, W / 2.0, 0], [0, focal, H / 2.0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    look = np.array([0.0, 0.0, 0.5 * (depth_range[0] + depth_range[1])])
    tar_c2w = _look_at([0.0, 0.0, 0.0], look)
    cams = []
    for v in range(V):
        ang = 2.0 * math.pi * (v + 0.25) / V
        r = baseline * (0.6 + 0.4 * rs.rand())
        c = np.array([r * math.cos(ang), 0.6 * r * math.sin(ang), 0.15 * baseline * (rs.rand() - 0.5)])
        tgt = look + 0.2 * baseline * (rs.rand(3) - 0.5)
        if push_forward and v == 0:
            c = c + np.array([0.0, 0.0, push_forward])
            tgt = c + np.array([0.05, 0.02, 1.0])
        if tilt and v == V - 1:
            tgt = look + np.array([math.tan(tilt) * look[2], 0.0, 0.0])
        cams.append(pack_camera(H, W, K, _look_at(c, tgt)))
    data = {
        'rgb': _smooth_image(gen, H, W)[None],
        'camera': torch.from_numpy(pack_camera(H, W, K, tar_c2w))[None],
        'rgb_path': ['synthetic_%d' % seed],
        'src_rgbs': torch.stack([_smooth_image(gen, H, W) for _ in range(V)])[None],
        'src_cameras': torch.from_numpy(np.stack(cams))[None],
        'depth_range': torch.tensor([[float(depth_range[0]), float(depth_range[1])]]),
    }
    return data


def feature_map_size(H, W):
    """ResUNet output size: every stride-2 stage gives ceil(n/2); two x2 upsamplings."""
    def c2(n):
        return (n + 1) // 2
    return 4 * c2(c2(c2(c2(H)))), 4 * c2(c2(c2(c2(W))))


def smooth_featmaps(V, C, Hf, Wf, seed=0, amp=1.0):
    """Random but spatially smooth [V,C,Hf,Wf] feature maps for CNN-free parity cases."""
    gen = torch.Generator().manual_seed(seed + 7919)
    h0, w0 = max(2, Hf // 3), max(2, Wf // 3)
    base = torch.randn(V, C, h0, w0, generator=gen)
    fm = torch.nn.functional.interpolate(base, size=(Hf, Wf), mode='bilinear', align_corners=True)
    return (amp * (fm + 0.1 * torch.randn(V, C, Hf, Wf, generator=gen))).contiguous()


def target_views(data, n, seed=100, radius=0.15):
    """n TARGET views of one scene for the universal loop (eval_adv.py:634-740 cycles over a scene's training views while the
    perturbation lives on the shared source views): `data` itself, then copies whose target camera is moved on a small circle around it
    (camera-to-world translation, scene units) with their own smooth target images."""
    views = [data]
    for i in range(1, n):
        ang = 2.0 * math.pi * i / max(n - 1, 1)
        cam = data['camera'].clone()
        c2w = cam[0, 18:34].reshape(4, 4).clone()
        c2w[:3, 3] += torch.tensor([radius * math.cos(ang), radius * math.sin(ang), 0.02 * i], dtype=cam.dtype)
        cam[0, 18:34] = c2w.reshape(-1)
        H, W = int(cam[0, 0]), int(cam[0, 1])
        v = dict(data)
        v['camera'] = cam
        v['rgb'] = _smooth_image(torch.Generator().manual_seed(seed + i), H, W)[None]
        v['rgb_path'] = ['synthetic_target_%d' % i]
        views.append(v)
    return views
