"""Projector: 3-D samples -> per-source-view colours, deep features, direction deltas and validity masks.

Mirrors ibrnet/projection.py:20-132 of the reference (class name, `compute` signature and return layout).  One fused
HIP kernel replaces the inverse/bmm/clamp/2x grid_sample/permute/cat chain; its backward scatter-adds into the feature
maps only, because in the attack the colour taps, `ray_diff` and the masks are constants."""
import torch

from .. import ops


class _ProjectGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, cam_ws, src_rgbs, featmaps):
        rgb_feat, ray_diff, mask, _ = ops.project_gather_fwd(xyz, cam_ws, src_rgbs, featmaps)
        ctx.save_for_backward(xyz, cam_ws)
        ctx.geom = (src_rgbs.shape[0], src_rgbs.shape[1], src_rgbs.shape[2], tuple(featmaps.shape))
        ctx.mark_non_differentiable(ray_diff, mask)
        ctx.set_materialize_grads(False)
        return rgb_feat, ray_diff, mask

    @staticmethod
    def backward(ctx, d_rgb_feat, _d_ray_diff, _d_mask):
        if d_rgb_feat is None:
            return None, None, None, None
        xyz, cam_ws = ctx.saved_tensors
        V, H, W, feat_shape = ctx.geom
        d_feat = ops.project_gather_bwd(xyz, cam_ws, V, H, W, d_rgb_feat, feat_shape)
        return None, None, None, d_feat


class Projector:
    def __init__(self, device=None):
        self.device = device

    def compute(self, xyz, query_camera, train_imgs, train_cameras, featmaps, cam_ws=None):
        """
        :param xyz: [n_rays, n_samples, 3]
        :param query_camera: [1, 34]   (H, W, K 4x4, c2w 4x4)
        :param train_imgs: [1, n_views, h, w, 3]
        :param train_cameras: [1, n_views, 34]
        :param featmaps: [n_views, d, hf, wf]   (any strides; channels-last is the fast layout)
        :param cam_ws: optional ops.camera_setup(query_camera, train_cameras) if the caller already has it
        :return: rgb_feat [n_rays, n_samples, n_views, 3+d], ray_diff [..., 4], mask [..., 1]
        """
        if not (train_imgs.shape[0] == 1 and train_cameras.shape[0] == 1 and query_camera.shape[0] == 1):
            raise AssertionError('only support batch_size=1 for now')
        R, S, _ = xyz.shape
        V = train_cameras.shape[1]
        if cam_ws is None:          # (render_rays hands in the workspace it built for the coarse level)
            cam_ws = ops.camera_setup(query_camera.detach(), train_cameras.detach())
        pts = xyz.detach().reshape(-1, 3)
        rgb_feat, ray_diff, mask = _ProjectGather.apply(pts, cam_ws, train_imgs[0].detach(), featmaps)
        C = featmaps.shape[1]
        rgb_feat = rgb_feat.view(R, S, V, 3 + C)
        if featmaps.requires_grad:
            # what the adjoint of this gather needs, for a consumer that fuses it into its own backward (IBRNet.forward: the scatter
            # then runs inside the row kernel and d rgb_feat is never written); anybody else differentiates rgb_feat as usual
            # (the version counter lets the consumer see an in-place edit of rgb_feat made after this point and fall back)
            rgb_feat._nf_gather = (pts, cam_ws, featmaps, rgb_feat._version)
        return rgb_feat, ray_diff.view(R, S, V, 4), mask.view(R, S, V, 1)

    def compute_projections(self, xyz, train_cameras):
        """pixel_locations [n_views, n_rays, n_samples, 2] (ibrnet/projection.py:42-62); the in-front flag is folded
        into `mask` by `compute` and is not returned separately here."""
        R, S, _ = xyz.shape
        V = train_cameras.shape[0]
        cam_ws = ops.camera_setup(train_cameras[0], train_cameras)
        dummy_rgb = torch.zeros(V, 2, 2, 3, dtype=torch.float32, device=xyz.device)
        dummy_feat = torch.zeros(V, 4, 2, 2, dtype=torch.float32, device=xyz.device)
        _, _, _, pix = ops.project_gather_fwd(xyz.reshape(-1, 3), cam_ws, dummy_rgb, dummy_feat, want_pix=True)
        return pix.view(V, R, S, 2)
