"""Masked MSE criterion (ibrnet/criterion.py:18-33 -> utils.py:48-58): sum((x-y)^2 * mask) / (sum(mask)*3 + 1e-6)."""
import torch
import torch.nn as nn

from .. import ops


class _MaskedMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, gt, mask, cnt_override):
        out, pm = ops.masked_mse_fwd(rgb, gt, mask, cnt_override)
        ctx.save_for_backward(rgb.contiguous(), gt.contiguous(), pm if pm is not None else torch.empty(0),
                              out[2:3] if cnt_override is None else cnt_override)
        ctx.has_mask = pm is not None
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, d_loss, _d_out):
        rgb, gt, pm, cnt = ctx.saved_tensors
        d_rgb = ops.masked_mse_bwd(rgb, gt, pm if ctx.has_mask else None, cnt, d_loss.contiguous())
        return d_rgb, None, None, None


def img2mse(x, y, mask=None, global_mask_count=None):
    """`global_mask_count` (device tensor [1]) replaces sum(mask) when the ray batch is sharded over ranks.
    Without a mask the plain mean over the 3R elements is returned (utils.py:55-56)."""
    if mask is None and global_mask_count is None:
        global_mask_count = torch.full((1,), float(x.shape[0]), dtype=torch.float32, device=x.device)
    loss, _ = _MaskedMSE.apply(x, y, mask, global_mask_count)
    return loss


def mse2psnr(x):
    import numpy as np
    return -10. * np.log(x + 1e-6) / np.log(10.)


class Criterion(nn.Module):
    def forward(self, outputs, ray_batch, scalars_to_log=None, global_mask_count=None):
        loss = img2mse(outputs['rgb'], ray_batch['rgb'], outputs['mask'], global_mask_count)
        return loss, scalars_to_log
