"""GNT criterion (gnt/criterion.py:5-22): masked MSE when the outputs carry a 'mask', plain MSE otherwise (GNT's do not)."""
import torch.nn as nn

from ..ibrnet.criterion import img2mse


class Criterion(nn.Module):
    def forward(self, outputs, ray_batch, scalars_to_log=None, global_count=None):
        mask = outputs['mask'] if 'mask' in outputs else None
        return img2mse(outputs['rgb'], ray_batch['rgb'], mask, global_count), scalars_to_log
