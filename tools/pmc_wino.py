"""Driver for `rocprofv3 --pmc ... -- python3 tools/pmc_wino.py`: a run of back-to-back Winograd launches per layer shape
(counters per dispatch: SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, ...)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfool_amd import ops                                   # noqa: E402

dev = 'cuda'
torch.manual_seed(0)
for (ci, co, H, W) in ((64, 64, 189, 252), (256, 256, 48, 63)):
    x = torch.randn(4, ci, H + 2, W + 2, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    rf = ops.wino_pack(w, False, dev)
    for _ in range(200):
        y = ops.conv3x3_wino(rf, x, co, 0)
torch.cuda.synchronize()
