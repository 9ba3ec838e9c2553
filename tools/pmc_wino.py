import os, sys, torch
sys.path.insert(0, '/root/repo')
from nerfool_amd import ops
dev='cuda'
torch.manual_seed(0)
for (ci, co, H, W) in ((64,64,189,252),(256,256,48,63)):
    x = torch.randn(4, ci, H+2, W+2, device=dev); w = torch.randn(co, ci, 3, 3, device=dev)*0.05
    rf = ops.wino_pack(w, False, dev)
    for _ in range(3):
        y = ops.conv3x3_wino(rf, x, co, 0)
torch.cuda.synchronize()
