"""A/B of the split backward-data pass (Winograd interior + border-ring kernel) against the one-launch form on the padded output:
usage: python tools/experimental/ab_split.py <min saving, e.g. 0.10 | none> [bench flags]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nerfool_amd import ops
mode = sys.argv[1]
if mode == 'none':
    ops.wino_bwd_split_plan = lambda H, W: None
else:
    frac = float(mode)
    orig = ops.wino_bwd_split_plan

    def plan(H, W):
        rows = min(H + 1, -(-H // 8) * 8)
        cols = min(W + 1, -(-W // 16) * 16)
        if ops._wino_blocks(rows, cols) > (1.0 - frac) * ops._wino_blocks(H + 2, W + 2):
            return None
        return orig(H, W) or (rows, cols, 1 | (0 if rows == H + 1 else 2) | 4 | (0 if cols == W + 1 else 8))
    ops.wino_bwd_split_plan = plan
sys.argv = ['bench.py'] + sys.argv[2:]
import bench
bench.main()
