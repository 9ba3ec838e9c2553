#!/usr/bin/env python
"""Round-3 golden vectors from the REFERENCE (imported from /root/reference, see _refimport.py):

  metrics_r03.npz
    img/<i>/{pred,gt}            two seeded image pairs [H,W,3] in [0,1] (one odd-sized, one smaller than two windows)
    img/<i>/gnt_ssim             eval/gnt/utils.py:211-277 `ssim(pred, gt, format='HWC')` -- 11x11 Gaussian, zero-padded SAME
                                 windows, mean over the whole map (what eval/gnt/eval.py:233 logs)
    img/<i>/gnt_psnr             eval/gnt/utils.py:29,55-71 `img2psnr(pred, gt)` -- -10 log10(mse + 1e-6)
    pose/<j>/{p1,p2,p3}          seeded camera-to-world matrices (float64 [4,4]; rotations from scipy's random rotations)
    pose/<j>/s, pose/<j>/out     eval/ibrnet/geo_interp.py:44-45 `interp3(p1, p2, p3, s12, s3)`; s = [s12, s3] scalars, or
                                 [s12_rot, s12_trans, s3_rot, s3_trans] for the decoupled form (:17-23)
    pose/<j>/out_gnt             eval/gnt/geo_interp.py:37-38 (scalar s only)

  ibrnet_tiny_v10.npz  (python tests/golden/make_golden_r03.py v10)
    a stage capture like make_golden.py's (inputs, per-stage outputs, loss, gradients w.r.t. the feature maps) at the reference's
    DEFAULT view count, num_source_views = 10 (config.py:63), 16 coarse + 16 importance samples: the matrix-core kernels run it on 16
    lanes per sample with six padding lanes.  (seed 6: with seed 5 one ray's inverse-CDF draw sits within fp32 rounding of a cdf
    edge -- the discontinuity DESIGN section 2 describes -- which the strict per-stage check of the fine depths does not admit)

    python tests/golden/make_golden_r03.py [v10]

Data only; runs only in the build container."""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import _refimport  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    _refimport.install('gnt')
    gu = _load('ref_gnt_eval_utils', os.path.join(_refimport.REF_ROOT, 'eval', 'gnt', 'utils.py'))
    gi_ibr = _load('ref_ibrnet_geo_interp', os.path.join(_refimport.REF_ROOT, 'eval', 'ibrnet', 'geo_interp.py'))
    gi_gnt = _load('ref_gnt_geo_interp', os.path.join(_refimport.REF_ROOT, 'eval', 'gnt', 'geo_interp.py'))
    out = {}
    rng = np.random.RandomState(303)
    for i, (H, W, noise) in enumerate(((37, 52, 0.05), (14, 19, 0.2))):
        gt = rng.rand(H, W, 3).astype(np.float32)
        # smooth the ground truth a little so the structure term is not pure noise
        gt = (0.5 * gt + 0.25 * np.roll(gt, 1, 0) + 0.25 * np.roll(gt, 1, 1)).astype(np.float32)
        pred = np.clip(gt + noise * rng.randn(H, W, 3), 0.0, 1.0).astype(np.float32)
        p, g = torch.from_numpy(pred), torch.from_numpy(gt)
        out['img/%d/pred' % i], out['img/%d/gt' % i] = pred, gt
        out['img/%d/gnt_ssim' % i] = np.float64(gu.ssim(p, g, format='HWC').item())
        out['img/%d/gnt_psnr' % i] = np.float64(gu.img2psnr(p, g))
    from scipy.spatial.transform import Rotation
    rots = Rotation.random(9, random_state=7).as_matrix()
    for j in range(3):
        poses = []
        for k in range(3):
            m = np.eye(4)
            m[:3, :3] = rots[3 * j + k]
            m[:3, 3] = rng.randn(3)
            poses.append(m)
        decoupled = j == 2
        s = rng.uniform(0, 1, size=4 if decoupled else 2)
        for k, m in enumerate(poses):
            out['pose/%d/p%d' % (j, k + 1)] = m
        out['pose/%d/s' % j] = s
        if decoupled:
            out['pose/%d/out' % j] = gi_ibr.interp3(poses[0], poses[1], poses[2], [s[0], s[1]], [s[2], s[3]]).numpy()
        else:
            out['pose/%d/out' % j] = gi_ibr.interp3(poses[0], poses[1], poses[2], s[0], s[1]).numpy()
            out['pose/%d/out_gnt' % j] = gi_gnt.interp3(poses[0], poses[1], poses[2], s[0], s[1]).numpy()
    np.savez_compressed(os.path.join(HERE, 'metrics_r03.npz'), **out)
    for k in sorted(out):
        if out[k].ndim == 0:
            print(k, float(out[k]))


if __name__ == '__main__':
    if sys.argv[1:] == ['v10']:
        sys.path.insert(0, os.path.dirname(HERE))
        import make_golden as mg                      # the IBRNet flavour of the reference (its own _refimport.install)
        mg.stage_case('ibrnet_tiny_v10', 32, 48, 10, 16, 16, 16, True, False, seed=6, tilt=0.4)
    else:
        main()
