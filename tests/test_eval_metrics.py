"""CPU: the torch restatement of tf.image.psnr / tf.image.ssim used by nerfool_amd/eval_views.py for the IBRNet flavour against a
direct numpy evaluation of the published definition (11x11 Gaussian, sigma 1.5, VALID windows); the GNT flavour's metrics
(SAME-padded SSIM, PSNR with +1e-6) and the unseen-view pose interpolation against values of the REFERENCE's own functions
(tests/golden/metrics_r03.npz, made by tests/golden/make_golden_r03.py from eval/gnt/utils.py and eval/*/geo_interp.py)."""
import os

import numpy as np
import torch

from nerfool_amd import eval_views as ev


def _ssim_numpy(x, y, max_val=1.0):
    size, sigma = 11, 1.5
    g = np.exp(-((np.arange(size) - 5.0) ** 2) / (2 * sigma ** 2))
    g = g / g.sum()
    w = np.outer(g, g)
    H, W, C = x.shape
    c1, c2 = (0.01 * max_val) ** 2, (0.03 * max_val) ** 2
    vals = []
    for c in range(C):
        acc = []
        for i in range(H - size + 1):
            for j in range(W - size + 1):
                a, b = x[i:i + size, j:j + size, c], y[i:i + size, j:j + size, c]
                mx, my = (w * a).sum(), (w * b).sum()
                sxx, syy, sxy = (w * a * a).sum() - mx * mx, (w * b * b).sum() - my * my, (w * a * b).sum() - mx * my
                acc.append(((2 * mx * my + c1) / (mx * mx + my * my + c1)) * ((2 * sxy + c2) / (sxx + syy + c2)))
        vals.append(np.mean(acc))
    return float(np.mean(vals))


def test_psnr_and_ssim_follow_the_tf_definitions():
    rng = np.random.RandomState(0)
    gt = rng.rand(24, 31, 3)
    pred = np.clip(gt + 0.05 * rng.randn(24, 31, 3), 0, 1)
    mse = np.mean((pred - gt) ** 2)
    assert abs(ev.psnr(torch.tensor(pred), torch.tensor(gt)) - (-10 * np.log10(mse))) < 1e-9
    assert abs(ev.ssim(torch.tensor(pred), torch.tensor(gt)) - _ssim_numpy(pred, gt)) < 1e-9
    assert abs(ev.ssim(torch.tensor(gt), torch.tensor(gt)) - 1.0) < 1e-12


def _golden():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'metrics_r03.npz'))


def test_gnt_flavour_metrics_match_the_reference_functions():
    """eval/gnt/utils.py:211-277 `ssim(format='HWC')` and :29,55-71 `img2psnr`, as eval/gnt/eval.py:233-234 calls them"""
    z = _golden()
    for i in range(2):
        pred, gt = torch.from_numpy(z['img/%d/pred' % i]), torch.from_numpy(z['img/%d/gt' % i])
        assert abs(ev.ssim(pred, gt, padding='same') - float(z['img/%d/gnt_ssim' % i])) < 1e-6
        assert abs(ev.psnr(pred, gt, tiny=1e-6) - float(z['img/%d/gnt_psnr' % i])) < 1e-6
        # and they are NOT the TF definitions the IBRNet flavour scores with
        assert abs(ev.ssim(pred, gt) - float(z['img/%d/gnt_ssim' % i])) > 1e-4
        assert abs(ev.psnr(pred, gt) - float(z['img/%d/gnt_psnr' % i])) > 1e-6


def test_pose_interpolation_matches_the_reference():
    """eval/ibrnet/geo_interp.py:44-45 interp3 (scalar and decoupled [rot, trans] parameters), eval/gnt/geo_interp.py:37-38"""
    from nerfool_amd import geo_interp as gi
    z = _golden()
    for j in range(3):
        p = [z['pose/%d/p%d' % (j, k)] for k in (1, 2, 3)]
        s = z['pose/%d/s' % j]
        if len(s) == 4:
            out = gi.interp3(p[0], p[1], p[2], [s[0], s[1]], [s[2], s[3]])
        else:
            out = gi.interp3(p[0], p[1], p[2], s[0], s[1])
            assert np.abs(out.numpy() - z['pose/%d/out_gnt' % j]).max() < 1e-12
        assert out.dtype == torch.float64 and np.abs(out.numpy() - z['pose/%d/out' % j]).max() < 1e-12
        R = out.numpy()[:3, :3]
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12
    # end points: s = 0 -> pose 1, s = 1 -> pose 2
    a, b = z['pose/0/p1'], z['pose/0/p2']
    assert np.abs(gi.interp(a, b, 0.0).numpy() - a).max() < 1e-9 and np.abs(gi.interp(a, b, 1.0).numpy() - b).max() < 1e-9


def test_unseen_camera_draws_follow_the_reference_order():
    """nerfool_amd.geo_interp.unseen_camera on numpy's global generator == oracle/attack_ref.unseen_camera_stream (the restated
    draw order of eval/ibrnet/eval_adv.py:652-691) for the three sampling variants"""
    from types import SimpleNamespace
    from nerfool_amd import geo_interp as gi
    from oracle import attack_ref as atk
    z = _golden()
    poses = [z['pose/%d/p%d' % (j, k)] for j in range(3) for k in (1, 2, 3)]
    camera = torch.arange(34, dtype=torch.float32)[None]
    base = dict(interp_upbound=0.9, interp_upbound_rot=0.7, interp_upbound_trans=0.4, decouple_interp_range=False,
                sample_based_on_depth=False, beta=0.5, temp=0.5)
    for variant in ({}, {'decouple_interp_range': True}, {'sample_based_on_depth': True},
                    {'sample_based_on_depth': True, 'decouple_interp_range': True}):
        a = SimpleNamespace(**dict(base, **variant))
        want = atk.unseen_camera_stream(a, poses, camera, 4, seed=5)
        np.random.seed(5)
        for w in want:
            got = gi.unseen_camera(a, poses, camera)
            assert got.shape == (1, 34) and got.dtype == torch.float32
            assert torch.equal(got, w)
