"""CPU: the torch restatement of tf.image.psnr / tf.image.ssim used by nerfool_amd/eval_views.py against a direct numpy
evaluation of the published definition (11x11 Gaussian, sigma 1.5, VALID windows)."""
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
