"""CPU: the hand-derived IBRNet backward (oracle/ibrnet_manual_bwd.py, blueprint of the HIP backward kernels)
against autograd of the oracle forward, on the golden inputs of the reference."""
import pytest
import torch

from fixtures import Golden, assert_close
from oracle import ibrnet_manual_bwd as mb
from oracle import ibrnet_ref as ib


@pytest.mark.parametrize('case', ['ibrnet_tiny_invu', 'ibrnet_tiny_lin_white', 'ibrnet_tiny_noaa_v5'])
def test_manual_backward_matches_autograd(case):
    g = Golden(case)
    cfg = g.stage_cfg()
    p = g.params('coarse')
    if not cfg['anti_alias_pooling']:
        p.pop('s', None)
    rgb_feat = g.t('coarse/rgb_feat').requires_grad_(True)
    ray_diff, mask = g.t('coarse/ray_diff'), g.t('coarse/mask')
    raw = ib.ibrnet_forward(p, rgb_feat, ray_diff, mask, cfg['anti_alias_pooling'])
    gen = torch.Generator().manual_seed(5)
    d_raw = torch.randn(raw.shape, generator=gen)
    ref, = torch.autograd.grad(raw, rgb_feat, d_raw)
    raw2, sv = mb.forward_saved(p, rgb_feat.detach(), ray_diff, mask, cfg['anti_alias_pooling'])
    assert_close(raw2, raw, 1e-5, 1e-6, 'forward_saved')
    mine = mb.backward_rgb_feat(p, sv, d_raw)
    assert_close(mine, ref, 1e-3, 1e-5 * float(ref.abs().max()), 'manual d rgb_feat')


@pytest.mark.parametrize('case', ['gnt_tiny_d2_v4', 'gnt_tiny_d3_v5'])
def test_gnt_manual_backward_matches_autograd(case):
    from oracle import gnt_manual_bwd as gmb
    from oracle import gnt_ref as gr
    g = Golden(case)
    depth = int(g.np('cfg')[5])
    p = g.params('net')
    rgb_feat = g.t('net_in/rgb_feat').requires_grad_(True)
    args = (g.t('net_in/ray_diff'), g.t('net_in/mask'), g.t('net_in/pts'), g.t('in/ray_d'))
    rgb = gr.gnt_forward(p, rgb_feat, *args, depth)
    d_rgb = torch.randn(rgb.shape, generator=torch.Generator().manual_seed(4))
    ref, = torch.autograd.grad(rgb, rgb_feat, d_rgb)
    rgb2, sv = gmb.forward_saved(p, rgb_feat.detach(), *args, depth)
    assert_close(rgb2, rgb, 1e-5, 1e-6, 'GNT forward_saved')
    mine = gmb.backward_rgb_feat(p, sv, args[1], d_rgb, depth)
    assert_close(mine, ref, 1e-3, 1e-5 * float(ref.abs().max()), 'GNT manual d rgb_feat')
