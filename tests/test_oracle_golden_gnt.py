"""CPU: the GNT oracle (oracle/gnt_ref.py) replayed on the golden vectors captured from the reference's gnt/ package."""
import numpy as np
import pytest
import torch

from fixtures import Golden, assert_close
from oracle import gnt_ref as gr

GNT_CASES = ['gnt_tiny_d2_v4', 'gnt_tiny_d3_v5', 'gnt_c4_d8_v10']      # the last: BASELINE config 4's network shape


def gnt_batch(g, device='cpu'):
    return {'ray_o': g.t('in/ray_o', device), 'ray_d': g.t('in/ray_d', device), 'rgb': g.t('in/gt_rgb', device),
            'camera': g.t('in/camera', device), 'depth_range': g.t('in/depth_range', device),
            'src_rgbs': g.t('in/src_rgbs', device), 'src_cameras': g.t('in/src_cameras', device)}


@pytest.mark.parametrize('case', GNT_CASES)
def test_gnt_network_and_renderer_match_reference(case):
    g = Golden(case)
    H, W, V, R, S, depth, Hf, Wf = [int(x) for x in g.np('cfg')]
    p = g.params('net')
    rgb = gr.gnt_forward(p, g.t('net_in/rgb_feat'), g.t('net_in/ray_diff'), g.t('net_in/mask'), g.t('net_in/pts'),
                         g.t('in/ray_d'), depth)
    assert_close(rgb, g.np('rgb'), 1e-4, 1e-5, 'GNT rgb (network only)')
    fm = g.t('in/featmap').requires_grad_(True)
    rb = gnt_batch(g)
    ret = gr.render_rays(rb, p, (fm, fm), S, depth, inv_uniform=True, det=True)
    assert ret['outputs_fine'] is None and ret['outputs_coarse']['weights'] is None
    assert_close(ret['outputs_coarse']['rgb'], g.np('rgb'), 1e-4, 2e-5, 'GNT rgb (render_rays)')
    loss = gr.criterion(ret['outputs_coarse'], rb)
    assert_close(loss, g.np('loss'), 1e-5, 1e-7, 'loss')
    grad, = torch.autograd.grad(loss, fm)
    ref = g.np('grad/featmap')
    assert_close(grad, ref, 1e-3, 1e-4 * float(np.abs(ref).max()), 'd loss / d featmap')


def test_gnt_ret_alpha_and_hierarchical_sampling_match_reference():
    """ret_alpha = True + N_importance > 0 with a single network (gnt/render_ray.py:249-277, transformer_network.py:196-200)."""
    g = Golden('gnt_alpha_d2_v3')
    H, W, V, R, S, depth, Hf, Wf, N_imp = [int(x) for x in g.np('cfg')]
    fm = g.t('in/featmap').requires_grad_(True)
    rb = gnt_batch(g)
    ret = gr.render_rays(rb, g.params('net'), (fm, fm), S, depth, inv_uniform=True, det=True, N_importance=N_imp, ret_alpha=True)
    for lvl in ('outputs_coarse', 'outputs_fine'):
        assert_close(ret[lvl]['rgb'], g.np(lvl + '/rgb'), 1e-4, 2e-5, lvl + ' rgb')
        assert_close(ret[lvl]['weights'], g.np(lvl + '/weights'), 1e-4, 1e-6, lvl + ' weights')
        assert_close(ret[lvl]['depth'], g.np(lvl + '/depth'), 1e-4, 1e-5, lvl + ' depth')
    loss = gr.criterion(ret['outputs_coarse'], rb) + gr.criterion(ret['outputs_fine'], rb)
    assert_close(loss, g.np('loss'), 1e-5, 1e-7, 'loss')
    grad, = torch.autograd.grad(loss, fm)
    ref = g.np('grad/featmap')
    assert_close(grad, ref, 1e-3, 1e-4 * float(np.abs(ref).max()), 'd loss / d featmap')
