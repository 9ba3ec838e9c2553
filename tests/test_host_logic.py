"""CPU: host-side logic that needs neither a GPU nor the kernels' stand-in build."""
import subprocess
import sys
import os
from types import SimpleNamespace

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene(seed=0, H=40, W=56, V=3, path=None):
    from nerfool_amd.synthetic import make_scene
    data = make_scene(H, W, V, seed=seed)
    data.pop('rgb_path', None)
    if path is not None:
        data['rgb_path'] = [path]
    return data


@pytest.mark.parametrize('with_path', [False, True])
def test_sampler_cache_keys_on_content_not_identity(with_path):
    """RaySamplerSingleImage.cached: a DataLoader hands out FRESH tensors for the same view every step of the reference's universal
    loop (eval/ibrnet/eval_adv.py:652-740 `for data in train_loader`) -- equal content must hit, an edit anywhere in an image must
    miss, with or without an `rgb_path` (every element enters the checksum; round 4's strided form under a path let a sparse
    in-place edit return the stale sampler)."""
    from nerfool_amd.ibrnet.sample_ray import RaySamplerSingleImage
    RaySamplerSingleImage._cache.clear()
    path = 'scene/images/007.png' if with_path else None
    a = _scene(path=path)
    s0 = RaySamplerSingleImage.cached(a, 'cpu')
    fresh = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in a.items()}
    assert RaySamplerSingleImage.cached(fresh, 'cpu') is s0
    assert RaySamplerSingleImage.cached(a, 'cpu') is s0                      # memoised per tensor object
    moved = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in a.items()}
    moved['camera'][0, 20] += 1e-3                                           # another target pose
    assert RaySamplerSingleImage.cached(moved, 'cpu') is not s0
    edited = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in a.items()}
    edited['src_rgbs'][0, 1, 5, 5, 1] += 1e-3                                # one element, same path
    assert RaySamplerSingleImage.cached(edited, 'cpu') is not s0
    dense = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in a.items()}
    dense['src_rgbs'] += 1e-3                                                # a perturbation of the whole image: seen on the stride too
    assert RaySamplerSingleImage.cached(dense, 'cpu') is not s0
    # an in-place edit of a tensor already seen bumps its version counter: no stale memo
    a['src_rgbs'][0, 0].mul_(0.5)
    assert RaySamplerSingleImage.cached(a, 'cpu') is not s0
    RaySamplerSingleImage._cache.clear()


def test_sampler_cache_key_accepts_slices_at_odd_float_offsets():
    """a contiguous slice whose storage offset is an odd number of floats (imgs[1:] with an odd number of floats per image) is a
    valid batch tensor: the checksum reads 32-bit words, which every float32 element is aligned to"""
    from nerfool_amd.ibrnet.sample_ray import RaySamplerSingleImage, _content_key
    RaySamplerSingleImage._cache.clear()
    a = _scene(H=41, W=57, V=3)                       # 41 * 57 * 3 = 7011 floats per image: odd
    stack = torch.cat([torch.zeros(1, 1, 41, 57, 3), a['src_rgbs'], a['rgb'][None]], dim=1)      # [1, 5, H, W, 3]
    view = dict(a, rgb=stack[:, 4], src_rgbs=stack[:, 1:4])
    assert (view['src_rgbs'].storage_offset() * 4) % 8 == 4 and view['src_rgbs'].is_contiguous()
    k = _content_key(view['src_rgbs'])
    assert k == _content_key(view['src_rgbs'].clone())
    s0 = RaySamplerSingleImage.cached(view, 'cpu')
    assert RaySamplerSingleImage.cached(a, 'cpu') is s0                      # same content, other storage
    RaySamplerSingleImage._cache.clear()


def test_gnt_training_mode_takes_consecutive_seeds():
    """the reference's universal GNT loop runs with Dropout(0.1) active (eval/gnt/eval_adv.py:739-878 before switch_to_eval at :959):
    a module in training mode runs the Dropout-active kernels, one seed per forward call, starting from torch's seed (the kernels
    themselves: tests/test_emu_parity.py::test_gnt_training_mode_dropout)"""
    from nerfool_amd.gnt.transformer_network import GNT
    torch.manual_seed(77)
    net = GNT(SimpleNamespace(netwidth=64, trans_depth=2), in_feat_ch=32, posenc_dim=63, viewenc_dim=63)
    assert net.training and net.dropout_p == 0.1 and net.dropout_seed is None
    assert net.next_dropout_seed() == 77 and net.next_dropout_seed() == 78
    net.dropout_seed = 0xffffffff
    assert net.next_dropout_seed() == 0xffffffff and net.dropout_seed == 0


def test_removed_environment_switches_warn_once():
    code = 'import warnings; warnings.simplefilter("error"); import nerfool_amd'
    env = dict(os.environ, PYTHONPATH=ROOT, NERFOOL_GATHER_FUSION='full')
    p = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
    assert p.returncode != 0 and 'NERFOOL_GATHER_FUSION' in p.stderr
    env.pop('NERFOOL_GATHER_FUSION')
    assert subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True).returncode == 0
