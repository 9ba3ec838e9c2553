"""CPU: the product library builds for gfx950, loads, and exports every symbol include/nerfool_hip.h declares (no compute
call is made: there is no GPU here); host-side logic of the binding."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built_library():
    import __graft_entry__ as entry
    if not os.path.exists('/opt/rocm/bin/hipcc'):
        pytest.skip('hipcc not available')
    return entry.build()


def test_header_symbols_are_exported(built_library):
    from nerfool_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'nerfool_hip.h')).read()
    declared = set(re.findall(r'\b(nf_[a-z0-9_]+)\s*\(', header))
    declared -= {'nf_stream_t'}
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    handle = ctypes.CDLL(built_library)
    for name in declared:
        assert hasattr(handle, name), name
    _lib.bind(handle)
    assert handle.nf_abi_version() == _lib.ABI_VERSION


def test_blob_layout_is_consistent(built_library):
    """Host-only ABI calls: the parameter blob covers every IBRNet parameter twice (transposed + native) plus biases."""
    from nerfool_amd import _lib
    handle = _lib.bind(ctypes.CDLL(built_library))
    total = handle.nf_ibrnet_blob_floats()
    name = ctypes.create_string_buffer(96)
    off, rows, cols, tr = ctypes.c_int64(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    idx, covered, keys = 0, 0, {}
    while handle.nf_ibrnet_blob_entry(idx, name, 96, ctypes.byref(off), ctypes.byref(rows), ctypes.byref(cols),
                                      ctypes.byref(tr)) == 0:
        assert off.value == covered, 'entries must tile the blob without gaps'
        covered += rows.value * cols.value
        keys.setdefault(name.value.decode(), []).append(bool(tr.value))
        idx += 1
    assert covered == total
    n_params = sum(r for r in (1,)) - 1
    from oracle.ibrnet_ref import random_ibrnet_params
    p = random_ibrnet_params(8, 0)
    p.pop('pos_encoding')
    assert set(keys) == set(p), set(keys) ^ set(p)
    assert sum(v.numel() for v in p.values()) == 20136       # 20 136 parameters per net
    for k, flags in keys.items():
        if k.endswith('.weight') and 'layer_norm' not in k:
            assert sorted(flags) == [False, True], k


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from nerfool_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, '_emulated', False)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.lib()


def test_cpu_tensor_is_rejected_without_gpu(monkeypatch):
    import torch
    from nerfool_amd import _lib, ops
    monkeypatch.setattr(_lib, '_emulated', False)
    with pytest.raises(RuntimeError, match='GPU only'):
        ops.sample_along_ray(torch.zeros(4, 3), torch.ones(4, 3), torch.tensor([[2., 6.]]), 8, True)


def test_legacy_choice_matches_numpy(built_library):
    """Host-only ABI call: nf_legacy_choice consumes an MT19937 stream exactly like RandomState.choice(replace=False)
    (the reference's pixel pick, ibrnet/sample_ray.py:149-171), picks and advanced generator state bit for bit."""
    import numpy as np
    from nerfool_amd import _lib
    from nerfool_amd.ibrnet import sample_ray
    if _lib._lib is None:
        from host_harness import standin
        standin.use_library(built_library, emulated=False)
    gen = np.random.RandomState(234)
    for pop, size in ((756 * 1008, 512), (1, 1), (2, 2), (97, 0), (4096, 4096), (604 * 806, 4096), (65537, 300)):
        for _ in range(3):
            state = gen.get_state()
            mine, after = sample_ray.legacy_choice(state, pop, size)
            want = gen.choice(pop, size=(size,), replace=False)
            assert mine.dtype == want.dtype and np.array_equal(mine, want), (pop, size)
            now = gen.get_state()
            assert after[2] == now[2] and np.array_equal(after[1], now[1]) and after[3:] == now[3:]
            gen.random_sample(5)        # move the position around between the draws, including across refills
    handle = _lib.lib()
    key = np.zeros(624, np.uint32)
    pos = ctypes.c_int32(0)
    out = np.zeros(4, np.int64)
    assert handle.nf_legacy_choice(key.ctypes.data, ctypes.byref(pos), 3, 4, out.ctypes.data, out.ctypes.data) != 0
    assert b'without replacement' in handle.nf_last_error()


def test_pixel_lookahead_keeps_the_stream(built_library):
    """The look-ahead draw is invisible in the RandomState(234) stream: same picks as plain draws, also when the next call
    asks for something else, when somebody reseeds in between, and when a queued pick is never consumed."""
    import numpy as np
    from nerfool_amd import _lib
    from nerfool_amd.ibrnet import sample_ray
    if _lib._lib is None:
        from host_harness import standin
        standin.use_library(built_library, emulated=False)
    plain = np.random.RandomState(234)
    sample_ray.rng.seed(234)
    for pop, size, ahead in ((5000, 64, True), (5000, 64, True), (5000, 64, True), (7000, 64, True), (7000, 32, False),
                             (7000, 32, True)):
        got = sample_ray._choice(pop, size, ahead)
        assert np.array_equal(got, plain.choice(pop, size=(size,), replace=False))
    assert sample_ray._ahead is not None
    sample_ray.rng.seed(234)                      # a queued pick from the old state must not survive a reseed
    plain.seed(234)
    assert np.array_equal(sample_ray._choice(7000, 32, True), plain.choice(7000, size=(32,), replace=False))
    assert np.array_equal(sample_ray.rng.random_sample(3), plain.random_sample(3))      # direct use while a pick is queued
    assert np.array_equal(sample_ray._choice(7000, 32, False), plain.choice(7000, size=(32,), replace=False))
    assert sample_ray._ahead is None
    assert np.array_equal(sample_ray.rng.get_state()[1], plain.get_state()[1])


def test_feature_cnn_has_no_cpu_fallback():
    """A CPU tensor (outside the CPU stand-in build the parity tests bind explicitly) or trainable weights raise instead of
    silently running the nn.Module graph."""
    import torch
    from nerfool_amd import _lib
    from nerfool_amd.ibrnet import feature_network as fn
    if _lib.emulated() or fn.CNN_PATH != 'fused':
        pytest.skip('stand-in build bound / torch graph selected in this process')
    net = fn.ResUNet(coarse_out_ch=32, fine_out_ch=32)
    for p in net.parameters():
        p.requires_grad_(False)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        net(torch.zeros(1, 3, 32, 32))
