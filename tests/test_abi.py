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
