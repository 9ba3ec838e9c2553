"""ctypes binding of libnerfool_hip.so (C ABI declared in include/nerfool_hip.h).

The product path has exactly one backend: the gfx950 library built in-tree by `__graft_entry__.build()`.
If it is missing, or a tensor is not on a GPU, the call raises -- there is no CPU or PyTorch fallback.
The test-suite points the same bindings at the CPU stand-in build of the kernel sources from OUTSIDE the package
(tests/host_harness/standin.py sets `_lib` / `_emulated`); nothing in this package does.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_uint32, c_void_p

# torch first: it ships its own libamdhip64 / libhsa-runtime64, and the library below must bind to THAT runtime (same SONAME:
# whichever is loaded first serves both).  Loading the system runtime first leaves torch without a visible device.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libnerfool_hip.so')
ABI_VERSION = 1

_P = c_void_p      # every device pointer travels as void*
_PROTOTYPES = {
    'nf_abi_version': (c_int, []),
    'nf_last_error': (c_char_p, []),
    'nf_device_cu_count': (c_int, []),
    'nf_sample_along_ray': (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P, _P, _P, _P]),
    'nf_points_from_depths': (c_int, [_P, _P, _P, c_int64, c_int, _P, _P]),
    'nf_camera_setup': (c_int, [_P, _P, c_int, _P, _P]),
    'nf_project_gather_fwd': (c_int, [_P, c_int64, _P, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int64,
                                      c_int64, c_int64, c_int64, _P, _P, _P, _P, _P]),
    'nf_pixel_mask': (c_int, [_P, c_int64, c_int, _P, _P]),
    'nf_project_gather_bwd': (c_int, [_P, c_int64, _P, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int64, c_int64,
                                      c_int64, c_int64, _P, _P]),
    'nf_project_gather_keys': (c_int, [_P, c_int64, _P, c_int, c_int, c_int, _P, _P, _P]),
    'nf_project_gather_bwd_sorted': (c_int, [_P, _P, _P, c_int64, _P, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64, _P, _P]),
    'nf_ibrnet_blob_floats': (c_int64, []),
    'nf_ibrnet_blob_entry': (c_int, [c_int, c_char_p, c_int, POINTER(c_int64), POINTER(c_int), POINTER(c_int),
                                     POINTER(c_int)]),
    'nf_ibrnet_workspace_floats': (c_int64, [c_int64, c_int, c_int, c_int]),
    'nf_ibrnet_fwd': (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P]),
    'nf_ibrnet_bwd': (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P]),
    'nf_ibrnet_mfma_blob_floats': (c_int64, []),
    'nf_ibrnet_pack_mfma': (c_int, [_P, _P]),
    'nf_ibrnet_mfma_supported': (c_int, [c_int, c_int]),
    'nf_ibrnet_rows_form': (c_int, [c_int]),
    'nf_ibrnet_sol_selected': (c_int, [c_int, c_int]),
    'nf_ibrnet_mfma_workspace_floats': (c_int64, [c_int64, c_int]),
    'nf_ibrnet_fwd_mfma': (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P]),
    'nf_ibrnet_bwd_mfma': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P]),
    'nf_ibrnet_fwd_mfma_gather': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, c_int, c_int, c_int64, c_int64, c_int64, c_int64,
                                          c_int64, c_int, c_int, c_int, _P, _P, _P, _P]),
    'nf_ibrnet_bwd_mfma_scatter': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, c_int64, c_int64,
                                           c_int64, c_int64, c_int, c_int, _P]),
    'nf_ibrnet_mfma_bf16_blob_floats': (c_int64, []),
    'nf_ibrnet_pack_mfma_bf16': (c_int, [_P, _P]),
    'nf_ibrnet_fwd_mfma_bf16': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P]),
    'nf_ibrnet_bwd_mfma_bf16': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P]),
    'nf_debug_mfma32': (c_int, [_P, _P, _P, _P, _P]),
    'nf_composite_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int64, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    'nf_composite_bwd': (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    'nf_sample_fine': (c_int, [_P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P]),
    'nf_sample_pdf': (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P, _P]),
    'nf_masked_mse_fwd': (c_int, [_P, _P, _P, c_int64, _P, _P, _P]),
    'nf_masked_mse_bwd': (c_int, [_P, _P, _P, c_int64, _P, _P, _P, _P]),
    'nf_gnt_blob_floats': (c_int64, [c_int]),
    'nf_gnt_blob_entry': (c_int, [c_int, c_int, c_char_p, c_int, POINTER(c_int64), POINTER(c_int), POINTER(c_int),
                                  POINTER(c_int)]),
    'nf_gnt_workspace_floats': (c_int64, [c_int64, c_int, c_int, c_int, c_int]),
    'nf_gnt_fwd': (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    'nf_gnt_bwd': (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P]),
    'nf_gnt_fwd_train': (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, _P, _P, _P, c_uint32, c_double, _P]),
    'nf_gnt_bwd_train': (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P, c_uint32, c_double, _P]),
    'nf_gnt_mfma_blob_floats': (c_int64, [c_int]),
    'nf_gnt_pack_mfma': (c_int, [c_int, _P, _P]),
    'nf_gnt_mfma_supported': (c_int, [c_int, c_int]),
    'nf_gnt_fwd_mfma': (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    'nf_gnt_bwd_mfma': (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P]),
    'nf_gnt_fwd_train_mfma': (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, _P, _P, _P, c_uint32, c_double, _P, _P]),
    'nf_gnt_bwd_train_mfma': (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P, c_uint32, c_double, _P, _P]),
    'nf_conv1x1_pack_floats': (c_int64, [c_int, c_int]),
    'nf_conv1x1_pack': (c_int, [_P, c_int, c_int, c_int, _P]),
    'nf_conv1x1': (c_int, [_P, _P, _P, c_int64, c_int64, c_int64, c_int64, _P, c_int64, c_int64, c_int64, c_int64, c_int, c_int,
                           c_int, c_int, c_int, _P, c_int, _P]),
    'nf_pad_gather_fwd': (c_int, [_P, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                  c_int64, c_int64, _P]),
    'nf_pad_gather_bwd': (c_int, [_P, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int64, c_int64,
                                  c_int64, c_int64, _P]),
    'nf_upsample2x_pad_bwd': (c_int, [_P, c_int64, c_int, c_int, c_int, _P, c_int64, c_int64, _P]),
    'nf_conv_s2_pack_floats': (c_int64, [c_int, c_int, c_int, c_int]),
    'nf_conv_s2_pack': (c_int, [_P, c_int, c_int, c_int, c_int, _P]),
    'nf_conv_s2_fwd': (c_int, [_P, c_int, _P, c_int64, c_int64, c_int64, c_int, c_int, _P, c_int64, c_int64, c_int64, c_int, c_int, c_int,
                               c_int, c_int, _P]),
    'nf_conv_s2_bwd': (c_int, [_P, c_int, _P, c_int64, c_int64, c_int64, c_int, c_int, _P, c_int64, c_int64, c_int64, c_int, c_int, c_int,
                               c_int, c_int, _P]),
    'nf_conv_s2_x3_pack_floats': (c_int64, [c_int, c_int, c_int]),
    'nf_conv_s2_x3_pack': (c_int, [_P, c_int, c_int, c_int, _P]),
    'nf_conv_s2_fwd_x3': (c_int, [_P, _P, c_int64, c_int64, c_int64, c_int, c_int, _P, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int,
                                  c_int, _P]),
    'nf_conv_s2_stem_x3_pack_floats': (c_int64, [c_int]),
    'nf_conv_s2_stem_x3_pack': (c_int, [_P, c_int, c_int, _P]),
    'nf_conv_s2_stem_fwd_x3': (c_int, [_P, _P, c_int64, c_int64, c_int64, c_int, c_int, _P, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int,
                                       c_int, _P]),
    'nf_conv_s2_bwd_x3': (c_int, [_P, _P, c_int64, c_int64, c_int64, c_int, c_int, _P, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int,
                                  c_int, _P]),
    'nf_wino_pack_floats': (c_int64, [c_int, c_int, c_int]),
    'nf_wino_pack': (c_int, [_P, c_int, c_int, c_int, c_int, _P]),
    'nf_conv3x3_wino': (c_int, [_P, c_int, _P, c_int64, c_int64, c_int64, c_int, c_int, c_int, _P, c_int64, c_int64, c_int64, c_int,
                                c_int, c_int, c_int, c_int, c_int, _P]),
    'nf_wino_bf_pack_floats': (c_int64, [c_int, c_int, c_int, c_int]),
    'nf_wino_bf_pack': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P]),
    'nf_conv3x3_wino_bf': (c_int, [_P, c_int, c_int, _P, c_int64, c_int64, c_int64, c_int, c_int, c_int, _P, c_int64, c_int64, c_int64, c_int,
                                   c_int, c_int, c_int, c_int, _P]),
    'nf_wino_ring_pack_floats': (c_int64, [c_int, c_int]),
    'nf_wino_ring_pack': (c_int, [_P, c_int, c_int, _P]),
    'nf_conv3x3_bwd_ring': (c_int, [_P, _P, c_int64, c_int64, c_int64, c_int, c_int, _P, c_int64, c_int64, c_int64, c_int, c_int, c_int,
                                    c_int, _P]),
    'nf_in_act_pad_fwd': (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_float, _P, c_int64, c_int64, c_int64, c_int64,
                                  c_int, c_int, _P, c_int64, _P, _P, _P, _P]),
    'nf_in_act_pad_bwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, c_int64,
                                  _P, _P]),
    'nf_upsample2x_pad_fwd': (c_int, [_P, c_int64, c_int64, c_int64, c_int, c_int, c_int, _P, _P]),
    'nf_legacy_choice': (c_int, [_P, _P, c_int64, c_int64, _P, _P]),
    'nf_project_perturb': (c_int, [_P, _P, c_int64, c_float, c_float, c_float, _P]),
    'nf_pgd_adam_step': (c_int, [_P, _P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_float,
                                 c_float, c_float, c_float, _P]),
    'nf_pgd_adam_step_dev': (c_int, [_P, _P, _P, _P, _P, c_int64, _P, c_float, c_float, c_float, c_float, c_float, c_float, c_float, _P]),
    'nf_pgd_sign_step': (c_int, [_P, _P, _P, c_int64, c_float, c_float, c_float, c_float, _P]),
}

EXPORTED_SYMBOLS = tuple(_PROTOTYPES)

_lib = None
_emulated = False


def bind(cdll):
    """Attach argtypes/restype for every symbol of include/nerfool_hip.h; raises AttributeError on a missing one."""
    for name, (res, args) in _PROTOTYPES.items():
        fn = getattr(cdll, name)
        fn.restype = res
        fn.argtypes = args
    return cdll


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('nerfool_amd: %s is missing -- build it with `python -c "import __graft_entry__ as g; '
                               'g.build()"` (hipcc --offload-arch=gfx950); there is no CPU fallback' % LIB_PATH)
        handle = bind(ctypes.CDLL(LIB_PATH))
        if handle.nf_abi_version() != ABI_VERSION:
            raise RuntimeError('nerfool_amd: ABI mismatch (library %d, binding %d)' % (handle.nf_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def emulated():
    return _emulated


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed: %s' % (what, lib().nf_last_error().decode()))
