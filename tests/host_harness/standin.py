"""TEST HOOK (lives with the tests, not in the product package): point nerfool_amd's ctypes bindings at another build of the
kernel sources -- the CPU stand-in of this directory (`emulated=True`: CPU tensors are then accepted by the wrappers) or a
GPU tuning variant built by tools/build_variant.sh (`emulated=False`).  The product itself only ever loads
nerfool_amd/libnerfool_hip.so and raises without it."""
import ctypes
import os
import subprocess

HARNESS = os.path.dirname(os.path.abspath(__file__))
EMU_LIB = os.path.join(HARNESS, 'libnerfool_emu.so')
CLANG = '/opt/rocm/lib/llvm/bin/clang++'


def use_library(path, emulated=True):
    from nerfool_amd import _lib
    _lib._lib = _lib.bind(ctypes.CDLL(path))
    _lib._emulated = bool(emulated)
    return _lib._lib


def build_emulated():
    """(re)build the CPU stand-in when its sources are newer; returns the library path"""
    subprocess.run([os.path.join(HARNESS, 'build.sh')], check=True, capture_output=True)
    return EMU_LIB


def install_emulated(build=False):
    if build or not os.path.exists(EMU_LIB):
        build_emulated()
    return use_library(EMU_LIB, emulated=True)
