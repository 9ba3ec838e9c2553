"""CPU: the kernel SOURCES of nerfool_amd/csrc compiled through the HIP stand-in (tests/host_harness) and driven through
the same C ABI and the same host layer as on the GPU.  Catches indexing / layout / math errors without a GPU; the
authoritative parity run is tests/test_gpu_parity.py on the MI355X."""
import os
import subprocess

import pytest

import parity_cases as pc
from fixtures import STAGE_CASES

HARNESS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'host_harness')
CLANG = '/opt/rocm/lib/llvm/bin/clang++'
TINY = [c for c in STAGE_CASES if c != 'ibrnet_medium']


@pytest.fixture(scope='module', autouse=True)
def emulated_library():
    if not os.path.exists(CLANG):
        pytest.skip('clang++ of the ROCm toolchain is needed to build the CPU stand-in')
    subprocess.run([os.path.join(HARNESS, 'build.sh')], check=True, capture_output=True)
    from nerfool_amd import _lib
    saved = (_lib._lib, _lib._emulated)
    _lib.use_library_for_tests(os.path.join(HARNESS, 'libnerfool_emu.so'))
    yield
    _lib._lib, _lib._emulated = saved


@pytest.mark.parametrize('case', TINY)
def test_stage_kernels(case):
    pc.check_stage_kernels(case, 'cpu')


@pytest.mark.parametrize('case', TINY)
def test_ibrnet_backward(case):
    pc.check_ibrnet_backward(case, 'cpu')


@pytest.mark.parametrize('case', TINY)
def test_gather_and_composite_backward(case):
    pc.check_gather_and_composite_backward(case, 'cpu')


@pytest.mark.parametrize('case', TINY)
def test_render_rays(case):
    pc.check_render_rays(case, 'cpu')


def test_ray_sampler():
    pc.check_ray_sampler('cpu')


def test_feature_net():
    pc.check_feature_net('cpu')


def test_init_perturb():
    pc.check_init_perturb('cpu')


def test_attack_steps():
    pc.check_attack_steps('cpu', free_steps=2)


def test_render_single_image():
    pc.check_render_single_image('cpu', rows=4)
