"""CPU: the kernel SOURCES of nerfool_amd/csrc compiled through the HIP stand-in (tests/host_harness) and driven through
the same C ABI and the same host layer as on the GPU.  Catches indexing / layout / math errors without a GPU; the
authoritative parity run is tests/test_gpu_parity.py on the MI355X."""
import os
import subprocess

import pytest

import parity_cases as pc
from fixtures import END_TO_END_ONLY, STAGE_CASES

HARNESS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'host_harness')
CLANG = '/opt/rocm/lib/llvm/bin/clang++'
TINY = [c for c in STAGE_CASES if c not in END_TO_END_ONLY]


@pytest.fixture(scope='module', autouse=True)
def emulated_library():
    if not os.path.exists(CLANG):
        pytest.skip('clang++ of the ROCm toolchain is needed to build the CPU stand-in')
    subprocess.run([os.path.join(HARNESS, 'build.sh')], check=True, capture_output=True)
    from nerfool_amd import _lib
    from nerfool_amd.ibrnet import mlp_network
    saved = (_lib._lib, _lib._emulated, mlp_network.KERNEL_PATH)
    _lib.use_library_for_tests(os.path.join(HARNESS, 'libnerfool_emu.so'))
    # emulating the matrix-core kernels costs ~30x the generic ones: the end-to-end cases below run on the generic
    # kernels, the MFMA kernels have their own (small) cases at the end of this file
    mlp_network.KERNEL_PATH = 'generic'
    from nerfool_amd.ibrnet import feature_network
    saved_cnn, feature_network.CNN_PATH = feature_network.CNN_PATH, 'torch'    # ditto for the fused CNN glue
    yield
    _lib._lib, _lib._emulated, mlp_network.KERNEL_PATH = saved
    feature_network.CNN_PATH = saved_cnn


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


def test_conv_s2():
    pc.check_conv_s2('cpu', shapes=((1, 64, 64, 3, 21, 25), (1, 3, 64, 7, 29, 41), (1, 24, 40, 3, 19, 23)))


def test_pad_glue():
    pc.check_pad_glue('cpu')


def test_conv3x3_wino4():
    pc.check_conv3x3_wino4('cpu', shapes=((1, 8, 32, 18, 34), (1, 20, 40, 9, 13)))


def test_gather_bwd_deterministic():
    pc.check_gather_bwd_deterministic('cpu')


def test_fused_cnn_glue():
    pc.check_fused_cnn_glue('cpu')


def test_init_perturb():
    pc.check_init_perturb('cpu')


def test_attack_steps():
    pc.check_attack_steps('cpu', free_steps=2)


def test_pseudo_gt():
    pc.check_pseudo_gt('cpu')


def test_universal_trajectory():
    pc.check_universal_trajectory('cpu', steps=2)      # both target views once; the full loop runs on the GPU


def test_gather_fused_forward():
    from nerfool_amd.ibrnet import mlp_network
    mlp_network.KERNEL_PATH = 'auto'          # the gather-fused forward belongs to the matrix-core kernels
    try:
        pc.check_gather_fused_forward('cpu', shapes=((3, 32, 4),))
    finally:
        mlp_network.KERNEL_PATH = 'generic'


def test_ragged_ray_batches():
    pc.check_ragged_ray_batches('cpu')


def test_hybrid_and_sample_pdf():
    pc.check_hybrid_and_sample_pdf('cpu')


@pytest.mark.parametrize('case', ['gnt_tiny_d2_v4', 'gnt_tiny_d3_v5'])
def test_gnt(case):
    pc.check_gnt(case, 'cpu')


def test_gnt_attack_step():
    pc.check_gnt_attack_step('cpu')


def test_render_single_image():
    pc.check_render_single_image('cpu', rows=4)


def test_mfma_kernels_match_generic_kernels():
    """matrix-core forward (emulated v_mfma_f32_32x32x2_f32) vs the generic kernel, ragged tile count, V = 4 and 2."""
    import torch
    from nerfool_amd import ops
    from oracle.ibrnet_ref import random_ibrnet_params
    for R, S, V in ((3, 10, 4), (2, 9, 2), (5, 32, 4), (3, 64, 2)):      # S = 32 / 64: per-ray part on MFMA too
        gen = torch.Generator().manual_seed(S)
        p = random_ibrnet_params(S, seed=3)
        blob = ops.pack_ibrnet_blob(p, 'cpu')
        mblob = ops.pack_ibrnet_mfma_blob(blob)
        rgb_feat = torch.randn(R, S, V, 35, generator=gen)
        rd = torch.randn(R, S, V, 4, generator=gen)
        rd[..., 3] = 1 - 0.05 * torch.rand(R, S, V, generator=gen)
        mask = (torch.rand(R, S, V, generator=gen) > 0.25).float()
        mask[0, :3] = 0
        args = (p['pos_encoding'], rgb_feat, rd, mask, True)
        a = ops.ibrnet_fwd(blob, *args)
        b, _ = ops.ibrnet_fwd_mfma(mblob, blob, *args)
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(a.abs().max()))
        d_raw = torch.randn(R, S, 4, generator=gen)
        ga = ops.ibrnet_bwd(blob, args[0], rgb_feat, rd, mask, d_raw, True)
        gb = ops.ibrnet_bwd_mfma(mblob, blob, args[0], rgb_feat, rd, mask, _, d_raw, True)
        assert float((ga - gb).abs().max()) <= 1e-4 * max(1.0, float(ga.abs().max()))


def test_render_rays_through_mfma_kernels():
    """end-to-end render_rays + loss + gradients with the matrix-core kernels selected (V = 4 case)."""
    from nerfool_amd.ibrnet import mlp_network
    mlp_network.KERNEL_PATH = 'auto'
    try:
        pc.check_render_rays('ibrnet_tiny_invu', 'cpu')       # default: stand-alone gather, scatter fused into the backward
    finally:
        mlp_network.KERNEL_PATH = 'generic'


def test_gnt_matrix_core_forward_matches_generic():
    pc.check_gnt_mfma_vs_generic('cpu', shapes=((2, 32, 3, 2),))


def test_gnt_ret_alpha_and_hierarchical_sampling():
    # shape-generic kernels here (4 s); the matrix-core kernels take 4 minutes to emulate on this case: their ret_alpha output
    # is compared with the generic one in test_gnt_matrix_core_forward_matches_generic, the full case runs on the GPU
    pc.check_gnt_alpha('cpu', kernel_path='generic')
