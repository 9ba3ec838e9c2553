"""CPU: the kernel SOURCES of nerfool_amd/csrc compiled through the HIP stand-in (tests/host_harness) and driven through
the same C ABI and the same host layer as on the GPU.  Catches indexing / layout / math errors without a GPU; the
authoritative parity run is tests/test_gpu_parity.py on the MI355X."""
import contextlib
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
    from host_harness import standin
    standin.use_library(os.path.join(HARNESS, 'libnerfool_emu.so'))
    # the product's default dispatch: matrix-core kernels (emulated MFMA) and the fused CNN executor, exactly what runs on the GPU
    yield
    _lib._lib, _lib._emulated, mlp_network.KERNEL_PATH = saved


@contextlib.contextmanager
def fast_paths():
    """shape-generic IBRNet kernels and the nn.Module CNN graph for the multi-step loops: emulating the matrix-core kernels and
    the fused executor (and evaluating the float64 oracle their check needs) costs minutes there; one full step on the product's
    own dispatch is test_attack_step_on_the_product_dispatch, everything else in this file runs on it anyway"""
    from nerfool_amd.ibrnet import feature_network, mlp_network
    saved = (mlp_network.KERNEL_PATH, feature_network.CNN_PATH)
    mlp_network.KERNEL_PATH, feature_network.CNN_PATH = 'generic', 'torch'
    try:
        yield
    finally:
        mlp_network.KERNEL_PATH, feature_network.CNN_PATH = saved


@contextlib.contextmanager
def wino_fp32_operands():
    """the GNT attack tests below exercise the GNT kernels; their feature CNN runs on the fp32-operand Winograd kernel, which the
    stand-in emulates three times faster than the bf16x3 default (that form has its own tests: test_fused_cnn_glue, test_feature_net,
    test_attack_step_on_the_product_dispatch)"""
    from nerfool_amd.ibrnet import feature_network
    saved, feature_network.WINO_OPERANDS = feature_network.WINO_OPERANDS, 'fp32'
    try:
        yield
    finally:
        feature_network.WINO_OPERANDS = saved


@pytest.mark.parametrize('case', TINY)
def test_stage_kernels(case):
    pc.check_stage_kernels(case, 'cpu')


@pytest.mark.parametrize('case', TINY)
def test_row_kernel_forms(case):
    pc.check_row_kernel_forms(case, 'cpu')


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


def test_gather_bwd_deterministic():
    pc.check_gather_bwd_deterministic('cpu')


def test_fused_cnn_glue():
    pc.check_fused_cnn_glue('cpu')


def test_init_perturb():
    pc.check_init_perturb('cpu')


def test_attack_steps():
    with fast_paths():
        pc.check_attack_steps('cpu', free_steps=2)


def test_attack_step_late_in_the_trajectory():
    """the reference's state at iteration 99 of a 100-iteration attack (clamps active): loss, gradient, fused update"""
    with fast_paths():
        pc.check_attack_steps_late('cpu', iters=[99])


def test_attack_step_on_the_product_dispatch():
    """one teacher-forced PGD step through exactly what runs on the GPU -- fused CNN executor, matrix-core IBRNet kernels with
    the scatter fused into the backward -- with the float64 check on the ReLU pattern of this evaluation"""
    pc.check_attack_steps('cpu', free_steps=0, forced_steps=1)


def test_pseudo_gt():
    with fast_paths():
        pc.check_pseudo_gt('cpu')


def test_unseen_views():
    with fast_paths():
        pc.check_unseen_views('cpu')


def test_universal_trajectory():
    with fast_paths():
        pc.check_universal_trajectory('cpu', steps=2)      # both target views once; the full loop runs on the GPU


def test_bf16_row_network_config5():
    """BASELINE config 5's shape (V 8, 128 + 256 samples) through the emulated v_mfma_f32_32x32x16_bf16 rows, fused and stand-alone scatter"""
    pc.check_bf16_config5('cpu')


def test_bf16_attack_steps_and_universal_loop():
    """the bf16 row network (emulated v_mfma_f32_32x32x16_bf16) inside PGD steps; the CNN on the nn.Module graph"""
    from nerfool_amd.ibrnet import feature_network
    saved, feature_network.CNN_PATH = feature_network.CNN_PATH, 'torch'
    try:
        pc.check_bf16_attack('cpu')
    finally:
        feature_network.CNN_PATH = saved


def test_gather_fused_forward():
    pc.check_gather_fused_forward('cpu', shapes=((3, 32, 4), (2, 16, 5), (2, 32, 10)))


def test_ragged_ray_batches():
    pc.check_ragged_ray_batches('cpu')


def test_hybrid_and_sample_pdf():
    pc.check_hybrid_and_sample_pdf('cpu')


@pytest.mark.parametrize('case', ['gnt_tiny_d2_v4', 'gnt_tiny_d3_v5'])
def test_gnt(case):
    pc.check_gnt(case, 'cpu')


def test_gnt_config4_shape_on_the_matrix_core_kernels():
    """BASELINE config 4's network shape (depth 8, V 10, S 64) through the emulated matrix-core GNT kernels against the reference capture"""
    pc.check_gnt('gnt_c4_d8_v10', 'cpu', expect_mfma=True)


def test_gnt_attack_step():
    with wino_fp32_operands():
        pc.check_gnt_attack_step('cpu')


def test_render_single_image():
    pc.check_render_single_image('cpu', rows=4)


def test_mfma_kernels_match_generic_kernels():
    """matrix-core forward / backward (emulated v_mfma_f32_32x32x2_f32) vs the generic kernels: ragged tile counts, V = 4 and 2,
    and view counts that are not a power of two -- 3, 5, 10 (the reference's default num_source_views), 12 -- whose samples sit
    on 4 / 8 / 16 lanes with neutral padding lanes; a sample whose views are all masked (softmax over equal -1e9 logits)."""
    import torch
    from nerfool_amd import ops
    from oracle.ibrnet_ref import random_ibrnet_params
    for R, S, V in ((3, 10, 4), (2, 9, 2), (5, 32, 4), (3, 64, 2), (3, 11, 3), (2, 7, 5), (3, 32, 10), (2, 5, 12), (1, 3, 1)):      # S = 32 / 64: per-ray part on MFMA too
        gen = torch.Generator().manual_seed(S)
        p = random_ibrnet_params(S, seed=3)
        blob = ops.pack_ibrnet_blob(p, 'cpu')
        mblob = ops.pack_ibrnet_mfma_blob(blob)
        rgb_feat = torch.randn(R, S, V, 35, generator=gen)
        rd = torch.randn(R, S, V, 4, generator=gen)
        rd[..., 3] = 1 - 0.05 * torch.rand(R, S, V, generator=gen)
        mask = (torch.rand(R, S, V, generator=gen) > 0.25).float()
        mask[0, :3] = 0                  # every view of these samples masked
        args = (p['pos_encoding'], rgb_feat, rd, mask, True)
        a = ops.ibrnet_fwd(blob, *args)
        b, _ = ops.ibrnet_fwd_mfma(mblob, blob, *args)
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(a.abs().max()))
        d_raw = torch.randn(R, S, 4, generator=gen)
        ga = ops.ibrnet_bwd(blob, args[0], rgb_feat, rd, mask, d_raw, True)
        gb = ops.ibrnet_bwd_mfma(mblob, blob, args[0], rgb_feat, rd, mask, _, d_raw, True)
        assert float((ga - gb).abs().max()) <= 1e-4 * max(1.0, float(ga.abs().max()))


def test_render_rays_through_generic_kernels():
    """end-to-end render_rays + loss + gradients with the shape-generic kernels forced (the default dispatch above took the
    matrix-core ones for this V = 4 case)."""
    with fast_paths():
        pc.check_render_rays('ibrnet_tiny_invu', 'cpu')


def test_gnt_matrix_core_forward_matches_generic():
    pc.check_gnt_mfma_vs_generic('cpu', shapes=((2, 32, 3, 2),))


def test_gnt_attack_gradient_on_the_matrix_core_kernels():
    with wino_fp32_operands():
        pc.check_gnt_attack_gradient_kernel_paths('cpu', shapes=((4, 32, 3, 2),))


def test_gnt_ret_alpha_and_hierarchical_sampling():
    # shape-generic kernels here (4 s); the matrix-core kernels take 4 minutes to emulate on this case: their ret_alpha output
    # is compared with the generic one in test_gnt_matrix_core_forward_matches_generic, the full case runs on the GPU
    pc.check_gnt_alpha('cpu', kernel_path='generic')


def test_gnt_training_mode_dropout():
    """the Dropout-active GNT forward / backward of the reference's universal loop, through the emulated shape-generic kernels"""
    pc.check_gnt_train_mode('cpu')


def test_gnt_training_mode_dropout_on_the_matrix_core_kernels():
    """the Dropout-active network on the matrix-core kernels (emulated MFMA) against the reference capture with the same masks, 32
    samples per ray -- exact part only (the statistics over hundreds of seeds are the GPU test's)"""
    with wino_fp32_operands():
        pc.check_gnt_train_mode('cpu', 'gnt_train_mfma_d2', expect_mfma=True, stat_draws=2)


def test_gnt_universal_loop_in_training_mode():
    """a GNT attack step and the universal loop with the model left in training mode (the reference's eval/gnt/eval_adv.py:739-878)"""
    # (the Dropout sites live in the GNT kernels: the feature CNN runs as the plain nn.Module graph here, which the stand-in does not
    #  have to emulate -- the fused executor under a GNT step is test_gnt_attack_step's)
    from nerfool_amd.ibrnet import feature_network
    saved, feature_network.CNN_PATH = feature_network.CNN_PATH, 'torch'
    try:
        pc.check_gnt_attack_step('cpu', train=True, universal_iters=1)
    finally:
        feature_network.CNN_PATH = saved
