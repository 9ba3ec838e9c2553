"""A/B of the two forms of kernel A (nf_ibrnet_rows_form): the row form against the sample-on-the-lane form, forward (stand-alone and
gather-fused, i.e. the render path) and backward, at the attack and render sizes.  usage: python tools/bench_sol.py [iters] [library.so]"""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerfool_amd import ops                                   # noqa: E402
from nerfool_amd.ibrnet.mlp_network import IBRNet             # noqa: E402


def flops(R, S, V):
    return 2.0 * R * S * (V * 13256 + 6480 + 32 * S)


def timed(fn, iters):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    if len(sys.argv) > 2:
        sys.path.insert(0, os.path.join(ROOT, 'tests', 'host_harness'))
        import standin
        standin.use_library(sys.argv[2], emulated=False)
    dev = torch.device('cuda', 0)
    gen = torch.Generator().manual_seed(0)
    from nerfool_amd.synthetic import make_scene
    from nerfool_amd.ibrnet.sample_ray import RaySamplerSingleImage
    from nerfool_amd.ibrnet.render_ray import sample_along_camera_ray
    for (R, S, V) in ((512, 64, 4), (512, 128, 4), (4096, 64, 4), (4096, 128, 4), (4096, 64, 3), (4096, 64, 8), (4096, 64, 10), (512, 128, 8)):
        torch.manual_seed(3)
        net = IBRNet(SimpleNamespace(anti_alias_pooling=1, ibrnet_precision='fp32'), in_feat_ch=32, n_samples=S).to(dev)
        blob, mblob = net._packed(dev)
        rgb_feat = torch.randn(R, S, V, 35, generator=gen).to(dev)
        rd = torch.randn(R, S, V, 4, generator=gen)
        rd[..., :3] = torch.nn.functional.normalize(rd[..., :3], dim=-1)
        rd = rd.to(dev)
        mask = (torch.rand(R, S, V, generator=gen) > 0.1).float().to(dev)
        pe = net.pos_encoding
        d_raw = torch.randn(R, S, 4, generator=gen).to(dev)
        # gather-fused forward on a synthetic 800x800 scene
        data = make_scene(800, 800, V, seed=0)
        sampler = RaySamplerSingleImage(data, dev)
        rb = sampler.get_all()
        pts, _ = sample_along_camera_ray(rb['ray_o'][:R], rb['ray_d'][:R], rb['depth_range'], S, inv_uniform=True, det=True)
        fm = torch.randn(V, 200, 200, 32, generator=gen).to(dev).permute(0, 3, 1, 2)
        cam_ws = ops.camera_setup(rb['camera'], rb['src_cameras'])
        out = {}
        for form in ('rows', 'sol_fp32', 'auto'):
            ops.ibrnet_rows_form(form)
            raw, ws = ops.ibrnet_fwd_mfma(mblob, blob, pe, rgb_feat, rd, mask, True)
            tf = timed(lambda: ops.ibrnet_fwd_mfma(mblob, blob, pe, rgb_feat, rd, mask, True), iters)
            tb = timed(lambda: ops.ibrnet_bwd_mfma(mblob, blob, pe, rgb_feat, rd, mask, ws, d_raw, True), iters)
            with torch.no_grad():
                rg, _ = net.forward_gathered(pts, cam_ws, rb['src_rgbs'][0], fm)
                tg = timed(lambda: net.forward_gathered(pts, cam_ws, rb['src_rgbs'][0], fm), iters)
            out[form] = (tf, tb, tg, raw, rg)
        ops.ibrnet_rows_form('auto')
        F = flops(R, S, V)
        d1 = float((out['rows'][3] - out['auto'][3]).abs().max() / out['rows'][3].abs().max())
        d2 = float((out['rows'][4] - out['auto'][4]).abs().max() / out['rows'][4].abs().max())
        print('R %5d S %3d V %2d | fwd (rows + per-ray kernels): rows %.3f ms (%.1f TF), sol fp32 %.3f, sol bf16x3 %.3f ms (%.1f TF) x%.2f | gather-fused: '
              'rows %.3f, sol fp32 %.3f, sol bf16x3 %.3f ms x%.2f | bwd %.3f ms | max diff bf16x3 vs rows %.1e / %.1e'
              % (R, S, V, out['rows'][0], F / out['rows'][0] / 1e9, out['sol_fp32'][0], out['auto'][0], F / out['auto'][0] / 1e9,
                 out['rows'][0] / out['auto'][0], out['rows'][2], out['sol_fp32'][2], out['auto'][2], out['rows'][2] / out['auto'][2],
                 out['rows'][1], d1, d2), flush=True)

if __name__ == '__main__':
    main()
