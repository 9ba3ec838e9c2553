"""Forward-only rendering of 4096-ray chunks (800x800, V = 4, 64 samples coarse only) -- the program profiled with rocprofv3 --pmc
for the HBM traffic of the render path with and without the gather fused into the row kernel.
usage: python tools/render_chunks.py [chunks] [fused|separate] [library.so]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                              # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    from nerfool_amd.ibrnet import mlp_network
    mlp_network.GATHER_BWD_FUSION = sys.argv[2] if len(sys.argv) > 2 else 'fused'
    if len(sys.argv) > 3:          # a tuning build (tools/build_variant.sh)
        from nerfool_amd import _lib
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'host_harness'))
        import standin          # test hook: bind a tuning build of the kernel sources
        standin.use_library(sys.argv[3], emulated=False)
    a = argparse.Namespace(gpus=1, steps=1, warmup=0, n_rand=512, height=800, width=800, views=4, samples=64, importance=0, render_chunks=n,
                           model='ibrnet', config='c2', precision='fp32', depth=8, cnn_shard='replicated', scaling='weak', cpu_iters=0, extras=0)
    dev = torch.device('cuda', 0)
    args, data, model, sampler, src, projector, _ = bench.build_problem(a, dev)
    from nerfool_amd.ibrnet.render_ray import render_rays
    with torch.no_grad():
        fm = model.feature_net(src['src_rgbs'].squeeze(0).permute(0, 3, 1, 2))
        rays = sampler.get_all()
        for i in range(n + 1):
            rb = {k: (v[i * 4096:(i + 1) * 4096] if k in ('ray_o', 'ray_d', 'rgb') else v) for k, v in rays.items()}
            render_rays(rb, model, fm, projector, 64, inv_uniform=True, N_importance=0, det=True, src_ray_batch=src)
    torch.cuda.synchronize()
    print('rendered', n + 1, 'chunks')


if __name__ == '__main__':
    main()
