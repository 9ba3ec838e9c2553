"""GPU, single-GPU box: what a SHARDED PGD step costs the host when RCCL is the backend.

A one-rank RCCL group (`init_process_group('nccl', world_size=1)`: all a single-GPU box allows) with
`RayShard(exchange_when_alone=True)`, so that both multi-GPU forms issue every collective of their step -- 2 (replicated feature CNN:
16-byte counts / loss all-reduce + all-reduce of d delta) or 4 (CNN sharded by view: + all-gather of the feature maps + reduce-scatter
of their gradients) -- through torch.distributed on the RCCL backend, at BASELINE config 2's size.  Measured per form, eager
(launch by launch) and replayed as hipGraph segments split at the collectives (eval_adv._SegmentedCapture):

  host_issue_ms_per_step   host time to enqueue a step, no synchronisation inside the bracket
  ms_per_step              steps between synchronisations

The data a one-rank collective moves never leaves the GPU, so `ms_per_step` says nothing about xGMI; the HOST side -- graph launches,
the torch.distributed call path, RCCL's enqueue -- is what an 8-GPU rank pays too.

    python tools/shard_host_issue.py [--json out.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--json')
    ap.add_argument('--steps', type=int, default=20)
    o = ap.parse_args()
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29641')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    sys.argv = [sys.argv[0]]
    a = bench.parse()
    args, data, model, sampler, src, projector, EA = bench.build_problem(a, dev)
    out = {'workload': 'BASELINE config 2 step (756x1008, V 4, 64+64 samples, N_rand 512) on a ONE-rank RCCL group, every collective issued',
           'forms': {}}

    def sync():
        torch.cuda.synchronize()

    for form in ('unsharded', 'replicated', 'view'):
        for graph in (False, None):
            shard = None if form == 'unsharded' else EA.RayShard(shard_views=form == 'view', exchange_when_alone=True)
            atk = EA.PGDAttack(bench.make_args(a, a.n_rand), model, projector, src, shard=shard, graph=graph)
            for _ in range(5):
                atk.step(data)
            sync()
            t0 = time.perf_counter()
            for _ in range(o.steps):
                atk.step(data)
            sync()
            ms = 1e3 * (time.perf_counter() - t0) / o.steps
            sync()
            t0 = time.perf_counter()
            for _ in range(3):
                atk.step(data)
            issue = 1e3 * (time.perf_counter() - t0) / 3
            sync()
            segs = [len(v[0].graphs) for v in atk._graphs.values()]
            out['forms']['%s/%s' % (form, 'eager' if graph is False else 'graph')] = {
                'ms_per_step': round(ms, 4), 'host_issue_ms_per_step': round(issue, 4), 'graph_segments_per_step': segs[0] if segs else 0,
                'collectives_per_step': 0 if shard is None else shard.collectives / float(atk.iters)}
            print(form, 'eager' if graph is False else 'graph', out['forms']['%s/%s' % (form, 'eager' if graph is False else 'graph')], flush=True)
            del atk
    dist.destroy_process_group()
    if o.json:
        with open(o.json, 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
