"""profiles/rNN_rocprofv3_summary.md from the round's artefacts (bench JSON lines, steady-state table, PMC traffic).
usage: python tools/make_profile_summary.py [round, default r02]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')


def load(name):
    with open(os.path.join(P, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def table(kernels):
    out = ['| C-ABI entry point | launches | mean ms | achieved | of peak |', '|---|---|---|---|---|']
    for k, v in kernels.items():
        out.append('| %s | %d | %.4f | %.1f %s | %.3f |' % (k, v['launches'], v['mean_ms'], v['achieved'], v['unit'], v['frac']))
    return '\n'.join(out)


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else 'r02'
    b, g, c5 = load(rnd + '_bench_ibrnet.json'), load(rnd + '_bench_gnt.json'), load(rnd + '_bench_c5_bf16.json')
    with open(os.path.join(P, rnd + '_pmc_traffic.json')) as f:
        pmc = json.load(f)
    steady = open(os.path.join(P, rnd + '_steady_state_kernels.txt')).read().rstrip()
    r, c, ex = b['roofline'], b['cpu_baseline'], b['extra']
    md = []
    md.append('# Round %s profiles (MI355X, gfx950) -- final state of the round\n' % rnd[1:].lstrip('0'))
    md.append('Commands (all through `tools/profile_round.sh` on one box; `cd /tmp; export TMPDIR=/tmp` first):\n')
    md.append('```\n'
              'python3 bench.py --steps 20 --warmup 3                                   -> %(r)s_bench_ibrnet.json\n'
              'rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --extras 0\n'
              '                                                                         -> %(r)s_rocprofv3_kernel_stats_bench_steps10.csv\n'
              'rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 bench.py --steps 3 --warmup 2 --extras 0\n'
              'rocprofv3 --pmc WRITE_SIZE --kernel-trace ... -- (same)                  -> %(r)s_pmc_traffic.json (tools/pmc_traffic.py)\n'
              'python3 bench.py --config c4 --steps 5 --warmup 2 --render-chunks 2      -> %(r)s_bench_gnt.json\n'
              'python3 bench.py --config c5 --steps 10 --warmup 3                       -> %(r)s_bench_c5_bf16.json\n'
              '```\n' % {'r': rnd})
    md.append('The `--stats` CSV covers the whole process, including the one-off per-shape timing of the two Winograd workgroup widths in the\n'
              'first warm-up step (outside the timed region).  The table below is the TIMED region only: the kernel trace cut between the\n'
              'fused update kernels of the last warm-up step and of the last timed step (`tools/steady_state_kernels.py` ->\n'
              '`%s_steady_state_kernels.txt`).  The bench itself brackets only the kernels of its roofline table with HIP events inside\n'
              'the timed region.\n' % rnd)
    md.append('## Headline bench line\n')
    k = ex['kernels'][r['kernel']]
    md.append('`%.0f rays/s`, `%.2f ms/step` (N_rand 512, 756x1008, V 4, 64+64 samples); roofline of the dominant hand-written entry\n'
              'point `%s` (%d launches per step): %.1f %s = %.3f of the fp32 matrix peak (Winograd-domain products; %.0f TFLOP/s in\n'
              'direct-form terms), HBM traffic %.1f MB per launch (PMC) vs %.1f MB algorithmic.  Whole step: %.1f TFLOP/s of direct-form\n'
              'FLOPs = %.2f of the fp32 matrix peak.  cpu_baseline %.1f rays/s (%.2f s per PGD iteration, %d timed) on %d physical cores\n'
              '(%s); CPU render leg %.0f rays/s.\n'
              % (b['value'], b['ms_per_step'], r['kernel'], k['launches'] // b['steps'], r['achieved'], r['unit'], r['frac'],
                 k.get('direct_form_equivalent_tflops', 0.0), (r['traffic'] or 0) / 1e6, (r['traffic_algorithmic'] or 0) / 1e6,
                 ex['whole_step']['achieved_tflops'], ex['whole_step']['frac_of_fp32_mfma_peak'], c['value'], c['attack_s_per_iter'], 10,
                 c['cores'], c['cpu_model'], c['render_rays_per_s']))
    md.append(table(ex['kernels']) + '\n')
    md.append('Other legs of the same run: N_rand 4096: %.2f ms/step (%.0f rays/s); render 4096-ray chunks 64+64: %.2f M rays/s; whole\n'
              '756x1008 image through render_single_image (187 chunks, D2H of all outputs): %.3f s (%.2f M rays/s); 800x800, 64 samples\n'
              'coarse only: %.2f M rays/s (%.2f of the fp32 matrix peak), 64+64: %.2f M rays/s.  3x3 choice per layer: %s.\n'
              % (ex['attack_n_rand_4096']['ms_per_step'], ex['attack_n_rand_4096']['rays_per_s'], ex['render']['rays_per_s'] / 1e6,
                 ex['render_single_image']['seconds'], ex['render_single_image']['rays_per_s'] / 1e6,
                 ex['render_800x800_64']['rays_per_s'] / 1e6, ex['render_800x800_64']['mfma_frac_of_peak'],
                 ex['render_800x800_64+64']['rays_per_s'] / 1e6, json.dumps(ex['conv3x3_choice'])))
    md.append('## Steady-state kernel table (per PGD step, under the profiler)\n')
    md.append('```\n' + steady + '\n```\n')
    md.append('Reading: `k_wino3x3` (the 27 stride-1 3x3 convolutions, forward + backward-data) is 47 % of the GPU-busy time; the stride-2\n'
              'convolutions (`k_conv_s2_*`, 8 launches) 1.07 ms where MIOpen / rocBLAS took 1.85 ms; no `miopen*`, `Cijk_*`, `igemm_*`,\n'
              '`Col2Im*` or `batched_transpose*` row is left; the remaining `at::native` rows are the src + delta add, the zero fill of the\n'
              'scatter target and the ray-index gather (0.07 ms).  Fused CNN glue (`k_in_*`, `k_plane_*`, `k_upsample2x_pad*`,\n'
              '`k_pad_gather_*`) + `k_conv1x1` 2.8 ms; IBRNet network kernels 0.86 ms; gather fwd+bwd 0.27 ms.\n')
    md.append('## HBM traffic (PMC, per launch, timed steps)\n')
    md.append('| C-ABI entry point | HBM bytes (2 x FETCH_SIZE + WRITE_SIZE) | algorithmic bytes |\n|---|---|---|')
    for kk, v in pmc['abi_kernels'].items():
        alg = v.get('algorithmic_bytes_per_launch')
        if alg is None and kk == 'nf_conv3x3_wino':
            alg = r.get('traffic_algorithmic')
        md.append('| %s | %.1f MB | %s |' % (kk, v['hbm_bytes_per_launch'] / 1e6, '—' if not alg else '%.1f MB' % (alg / 1e6)))
    md.append('\nThe x2 FETCH_SIZE correction is confirmed by the update kernel: 5 read + 3 written streams of 36.58 MB = 292.6 MB.\n'
              '`nf_project_gather_fwd` fetches half its algorithmic bytes: the 4 bilinear taps of neighbouring samples share lines in L2.\n'
              '`nf_conv3x3_wino` (mean over the 54 launches of a step): the x2 correction over-counts its 8-byte window reads, the remaining\n'
              'excess is the 18x10 window per 16x8 output block (1.4x on the input side) and the re-read of a window by the 2-4\n'
              'output-channel groups of the wide layers when they miss in L2.  `nf_conv_s2_fwd/bwd` (mean over their 4 launches each):\n'
              'their input + output tensors.\n')
    md.append('## GNT (config 4: depth 8, 800x800, V 10, S 64, N_rand 512)\n')
    gk = g['extra']['kernels']
    md.append('`%.1f ms/step` (`%.0f rays/s`).  %s.  Render leg: %.0f rays/s.\n'
              % (g['ms_per_step'], g['value'], ', '.join('%s %.2f ms x %d' % (kk, v['mean_ms'], v['launches'] // g['steps']) for kk, v in gk.items()),
                 g['extra']['render']['rays_per_s']))
    md.append('## Config 5 (IBRNet, 512x512, V 8, 128+128 samples, bf16 row network)\n')
    ck = c5['extra']['kernels']
    md.append('`%.2f ms/step` (`%.0f rays/s`), dtype %s.  %s.  `%s_bench_c5_fp32.json`: the same workload with fp32 rows.\n'
              % (c5['ms_per_step'], c5['value'], c5['dtype'],
                 ', '.join('%s %.3f ms (%.3f of its peak)' % (kk, v['mean_ms'], v['frac']) for kk, v in ck.items() if 'ibrnet' in kk), rnd))
    try:
        lr = load(rnd + '_bench_1000iters_ibrnet.json')
        md.append('## 1000-iteration attack, end to end\n')
        md.append('`python3 bench.py --steps 1000 --warmup 3 --extras 0`: %.2f s (`%s_bench_1000iters_ibrnet.json`, %.0f rays/s); the CPU oracle\n'
                  'extrapolates to %.0f s on the %d host cores.\n' % (lr['ms_per_step'], rnd, lr['value'], 1000 * c['attack_s_per_iter'], c['cores']))
    except OSError:
        pass
    md.append('## Other artefacts\n')
    md.append('`%(r)s_grad_budget_before.json` (per-stage gradient error budget against float64 before the ReLU-pattern analysis,\n'
              '`tools/diag_grad_budget.py`), `%(r)s_bench_2rank_gloo_one_gpu_functional.json` (all four multi-GPU forms through gloo on one\n'
              'GPU: functional check, not a measurement).\n' % {'r': rnd})
    open(os.path.join(P, rnd + '_rocprofv3_summary.md'), 'w').write('\n'.join(md))
    print('written', os.path.join(P, rnd + '_rocprofv3_summary.md'))


if __name__ == '__main__':
    main()
