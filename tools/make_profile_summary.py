"""profiles/rNN_rocprofv3_summary.md from the round's artefacts (bench JSON lines, steady-state tables, PMC traffic).
usage: python tools/make_profile_summary.py [round, default r05]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')


def load(name):
    with open(os.path.join(P, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def text(name, n=None):
    with open(os.path.join(P, name)) as f:
        lines = f.read().rstrip().splitlines()
    return '\n'.join(lines if n is None else lines[:n])


def table(kernels):
    out = ['| C-ABI entry point | launches timed | mean ms | achieved | peak it is priced against | of peak |', '|---|---|---|---|---|---|']
    for k, v in kernels.items():
        pipe = 'bf16 matrix pipe, executed products' if v.get('pipe') else ('fp32 matrix pipe' if v['unit'] == 'TFLOP/s' and v['peak'] < 1000 else
                                                                              ('bf16 matrix pipe' if v['unit'] == 'TFLOP/s' else 'HBM'))
        out.append('| %s | %d | %.4f | %.1f %s | %.0f %s (%s) | %.3f |' % (k, v['launches'], v['mean_ms'], v['achieved'], v['unit'], v['peak'], v['unit'], pipe, v['frac']))
    return '\n'.join(out)


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else 'r06'
    b, g = load(rnd + '_bench_ibrnet.json'), load(rnd + '_bench_gnt.json')
    c5, c5f = load(rnd + '_bench_c5_bf16.json'), load(rnd + '_bench_c5_fp32.json')
    k1000 = load(rnd + '_bench_1000iters_ibrnet.json')
    with open(os.path.join(P, rnd + '_pmc_traffic.json')) as f:
        pmc = json.load(f)
    r, c, ex = b['roofline'], b['cpu_baseline'], b['extra']
    md = ['# Round %s profiles (MI355X, gfx950) -- final state of the round\n' % rnd[1:].lstrip('0')]
    md.append('All through `tools/profile_round.sh` on one box (`cd /tmp; export TMPDIR=/tmp` first); per configuration (c2 = BASELINE config 2, the\n'
              'headline; c4 = GNT; c5 = IBRNet V 8, 128 + 128 samples, bf16 rows):\n')
    md.append('```\n'
              'python3 bench.py --steps 20 --warmup 3                                           -> %(r)s_bench_ibrnet.json\n'
              'python3 bench.py --steps 1000 --warmup 3 --extras 0 --cpu-iters 0                -> %(r)s_bench_1000iters_ibrnet.json\n'
              'rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py [--config cN] --steps 10 --warmup 3 --extras 0 --event-every 0\n'
              '      -> %(r)s_rocprofv3_kernel_stats_cN.csv, %(r)s_steady_state_kernels_cN.txt (tools/steady_state_kernels.py), %(r)s_step_timeline_cN.txt\n'
              'rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 bench.py [--config cN] --steps 3 --warmup 2 --extras 0 --event-every 0\n'
              'rocprofv3 --pmc WRITE_SIZE --kernel-trace ... -- (same)   -> %(r)s_pmc_traffic_cN.txt (tools/pmc_kernels.py), %(r)s_pmc_traffic.json (c2)\n'
              'python3 bench.py --config c4 --steps 5 --warmup 2 --render-chunks 4              -> %(r)s_bench_gnt.json\n'
              'python3 bench.py --config c5 [--precision fp32] --steps 10 --warmup 3            -> %(r)s_bench_c5_bf16.json, %(r)s_bench_c5_fp32.json\n'
              'bash tools/pmc_render.sh                                                         -> %(r)s_pmc_render_traffic.txt\n'
              'python3 -m pytest tests -m gpu -q -s | grep ...                                  -> %(r)s_parity_numbers.txt\n'
              '```\n' % {'r': rnd})
    md.append('The `--stats` CSVs cover the whole process (warm-up steps included; the Winograd workgroup width is a rule on the shape since round 4, no\n'
              'timing runs); the steady-state tables are the TIMED region only: the kernel trace cut between the fused update kernels of\n'
              'the last warm-up step and of the last timed step.  In the traced runs the bench\'s HIP-event brackets are off (`--event-every 0`);\n'
              'in the bench lines they are live on every 10th timed step (`roofline_sampling`).  FETCH_SIZE is doubled as MI355X_MICROARCH.md\n'
              'prescribes for gfx950 (confirmed on the delta update: %.2f MB counted vs %.2f MB = 8 streams).\n'
              % (pmc['abi_kernels']['nf_pgd_adam_step']['hbm_bytes_per_launch'] / 1e6,
                 pmc['abi_kernels']['nf_pgd_adam_step']['algorithmic_bytes_per_launch'] / 1e6))
    md.append('## Config 2 (headline)\n')
    k = ex['kernels'][r['kernel']]
    md.append('`%.0f rays/s`, `%.2f ms/step` (N_rand 512, 756x1008, V 4, 64+64 samples; 1000 steps: %.3f ms/step); dominant hand-written entry\n'
              'point `%s`: %.1f %s executed on the bf16 matrix pipe = %.3f of its 2500 TFLOP/s dense peak (six bf16 products per Winograd-domain product of the forward; %.0f TFLOP/s in direct-form terms), HBM traffic\n'
              '%.1f MB per launch (PMC) vs %.1f MB algorithmic.  Whole step: %.1f TFLOP/s of direct-form FLOPs = %.3f of the fp32 matrix peak.\n'
              'cpu_baseline %.1f rays/s (%.2f s per PGD iteration) on %d physical cores (%s); CPU render leg %.0f rays/s.\n'
              % (b['value'], b['ms_per_step'], k1000['ms_per_step'], r['kernel'], r['achieved'], r['unit'], r['frac'],
                 k.get('direct_form_equivalent_tflops', 0.0), (r['traffic'] or 0) / 1e6, (r['traffic_algorithmic'] or 0) / 1e6,
                 ex['whole_step']['achieved_tflops'], ex['whole_step']['frac_of_fp32_mfma_peak'], c['value'], c['attack_s_per_iter'],
                 c['cores'], c['cpu_model'], c['render_rays_per_s']))
    md.append(table(ex['kernels']) + '\n')
    rs, r8, r88 = ex['render_single_image'], ex['render_800x800_64'], ex['render_800x800_64+64']
    md.append('Other legs of the same run: N_rand 4096: %.2f ms/step (%.0f rays/s); render 4096-ray chunks 64+64: %.2f M rays/s; whole\n'
              '756x1008 image through render_single_image (187 chunks, D2H of all outputs): %.3f s (%.2f M rays/s); 800x800, 64 samples\n'
              'coarse only: %.2f M rays/s (%.3f of the fp32 matrix peak), 64+64: %.2f M rays/s (%.3f).\n'
              % (ex['attack_n_rand_4096']['ms_per_step'], ex['attack_n_rand_4096']['rays_per_s'], ex['render']['rays_per_s'] / 1e6,
                 rs['seconds'], rs['rays_per_s'] / 1e6, r8['rays_per_s'] / 1e6, r8['mfma_frac_of_peak'], r88['rays_per_s'] / 1e6,
                 r88['mfma_frac_of_peak']))
    md.append('Steady-state kernel table (rocprofv3 kernel trace, timed steps only):\n\n```\n' + text(rnd + '_steady_state_kernels_c2.txt', 40) + '\n```\n')
    md.append('Issue counters of the same kernels (`tools/pmc_sq.sh`, one rocprofv3 --pmc pass per counter group; "mfma busy" =\n'
              'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the share of the launch in which a SIMD\'s matrix pipe executes,\n'
              'counting recomputed and padded products; "wait(cnt)" = SQ_WAIT_ANY / SQ_WAVE_CYCLES, the share of a wave\'s life spent in s_waitcnt):\n\n```\n'
              + text(rnd + '_sq_pipe_c2.txt', 26) + '\n```\n')
    md.append('PMC HBM traffic per launch of the C-ABI entry points (`%s_pmc_traffic.json`):\n' % rnd)
    md.append('| entry point | calls / step | HBM MB / launch (PMC) | algorithmic MB |\n|---|---|---|---|')
    for name, v in pmc['abi_kernels'].items():
        alg = v.get('algorithmic_bytes_per_launch')
        md.append('| %s | %g | %.1f | %s |' % (name, v['calls_per_step'], v['hbm_bytes_per_launch'] / 1e6, '-' if alg is None else '%.1f' % (alg / 1e6)))
    md.append('\nRender path (`%s_pmc_render_traffic.txt`):\n\n```\n%s\n```\n' % (rnd, text(rnd + '_pmc_render_traffic.txt')))
    for tag, bj, title in (('c4', g, 'Config 4 (GNT depth 8, 800x800, V 10, 64 samples)'), ('c5', c5, 'Config 5 (IBRNet, 512x512, V 8, 128 + 128 samples, bf16 rows)')):
        md.append('## %s\n' % title)
        md.append('`%.0f rays/s`, `%.2f ms/step`%s.\n' % (bj['value'], bj['ms_per_step'],
                                                    '' if tag == 'c4' else ' (fp32 rows: %.2f ms/step)' % c5f['ms_per_step']))
        md.append(table(bj['extra']['kernels']) + '\n')
        md.append('Steady-state kernel table:\n\n```\n' + text('%s_steady_state_kernels_%s.txt' % (rnd, tag), 24) + '\n```\n')
        md.append('Issue counters:\n\n```\n' + text('%s_sq_pipe_%s.txt' % (rnd, tag), 14) + '\n```\n')
        md.append('PMC HBM traffic per launch (2 x FETCH_SIZE + WRITE_SIZE, separate passes):\n\n```\n' + text('%s_pmc_traffic_%s.txt' % (rnd, tag), 14) + '\n```\n')
    def opt(title, name, note=''):
        if os.path.exists(os.path.join(P, rnd + name)):
            md.append('## %s (`%s`)\n\n%s```\n%s\n```\n' % (title, rnd + name, note, text(rnd + name)))
    opt('bf16-split operand mix against the fp32 mix (`tools/experimental/probe_bf3.hip`)', '_probe_bf16_split_mix.txt',
        'fp32 matrix instructions run at the vector rate and their time adds to the vector instructions around them; the bf16 ones run at\n'
        '16x the rate and beside the vector pipe: the 3-way split mix of a Winograd chunk takes 0.43 of the shipped fp32 mix.\n\n')
    opt('Winograd kernel, operand forms and ablations (`tools/bench_wino_bf.py`)', '_wino_bf16x3_ablation.txt',
        'fp32 / bf16x3 / bf16 per layer; then timing-only builds of the bf16x3 kernel (wrong results): one record piece streamed per step instead of\n'
        'three, no operand split arithmetic, both.\n\n')
    opt('IBRNet forward, row form vs sample-on-the-lane (`tools/bench_sol.py`)', '_row_kernel_forms.txt')
    opt('CNN glue micro-benchmark (`tools/bench_cnn_glue.py`)', '_cnn_glue_microbench.txt')
    opt('Plain bf16 operands in the 3x3 convolutions: measured and rejected (`tools/diag_bf16_cnn.py`)', '_plain_bf16_cnn_rejected.txt')
    opt('MFMA / VALU overlap probe (`tools/experimental/probe_overlap.hip`)', '_probe_mfma_valu_overlap.txt')
    opt('Winograd kernel: what a launch is made of (timing builds, phase timers)', '_wino_ablation.txt')
    opt('A sharded PGD step on the RCCL backend, one-rank group, every collective issued: eager vs hipGraph segments (`tools/shard_host_issue.py`)',
        '_sharded_step_one_rank_rccl.json')
    opt('Whole attacks against the reference\'s own float32 / float64 / other-order runs (`NERFOOL_PARITY_LOG` of the GPU tests)',
        '_attack100_outcome.txt')
    if 'universal' in ex:
        u = ex['universal']
        md.append('## Universal loop on one GPU (BASELINE config 3\'s loop, `extra.universal`)\n\n%d target views: %.2f ms per step once every view replays '
                  '(%d captured graphs, first three cycles %.2f s, allocator peak %.2f GB); all-bf16x3 step (`extra.step_ms_all_bf16x3`): %.2f ms; '
                  'host issue %.2f ms per step.\n' % (u['target_views'], u['ms_per_step'], u['graphs_captured'], u['first_three_cycles_s'],
                                                   u['hbm_peak_allocated_gb'], ex.get('step_ms_all_bf16x3', float('nan')), b['host_issue_ms_per_step']))
    if 'training_mode_step' in g['extra']:
        t = g['extra']['training_mode_step']
        md.append('## GNT in training mode (Dropout live, config 4\'s universal loop; `extra.training_mode_step` of the GNT bench line)\n\n'
                  '%.2f ms per step = %.3f x the eval-mode step; kernels %s.\n' % (t['ms_per_step'], t['vs_eval_mode_step'], t['kernels_ms']))
    md.append('## Parity figures printed by the GPU tests (`%s_parity_numbers.txt`)\n\n```\n%s\n```\n' % (rnd, text(rnd + '_parity_numbers.txt')))
    with open(os.path.join(P, rnd + '_rocprofv3_summary.md'), 'w') as f:
        f.write('\n'.join(md))
    print('wrote', rnd + '_rocprofv3_summary.md')


if __name__ == '__main__':
    main()
