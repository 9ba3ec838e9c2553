"""profiles/rNN_pmc_traffic.json from two rocprofv3 PMC passes of bench.py (`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each in
its own run, csv output): HBM bytes per launch of every hand-written kernel in the timed steps, FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950, and the per-launch sum for the C-ABI entry points bench.py reports on.
usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <timed steps> <out.json>"""
import collections
import csv
import json
import sys

ABI = {'nf_ibrnet_fwd_mfma': ('k_ibr_rows_fwd', 'k_ibr_sol_fwd', 'k_ibr_ray_fwd'),
       'nf_ibrnet_bwd_mfma': ('k_ibr_rows_bwd', 'k_ibr_ray_bwd'),
       'nf_project_gather_fwd': ('k_project_gather_fwd',), 'nf_project_gather_bwd': ('k_project_gather_bwd',),
       'nf_pgd_adam_step': ('k_pgd_adam_step',), 'nf_conv3x3_wino': ('k_wino3x3<', 'k_wino3x3_bf<2, 3>', 'k_wino3x3_bf<1, 3>', 'k_wino3x3_bf<2, 1>', 'k_wino3x3_bf<1, 1>'),
       'nf_conv3x3_wino_bwd': ('k_wino3x3_bf<2, 2>', 'k_wino3x3_bf<1, 2>'),
       'nf_conv_s2_fwd': ('k_conv_s2_fwd',), 'nf_conv_s2_bwd': ('k_conv_s2_bwd',)}


def load(path, counter, steps):
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    upd = [i for i, r in enumerate(rows) if 'k_pgd_adam_step' in r['Kernel_Name']]
    rows = rows[upd[-steps - 1] + 1:upd[-1] + 1]
    acc = collections.defaultdict(list)
    for r in rows:
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if name.startswith('k_'):
            acc[name].append(float(r['Counter_Value']))
    return acc


def main():
    fetch, write = load(sys.argv[1], 'FETCH_SIZE', int(sys.argv[3])), load(sys.argv[2], 'WRITE_SIZE', int(sys.argv[3]))
    steps = int(sys.argv[3])
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, [0.0]), write.get(name, [0.0])
        kernels[name] = {'launches_per_step': len(f) / steps, 'FETCH_SIZE_KB_avg': round(sum(f) / len(f), 1),
                         'WRITE_SIZE_KB_avg': round(sum(w) / len(w), 1),
                         'hbm_bytes_per_launch': int(2 * 1024 * sum(f) / len(f) + 1024 * sum(w) / len(w))}
    R, V = 512, 4
    alg = {}
    # the scatter of d rgb_feat runs inside the backward row kernel when no stand-alone k_project_gather_bwd was launched:
    # no d rgb_feat write (35 floats per row), 4 taps x 32 channels added to the feature-map gradient instead
    fused_scatter = not any(name.startswith('k_project_gather_bwd') for name in kernels)
    for S in (64, 128):
        n = R * S
        alg[S] = {'nf_ibrnet_fwd_mfma': n * (V * 40 * 4 + (72 + 72 + 4) * 4),
                  'nf_ibrnet_bwd_mfma': n * (V * ((40 + (0 if fused_scatter else 35)) * 4 + (4 * 32 * 4 if fused_scatter else 0)) + (76 + 72 + 72) * 4),
                  'nf_project_gather_fwd': n * V * (4 * 35 * 4 + 44 * 4), 'nf_project_gather_bwd': n * V * (35 * 4 + 4 * 32 * 4)}
    abi = {}
    for entry, prefixes in ABI.items():
        tot, launches = 0.0, 0.0
        for name, k in kernels.items():
            if name.startswith(prefixes):
                tot += k['hbm_bytes_per_launch'] * k['launches_per_step']
        calls = 1 if entry == 'nf_pgd_adam_step' else 2          # coarse + fine level per step
        if entry in ('nf_conv3x3_wino', 'nf_conv3x3_wino_bwd', 'nf_conv_s2_fwd', 'nf_conv_s2_bwd'):     # one kernel launch per call
            calls = sum(k['launches_per_step'] for name, k in kernels.items() if name.startswith(prefixes)) or 1
        if tot == 0.0:          # not launched in this configuration (fused into another entry point)
            continue
        abi[entry] = {'hbm_bytes_per_launch': int(tot / calls), 'calls_per_step': calls}
        if entry in alg[64]:
            abi[entry]['algorithmic_bytes_per_launch'] = int((alg[64][entry] + alg[128][entry]) / 2)
        elif entry == 'nf_pgd_adam_step':
            abi[entry]['algorithmic_bytes_per_launch'] = 4 * 756 * 1008 * 3 * 32
    out = {'workload': {'model': 'ibrnet', 'n_rand': 512, 'height': 756, 'width': 1008, 'views': 4},
           'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 2 '
                      '--cpu-iters 0 --render-chunks 0 (one counter per run)',
           'correction': 'hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE (KB -> B); per launch = mean over the launches of the timed steps',
           'abi_kernels': abi, 'kernels': kernels}
    json.dump(out, open(sys.argv[4], 'w'), indent=1)
    for k, v in abi.items():
        print(k, v)


if __name__ == '__main__':
    main()
