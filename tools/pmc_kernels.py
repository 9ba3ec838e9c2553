"""HBM bytes per launch of every kernel in the timed steps of a bench.py run, from two rocprofv3 PMC passes (`--pmc FETCH_SIZE`, `--pmc
WRITE_SIZE`, each in its own run, csv output); FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950.  Any configuration
(tools/pmc_traffic.py adds the algorithmic bytes of the IBRNet config-2 entry points).
usage: python tools/pmc_kernels.py <fetch counter_collection.csv> <write counter_collection.csv> <timed steps>"""
import collections
import csv
import sys


def load(path, counter, steps):
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    upd = [i for i, r in enumerate(rows) if 'k_pgd_adam_step' in r['Kernel_Name']]
    rows = rows[upd[-steps - 1] + 1:upd[-1] + 1]
    acc = collections.defaultdict(list)
    for r in rows:
        acc[r['Kernel_Name'].split('(')[0].replace('void ', '')[:64]].append(float(r['Counter_Value']))
    return acc


def main():
    steps = int(sys.argv[3])
    fetch, write = load(sys.argv[1], 'FETCH_SIZE', steps), load(sys.argv[2], 'WRITE_SIZE', steps)
    print('%-66s %10s %14s %14s %16s' % ('kernel', 'calls/step', 'FETCH KB avg', 'WRITE KB avg', 'HBM MB / launch'))
    rows = []
    for name in set(fetch) | set(write):
        f, w = fetch.get(name, [0.0]), write.get(name, [0.0])
        mb = (2 * 1024 * sum(f) / len(f) + 1024 * sum(w) / len(w)) / 1e6
        rows.append((mb * len(f), name, len(f) / steps, sum(f) / len(f), sum(w) / len(w), mb))
    for _, name, calls, fk, wk, mb in sorted(rows, reverse=True):
        print('%-66s %10.1f %14.1f %14.1f %16.2f' % (name, calls, fk, wk, mb))


if __name__ == '__main__':
    main()
