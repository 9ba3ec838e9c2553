"""Per-step kernel table of the timed region of bench.py from a `rocprofv3 --kernel-trace --output-format csv` trace: the
window between the end of the last warm-up step and the end of the last timed step, delimited by the fused update kernel
(one launch per PGD step).  usage: python tools/steady_state_kernels.py <kernel_trace.csv> <timed steps> [rows]"""
import collections
import csv
import sys


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    upd = [r for r in rows if 'k_pgd_adam_step' in r['Kernel_Name'] or 'k_pgd_sign_step' in r['Kernel_Name']]
    t0, t1 = int(upd[-steps - 1]['End_Timestamp']), int(upd[-1]['End_Timestamp'])
    acc = collections.defaultdict(lambda: [0, 0])
    for r in rows:
        s = int(r['Start_Timestamp'])
        if t0 <= s <= t1:
            a = acc[r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]]
            a[0] += 1
            a[1] += int(r['End_Timestamp']) - s
    busy = sum(v[1] for v in acc.values())
    print('window %.3f ms/step (under the profiler), GPU busy %.3f ms/step, %d timed steps' %
          ((t1 - t0) / 1e6 / steps, busy / 1e6 / steps, steps))
    print('%-72s %9s %12s %10s %6s' % ('kernel', 'calls/step', 'ms/step', 'avg us', '%busy'))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%-72s %9.1f %12.4f %10.1f %6.1f' % (k, v[0] / steps, v[1] / 1e6 / steps, v[1] / v[0] / 1e3, 100.0 * v[1] / busy))


if __name__ == '__main__':
    main()
