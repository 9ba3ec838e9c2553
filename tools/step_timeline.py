"""Launch-by-launch timeline of ONE steady-state step of bench.py from a `rocprofv3 --kernel-trace --output-format csv` trace
(the last step: between the last two fused update kernels): start offset, duration, gap to the previous kernel, grid, name.
usage: python tools/step_timeline.py <kernel_trace.csv>"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    upd = [i for i, r in enumerate(rows) if 'k_pgd_adam_step' in r['Kernel_Name'] or 'k_pgd_sign_step' in r['Kernel_Name']]
    lo, hi = upd[-2] + 1, upd[-1]
    t0, prev_end = int(rows[lo]['Start_Timestamp']), None
    busy = gaps = 0
    for r in rows[lo:hi + 1]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = 0 if prev_end is None else s - prev_end
        busy += e - s
        gaps += max(gap, 0)
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]
        print('%9.1f us  %7.1f us  gap %5.1f  grid %8s x %4s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, r['Grid_Size_X'], r['Workgroup_Size_X'], name))
        prev_end = e
    print('step: %d launches, busy %.3f ms, gaps %.3f ms' % (hi + 1 - lo, busy / 1e6, gaps / 1e6))


if __name__ == '__main__':
    main()
