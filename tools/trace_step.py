#!/usr/bin/env python3
"""Timeline of ONE train step from a rocprofv3 kernel trace (`rocprofv3 --kernel-trace --output-format csv -d DIR -o k --
python3 bench.py ...`): the kernels between the last two Adam launches, in start order, with the idle time in front of
each (no kernel of any queue running) and the level kernels folded into one line per sweep.

  python tools/trace_step.py gpurun_out/r02k/k_kernel_trace.csv [min_us=50]
"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    adam = [i for i, r in enumerate(rows) if 'k_adam' in r['Kernel_Name']]
    step = rows[adam[-2] + 1:adam[-1] + 1]
    t0 = int(step[0]['Start_Timestamp'])
    busy_end = t0
    idle = 0
    fold = collections.OrderedDict()
    agg = collections.defaultdict(lambda: [0, 0])
    print('%10s %9s %9s  q  kernel' % ('start ms', 'dur us', 'idle us'))
    for r in step:
        n = r['Kernel_Name'].replace('void ', '').replace('at::native::', '')
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = max(0, s - busy_end)
        idle += gap
        busy_end = max(busy_end, e)
        agg[n[:48]][0] += 1
        agg[n[:48]][1] += e - s
        small = (e - s) < min_us * 1e3 and gap < min_us * 1e3
        if 'k_level_fwd' in n or 'k_level_bwd' in n or small:
            key = 'level kernels' if 'k_level_' in n else 'kernels < %.0f us' % min_us
            f = fold.setdefault(key, [0, 0, 0])
            f[0] += 1; f[1] += e - s; f[2] += gap
            continue
        for k, f in fold.items():
            print('%10s %9.1f %9.1f     ... %d %s' % ('', f[1] / 1e3, f[2] / 1e3, f[0], k))
        fold.clear()
        print('%10.3f %9.1f %9.1f  %s  %s' % ((s - t0) / 1e6, (e - s) / 1e3, gap / 1e3, r['Queue_Id'], n[:90]))
    for k, f in fold.items():
        print('%10s %9.1f %9.1f     ... %d %s' % ('', f[1] / 1e3, f[2] / 1e3, f[0], k))
    span = int(step[-1]['End_Timestamp']) - t0
    print('step: %.3f ms from first to last kernel, %d kernels, %.3f ms with no kernel running, %.3f ms summed kernel time'
          % (span / 1e6, len(step), idle / 1e6, sum(v[1] for v in agg.values()) / 1e6))
    print('by kernel (>= 0.1 ms):')
    for n, v in sorted(agg.items(), key=lambda x: -x[1][1]):
        if v[1] >= 1e5:
            print('   %-50s %4d  %8.3f ms' % (n, v[0], v[1] / 1e6))


if __name__ == '__main__':
    main()
