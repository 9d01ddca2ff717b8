#!/usr/bin/env python3
"""Where do fresh-batch steps lose time against resident-batch steps?  Reads the rocprofv3 kernel trace of `tools/fresh_probe.py`
(resident loop first, then the prefetched loops) and compares, per step (between consecutive optimizer launches), the summed duration of
the step's own kernels by name and the time no step kernel was running, for the resident loop and for the first prefetched loop.
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ft -o k -- python3 tools/fresh_probe.py 12 4
  python3 tools/fresh_trace.py gpurun_out/ft/k_kernel_trace.csv"""
import collections
import csv
import sys


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '?'))))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if 'adam' in r[2].lower()]
    main_stream = collections.Counter(rows[i][3] for i in adam).most_common(1)[0][0]
    steps = []
    for a, b in zip(adam[:-1], adam[1:]):
        t0, t1 = rows[a][1], rows[b][1]
        ks = rows[a + 1:b + 1]
        own = [k for k in ks if k[3] == main_stream]
        other = [k for k in ks if k[3] != main_stream]
        steps.append((t1 - t0, own, other))
    # the resident loop: steps with (almost) no kernels on other streams that are not the trainer's side stream; classify by the
    # presence of plan-builder kernels
    def is_plan(k):
        return any(s in k[2] for s in ('k_csr_', 'k_colour', 'k_run_', 'k_rep_', 'k_seg_level', 'radix_sort', 'k_order_rows', 'k_bucket_keys'))
    # fresh_probe.py's order: 3 + n resident steps, then 3 + n prefetched ones (plan on the worker), then 3 + n with the plan in the step
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    res, fre = steps[3:3 + n - 1], steps[3 + n + 3:3 + n + 3 + n - 1]
    print('%d resident steps, %d prefetched steps' % (len(res), len(fre)))

    def summary(group, label):
        n = len(group)
        wall = sum(s[0] for s in group) / n / 1e6
        by = collections.defaultdict(float)
        for _, own, other in group:
            for k in own + other:
                if not is_plan(k):
                    by[k[2].split('(')[0][-60:]] += (k[1] - k[0]) / n / 1e6
        plan_ms = sum((k[1] - k[0]) for _, _, other in group for k in other if is_plan(k)) / n / 1e6
        n_plan = sum(1 for _, _, other in group for k in other if is_plan(k)) / n
        print('%s: %.2f ms per step wall; plan-builder kernels beside it: %.2f ms in %.0f launches' % (label, wall, plan_ms, n_plan))
        return wall, by
    def gaps(group, label):
        # the step stream's own timeline: time covered by its kernels, and the idle time between them split by whether a kernel of
        # ANOTHER stream was running in the gap (the device was busy with the batch builder) or nothing ran at all (the host was late)
        n = len(group)
        cover = idle_busy = idle_empty = 0.0
        for _, own, other in group:
            own = sorted(own)
            oth = sorted(other)
            end = own[0][0]
            for k in own:
                if k[0] > end:
                    g0, g1 = end, k[0]
                    busy = 0
                    for o in oth:
                        lo, hi = max(o[0], g0), min(o[1], g1)
                        if hi > lo:
                            busy += hi - lo
                    busy = min(busy, g1 - g0)
                    idle_busy += busy
                    idle_empty += (g1 - g0) - busy
                cover += max(0, k[1] - max(end, k[0]))
                end = max(end, k[1])
        print('%s: step stream covered %.2f ms per step, gaps with another stream\'s kernels running %.2f ms, gaps with nothing running %.2f ms'
              % (label, cover / n / 1e6, idle_busy / n / 1e6, idle_empty / n / 1e6))
    gaps(res, 'resident')
    gaps(fre, 'fresh   ')
    w0, b0 = summary(res, 'resident')
    w1, b1 = summary(fre, 'fresh   ')
    print('per kernel name, summed duration per step (ms): resident -> fresh (difference), largest differences first')
    diffs = sorted(((b1.get(k, 0.0) - b0.get(k, 0.0), k) for k in set(b0) | set(b1)), reverse=True)
    for d, k in diffs[:14]:
        print('  %+7.3f  %8.3f -> %8.3f  %s' % (d, b0.get(k, 0.0), b1.get(k, 0.0), k))
    print('  sum of all differences: %+.3f ms; wall difference %+.3f ms' % (sum(d for d, _ in diffs), w1 - w0))


if __name__ == '__main__':
    main()
