"""Every launch of the kernels whose name contains PATTERN within the last step of a rocprofv3 kernel trace: start offset, duration, grid.
  python tools/trace_kernel_calls.py k_kernel_trace.csv PATTERN [PATTERN ...]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)))
rows.sort()
marks = [i for i, r in enumerate(rows) if 'k_adam' in r[2]]
step = rows[marks[-2] + 1:marks[-1] + 1]
t0 = step[0][0]
for s, e, n, wg in step:
    if any(p in n for p in sys.argv[2:]):
        print('%9.3f ms  %8.1f us  %7d workgroups  %s' % ((s - t0) / 1e6, (e - s) / 1e3, wg, n.split('(')[0][-50:]))
