"""Which launches of a steady-state step are NOT hand-written kernels (fills, copies, torch elementwise / sort kernels)?
Reads a rocprofv3 kernel trace (k_kernel_trace.csv) of `bench.py --steps K`, takes the launches between the last two
`k_adam` launches (one step) and prints them grouped by name: calls, total us, share of the step's kernel time."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Queue_Id', '?')))
rows.sort()
marks = [i for i, r in enumerate(rows) if 'k_adam' in r[2] or 'adam' in r[2].lower()]
if len(marks) < 2:
    sys.exit('no optimizer launches found')
lo, hi = marks[-2] + 1, marks[-1] + 1
step = rows[lo:hi]
wall = (step[-1][1] - step[0][0]) / 1e3
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n, _g, _q in step:
    short = n.split('(')[0][-90:]
    own = 'mgv::' in n
    t = tot[('own ' if own else 'GLUE ') + short]
    t[0] += 1; t[1] += e - s
ksum = sum(v[1] for v in tot.values()) / 1e3
print('step: %d launches, %.1f us from first start to last end, %.1f us summed kernel time' % (len(step), wall, ksum))
glue = sum(v[1] for k, v in tot.items() if k.startswith('GLUE')) / 1e3
print('glue (not hand-written): %.1f us = %.2f %% of the summed kernel time, %d launches' % (glue, 100 * glue / ksum, sum(v[0] for k, v in tot.items() if k.startswith('GLUE'))))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    if k.startswith('GLUE') or v[1] > 0.005 * ksum * 1e3:
        print('%9.1f us %5d  %s' % (v[1] / 1e3, v[0], k))
print('\nglue launches over 100 us, with the launches around them (by start time):')
for i, (s, e, n, grid, q) in enumerate(step):
    if 'mgv::' not in n and e - s > 100000:
        ctx = 'grid %s queue %s: ' % (grid, q) + ' | '.join('[q%s] ' % x[4] + x[2].split('(')[0][-40:] for x in step[max(0, i - 4):i]) + '  >>  ' + n.split('(')[0][-60:] + ' %.0f us  >>  ' % ((e - s) / 1e3) + ' | '.join('[q%s] ' % x[4] + x[2].split('(')[0][-40:] for x in step[i + 1:i + 4])
        print(ctx)
