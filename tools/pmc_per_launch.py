#!/usr/bin/env python3
"""Per-dispatch values of one PMC counter for the kernels whose name contains a pattern, in dispatch order:
  python tools/pmc_per_launch.py <p_counter_collection.csv> <COUNTER> <pattern> [<pattern> ...]"""
import collections
import csv
import sys


def main():
    path, counter, pats = sys.argv[1], sys.argv[2], sys.argv[3:]
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter or not any(p in r['Kernel_Name'] for p in pats):
            continue
        k = (int(r['Dispatch_Id']), r['Kernel_Name'][:60])
        acc[k] = acc.get(k, 0.0) + float(r['Counter_Value'])
    for (d, n), v in sorted(acc.items()):
        print('%6d  %-60s %14.0f' % (d, n, v))


if __name__ == '__main__':
    main()
