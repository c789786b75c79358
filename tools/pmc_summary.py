#!/usr/bin/env python
"""Sum / average rocprofv3 --pmc counter CSVs per kernel:  python tools/pmc_summary.py <counter_collection.csv> [...]"""
import csv, sys
from collections import defaultdict
for path in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r['Kernel_Name'].split('(')[0]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            calls[k].add(r['Dispatch_Id'])
    print('==', path)
    for k in sorted(acc, key=lambda k: -len(calls[k])):
        n = len(calls[k])
        print('  {} ({} dispatches)'.format(k[:100], n))
        for c, v in sorted(acc[k].items()):
            print('      {:32s} per dispatch {:16.1f}'.format(c, v / n))
