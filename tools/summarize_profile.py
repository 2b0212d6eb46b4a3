#!/usr/bin/env python3
"""Condense rocprofv3 output dirs (tools/profile.sh) into a small text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = name.replace('void vp::', '').replace('vp::', '')
    return name[:110]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, pattern), recursive=True))


for f in find('stats/**/*kernel_stats.csv'):
    print(f'== kernel stats ({os.path.relpath(f, root)})')
    rows = list(csv.DictReader(open(f)))
    for r in rows[:25]:
        print(f"{short(r['Name']):110s} calls {r['Calls']:>6s} total_ns {r['TotalDurationNs']:>12s} avg_ns {float(r['AverageNs']):12.1f} pct {r['Percentage']}")
for d in ('pmc_sq', 'pmc_tcc', 'pmc_fetch', 'pmc_write'):
    for f in find(f'{d}/**/*counter_collection.csv'):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            key = (k, r['Dispatch_Id'])
            if key not in seen:
                seen.add(key)
                cnt[k] += 1
        print(f'== {d}: per-kernel counter sums / dispatch counts')
        for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:14]:
            vals = '  '.join(f'{c}={v / cnt[k]:.4g}' for c, v in sorted(acc[k].items()))
            print(f'{k:110s} n={cnt[k]:4d}  per-dispatch: {vals}')
