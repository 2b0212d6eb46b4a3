#!/usr/bin/env python3
"""Condense the rocprofv3 (rocpd sqlite) outputs of tools/profile.sh into a small text
summary that is committed under profiles/.

    python tools/summarize_profile.py gpurun_out/prof_<tag> > profiles/<name>.txt
"""
import glob
import os
import re
import sqlite3
import sys

root = sys.argv[1]
json_out = sys.argv[2] if len(sys.argv) > 2 else None
pmc_json = {}


def short(name):
    name = name.replace('void ', '').replace('(anonymous namespace)::', '').replace('vp::', '')
    name = re.sub(r'\(.*$', '', name).replace(' >', '>')
    return name[:100]


def dbs(sub):
    return sorted(glob.glob(os.path.join(root, sub, '**', '*.db'), recursive=True))


for db in dbs('stats'):
    cur = sqlite3.connect(db).cursor()
    print(f'== rocprofv3 --kernel-trace --stats : {os.path.relpath(db, root)}  (durations in us)')
    print(f"{'kernel':100s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, total, avg, pct in cur.execute('select name,total_calls,total_duration,average,percentage from top_kernels limit 16'):
        print(f'{short(name):100s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:6.2f}')
for sub in ('pmc_sq', 'pmc_tcc', 'pmc_fetch', 'pmc_write'):
    for db in dbs(sub):
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute('''select kernel_name, counter_name, sum(value), count(distinct dispatch_id), avg(duration)
                              from counters_collection group by kernel_name, counter_name''').fetchall()
        per = {}
        for k, c, v, n, dur in rows:
            per.setdefault(k, {'n': n, 'dur': dur})[c] = v / n
        print(f'== rocprofv3 --pmc ({sub}): per-dispatch averages (sum over SEs/XCDs / dispatches)')
        for k in sorted(per, key=lambda k: -per[k]['dur'] * per[k]['n'])[:12]:
            d = per[k]
            pmc_json.setdefault(short(k), {}).update({c: v for c, v in d.items() if c not in ('n', 'dur')})
            pmc_json[short(k)].setdefault('avg_ns_' + sub, d['dur'])
            vals = '  '.join(f'{c}={v:.5g}' for c, v in sorted(d.items()) if c not in ('n', 'dur'))
            print(f"{short(k):100s} n={d['n']:4d} avg_ns={d['dur']:.0f}  {vals}")

if json_out:
    import json
    # HBM traffic per dispatch, MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are KiB from separate --pmc
    # passes; on gfx950 FETCH_SIZE counts 64 B per 128-B request -> x2 (calibrated here on layernorm: 151 MB read).
    for k, d in pmc_json.items():
        if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
            d['hbm_bytes_per_dispatch'] = (2.0 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024.0
    json.dump(pmc_json, open(json_out, 'w'), indent=1, sort_keys=True)
