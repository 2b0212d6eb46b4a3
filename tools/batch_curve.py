#!/usr/bin/env python3
"""The step over the batch: ms per step, us per layer and the kernels of qkv | fc1 | proj | fc2 at every batch size of a range, with the two signs of a tile rule at a bad spot
flagged -- a batch that is FASTER than the next smaller one, and a per-crop cost that jumps by more than 3 % for one more crop.  This is the scan behind round 6's tile rules,
encoder padding and fold rule (profiles/small_batch_r6.txt calls 17, 20, 27): run it, re-sweep every flagged size with tools/small_sweep.py --sets (every instantiated tile, in
situ, bit-identity checked), turn what wins into a rule of csrc/tile_rules.hip, walk the rule in tests/test_host_logic.py.
GPU box:   python tools/batch_curve.py b:coco:1-72 l:coco_25:1-72 [--iters 30]        (runs tools/small_sweep.py underneath: measurement build)
anywhere:  python tools/batch_curve.py --view gpurun_out/scan.txt [--model B]          (re-reads a saved scan)"""
import argparse
import os
import re
import subprocess
import sys


def short(k):
    k = re.sub(r'[<>]', '', k.strip()).replace('qkvattn_kernelF16', 'fused qkv+attn').replace('qkvattn_kernelBF16', 'fused qkv+attn')
    m = re.match(r'gemm8_kernelB?F16, (\d+), G8(\d+), (\d+)', k)
    if m:
        return f'8-phase {m.group(3)}x{m.group(2)}'
    m = re.match(r'(\d+), (\d+), 64, (\d+), (\d+), (\d), (\d), 0', k)
    if m:
        bm, bn, wm, wn = (int(m.group(i)) for i in range(1, 5))
        return f'{bm}x{bn}/{m.group(5)}' + ('w8' if (bm // wm) * (bn // wn) == 8 else '') + ('k2' if m.group(6) == '6' else '')
    return k[:24]


def view(lines, model=None):
    cur = prev = None
    for line in lines:
        m = re.match(r'# ViTPose-(\w+) / (\w+), (\d+) crops', line)
        if m:
            cur = (m.group(1), int(m.group(3)))
            continue
        m = re.match(r'\S+\s+([\d.]+) ms \|\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+) \| (.*)', line)
        if not (m and cur) or (model and cur[0] != model.upper()):
            continue
        ks = [short(k) for k in m.group(7).split('|')][:4]
        ms, n = float(m.group(1)), cur[1]
        flag = ''
        if prev and prev[0] == cur[0]:
            if ms < prev[2]:
                flag = '   <-- faster than the smaller batch'
            elif (ms / n) / (prev[2] / prev[1]) > 1.03:
                flag = '   <-- per crop +%.0f %%' % (((ms / n) / (prev[2] / prev[1]) - 1) * 100)
        us = [float(m.group(i)) for i in range(2, 7)]
        print(f'{cur[0]} {n:3d}  {ms:7.3f} ms {ms / n * 1e3:7.1f} us/crop  qkv {us[0]:5.1f} fc1 {us[1]:5.1f} proj {us[2]:5.1f} fc2 {us[3]:5.1f} att {us[4]:4.1f} | ' + ' | '.join(ks) + flag)
        prev = (cur[0], n, ms)


ap = argparse.ArgumentParser()
ap.add_argument('ranges', nargs='*', help='variant:dataset:lo-hi, e.g. b:coco:1-72')
ap.add_argument('--iters', type=int, default=30)
ap.add_argument('--view', default='')
ap.add_argument('--model', default='')
args = ap.parse_args()
if args.view:
    view(open(args.view).read().split('\n'), args.model or None)
    sys.exit(0)
cases = []
for r in args.ranges:
    v, d, span = r.split(':')
    lo, _, hi = span.partition('-')
    cases += [f'{v}:{d}:{n}' for n in range(int(lo), int(hi or lo) + 1)]
if not cases:
    ap.error('give at least one range or --view FILE')
here = os.path.dirname(os.path.abspath(__file__))
out = subprocess.run([sys.executable, os.path.join(here, 'small_sweep.py'), '--iters', str(args.iters), '--cases', ','.join(cases), '--sets', 'default='],
                     capture_output=True, text=True).stdout
view(out.split('\n'), args.model or None)
