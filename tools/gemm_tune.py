#!/usr/bin/env python3
"""Sweep GEMM tile configurations x tile orders on the ViTPose shapes (GPU box only).

    python tools/gemm_tune.py [--batch 256] [--variant b] > gpurun_out/gemm_tune.txt
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402  (measurement build of the library)
from easy_vitpose_amd import _capi as capi
from easy_vitpose_amd.configs import VARIANTS

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--variant', default='b')
ap.add_argument('--dtype', default='fp16')
ap.add_argument('--iters', type=int, default=6)
ap.add_argument('--cfgs', default='0,1,2,3,4,5,6,7,8,9,10')
ap.add_argument('--groups', default='0,8')
args = ap.parse_args()
lib = capi.load_library()
D = VARIANTS[args.variant][0]
M = args.batch * 192
shapes = [('qkv', 0, 3 * D, D), ('proj', 2, D, D), ('fc1', 1, 4 * D, D), ('fc2', 2, D, 4 * D)]
print(f'# M={M} D={D} dtype={args.dtype}; TFLOP/s per (tile cfg, group_m)')
for name, epi, N, K in shapes:
    best = None
    for v in [int(x) for x in args.cfgs.split(',')]:
        row = []
        for gm in [int(x) for x in args.groups.split(',')]:
            ms = C.c_float()
            rc = lib.vp_dbg_gemm_bench(0, capi.DTYPES[args.dtype], epi, v, gm, M, N, K, args.iters, C.byref(ms))
            if rc != 0:
                row.append(f'err{rc}')
                continue
            tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
            row.append(f'{tf:7.1f}')
            if best is None or tf > best[0]:
                best = (tf, v, gm, ms.value)
        print(f'{name:5s} N={N:5d} K={K:5d} cfg{v}: ' + ' '.join(row), flush=True)
    print(f'{name:5s} BEST {best[0]:.1f} TF/s cfg{best[1]} group_m={best[2]} ({best[3]*1e3:.1f} us)', flush=True)
