#!/usr/bin/env python3
"""Tile-walk group size (GemmArgs::group_m: m-tiles per group of the XCD-contiguous walk) of the 8-phase launches at batch 256: us per launch.
    python tools/group_sweep.py [variant] [batch]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402
from easy_vitpose_amd import _capi as capi
from easy_vitpose_amd.configs import VARIANTS

lib = capi.load_library()
variant = sys.argv[1] if len(sys.argv) > 1 else 'b'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
D = VARIANTS[variant][0]
M = batch * 192
res_v = 17 if D % 192 == 0 else 16
print(f'# ViTPose-{variant.upper()} batch {batch}: M={M} D={D}; us per launch (best of 3 x 30 launches) by group_m')
for name, epi, v, flags, N, K in (('qkv', 0, 16, 16, 3 * D, D), ('fc1', 1, 16, 16 | 2, 4 * D, D), ('fc2', 6, res_v, 4 | 8, D, 4 * D), ('fc2 fwd', 6, res_v, 4, D, 4 * D)):
    row = []
    for gm in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48):
        best = 1e9
        for _ in range(3):
            ms = C.c_float()
            rc = lib.vp_dbg_gemm_bench2(0, 0, epi, v, gm, flags, M, N, K, 30, C.byref(ms))
            if rc:
                best = float('nan')
                break
            best = min(best, ms.value * 1e3)
        row.append(f'{gm}: {best:.1f}')
    print(f'{name:8s} ' + '  '.join(row), flush=True)
