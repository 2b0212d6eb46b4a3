#!/usr/bin/env python3
"""qkv / fc1 on 256 x 192 tiles (variant 17) instead of 256 x 256 (variant 16): qkv's 1728 wide tiles are 6.75 per CU (a 75 %-full seventh round),
2304 narrow ones are exactly 9.   python tools/tilewidth_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402
from easy_vitpose_amd import _capi as capi

lib = capi.load_library()
M = 49152
for name, epi, flags, N, K in (('qkv', 0, 16, 2304, 768), ('fc1', 1, 16 | 2, 3072, 768)):
    row = []
    for v in (16, 17):
        for gm in (2, 4, 8):
            best = 1e9
            for _ in range(3):
                ms = C.c_float()
                rc = lib.vp_dbg_gemm_bench2(0, 0, epi, v, gm, flags, M, N, K, 30, C.byref(ms))
                if rc:
                    best = float('nan')
                    break
                best = min(best, ms.value * 1e3)
            row.append(f'v{v} g{gm}: {best:.1f}')
    print(f'{name:4s} ' + '  '.join(row), flush=True)
