#!/usr/bin/env python3
"""Small-batch GEMM shapes (a few crops per GPU: one GPU's share of a sharded frame) across the small tile configurations:
time per launch and bit identity against Cfg9.  GPU box only.   python tools/gemm_small.py [variant] [crops]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402  (measurement build of the library)
from easy_vitpose_amd import _capi as capi
from easy_vitpose_amd.configs import VARIANTS

lib = capi.load_library()
variant = sys.argv[1] if len(sys.argv) > 1 else 'l'
crops = int(sys.argv[2]) if len(sys.argv) > 2 else 8
D = VARIANTS[variant][0]
M = crops * 192
cfgs = [9, 14, 12, 15, 1, 13, 0, 8, 11]
print(f'# ViTPose-{variant.upper()} {crops} crops: M={M} D={D}; us per launch by tile configuration (* = bit-identical to cfg9)')
for name, epi, N, K, fl in (('qkv', 0, 3 * D, D, 16), ('fc1', 1, 4 * D, D, 16 | 2), ('proj', 6, D, D, 0), ('fc2', 6, D, 4 * D, 4 | 8)):
    row = []
    for v in cfgs:
        ms = C.c_float()
        rc = lib.vp_dbg_gemm_bench2(0, 0, epi, v, 0, fl, M, N, K, 20, C.byref(ms))
        if rc:
            row.append(f'cfg{v}: err')
            continue
        nm, md = C.c_uint64(), C.c_double()
        rc2 = lib.vp_dbg_gemm_compare(0, 0, epi, v, 0, fl, 9, 0, fl, M, N, K, 1, C.byref(nm), C.byref(md))
        row.append(f'cfg{v}: {ms.value * 1e3:.1f}{"*" if rc2 == 0 and nm.value == 0 else "!"}')
    print(f'{name:5s} N={N:5d} K={K:5d}  ' + '  '.join(row), flush=True)
