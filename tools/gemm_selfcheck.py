#!/usr/bin/env python3
"""A kernel configuration against ITSELF (and against another tile configuration) on random operands: any mismatch is a race or an
uninitialised read.  GPU box only.   python tools/gemm_selfcheck.py [D] [batch]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402  (measurement build of the library)
from easy_vitpose_amd import _capi as capi

lib = capi.load_library()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 384
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
M = B * 192
LN, OUTB, AB, REV = 16, 2, 4, 8
bad = 0
for name, epi, N, K, fl in (('qkv', 0, 3 * D, D, LN), ('fc1', 1, 4 * D, D, LN | OUTB), ('proj', 6, D, D, 0), ('fc2', 6, D, 4 * D, AB | REV),
                            ('qkv-noln', 0, 3 * D, D, 0), ('fc1-noln', 1, 4 * D, D, 0)):
    for va, vb in ((9, 9), (1, 1), (9, 1), (8, 9), (11, 9)):
        nm, md = C.c_uint64(), C.c_double()
        rc = lib.vp_dbg_gemm_compare(0, 0, epi, va, 0, fl, vb, 0, fl, M, N, K, 3, C.byref(nm), C.byref(md))
        bad += (rc != 0) or nm.value != 0
        print(f'{name:9s} M={M} N={N} K={K} cfg{va} vs cfg{vb}: rc={rc} mismatches={nm.value} max|d|={md.value:.3e} {capi.last_error() if rc else ""}', flush=True)
sys.exit(1 if bad else 0)
