#!/usr/bin/env python3
"""The CU's store path as a gemm8 epilogue uses it (GPU box only): nanoseconds (and cycles at 2.0 GHz) per 1 KiB store instruction and CU, for
4 / 8 / 16 waves per CU, three address patterns of a store instruction (1 KiB contiguous; 16 rows x 64 contiguous bytes; 16 rows x four 16-byte
pieces 32 bytes apart = what the round-2/3 epilogues issue), L2-resident and 1 GiB targets, row stride 128 B and 6144 B."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easy_vitpose_amd import _capi as capi

lib = capi.load_library()
pats = ['1 KiB contiguous', '16 rows x 64 B', '16 rows x 4 x 16 B (epilogue)']
for big in (0, 1):
    for strided in (0, 1):
        print(f'== target {"1 GiB" if big else "L2-resident"}, row stride {6144 if strided else 128} B: ns (cycles at 2.0 GHz) per store instruction and CU')
        for wc, nw in ((0, 4), (1, 8), (2, 16)):
            row = []
            for pat in range(3):
                r = C.c_double()
                rc = lib.vp_dbg_peak(0, 200 + 16 * wc + 4 * pat + 2 * big + strided, C.byref(r))
                row.append(f'{pats[pat]}: ' + ('err' if rc else f'{r.value:6.1f} ns ({r.value * 2.0:5.0f})'))
            print(f'  {nw:2d} waves/CU  ' + ' | '.join(row), flush=True)
