#!/usr/bin/env python3
"""Main-loop rate of each tile configuration with the prologue / epilogue amortised away (large K):
how fast a persistent kernel with overlapped tile boundaries could at best run.  GPU box only."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402  (measurement build of the library)
from easy_vitpose_amd import _capi as capi

lib = capi.load_library()
M = 49152
for N in (3072, 2304):
    for K in (768, 3072, 12288):
        for v, gm in ((8, 8), (3, 8), (2, 8), (6, 8), (10, 8), (7, 8)):
            ms = C.c_float()
            rc = lib.vp_dbg_gemm_bench(0, 0, 0, v, gm, M, N, K, 4, C.byref(ms))
            if rc:
                print(f'N={N} K={K} cfg{v}: err {rc}')
                continue
            print(f'N={N} K={K:6d} cfg{v:2d}: {2.0 * M * N * K / (ms.value * 1e-3) / 1e12:7.1f} TF/s  ({ms.value * 1e3:.0f} us)', flush=True)
