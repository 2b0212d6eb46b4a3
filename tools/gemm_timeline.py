#!/usr/bin/env python3
"""Where does a tile's time go?  Per-tile phase stamps of the persistent qkv / fc1 GEMM (wave 0 of every workgroup):
main loop vs epilogue vs gap, in shader cycles.  GPU box only."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402  (measurement build of the library)
from easy_vitpose_amd import _capi as capi

lib = capi.load_library()
M, K = 49152, 768
for name, epi, N in (('qkv', 0, 2304), ('fc1', 1, 3072)):
    wg = 512
    st = np.zeros((wg, 32, 8), dtype=np.uint64)
    rc = lib.vp_dbg_gemm_timeline(0, 0, epi, M, N, K, st.ctypes.data_as(C.POINTER(C.c_uint64)), wg)
    assert rc == 0, capi.last_error()
    ntile = int((st[:, :, 0] != 0).sum(1).max())
    s = st[:, :ntile].astype(np.int64)
    loop = s[:, :, 1] - s[:, :, 0]
    epi_c = s[:, :, 2] - s[:, :, 1]
    gap = s[:, 1:, 0] - s[:, :-1, 2]
    total = s[:, -1, 2] - s[:, 0, 0]
    print(f'{name}: {ntile} tiles per workgroup; cycles per tile (median over workgroups, by tile index):')
    print('   main loop', np.median(loop, 0).astype(int).tolist())
    print('   epilogue ', np.median(epi_c, 0).astype(int).tolist())
    print('   gap      ', np.median(gap, 0).astype(int).tolist())
    ph = np.diff(s[:, :, 3:8], axis=2)        # k-step 5: wait+barrier, DMA issue, reads + first MFMA block, rest
    names = ['vmcnt wait + barrier', '10 global_load_lds issues', 'fragment reads + 12 MFMAs', 'fragment reads + 36 MFMAs']
    for i, nme in enumerate(names):
        print(f'   k-step 5 phase: {nme:28s} median {int(np.median(ph[:, :, i]))} cycles')
    print(f'   workgroup total median {int(np.median(total))} cycles; main loop share {loop.sum() / total.sum():.3f}, '
          f'epilogue share {epi_c.sum() / total.sum():.3f}; per k-step {np.median(loop) / (K // 64):.0f} cycles '
          f'(MFMA work of the two co-resident waves of a SIMD: {2 * 48 * 16} cycles)')
