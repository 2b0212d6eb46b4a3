#!/usr/bin/env python3
"""What the residual epilogue's STORES cost (8-phase kernel, 256 x 192 tiles, GPU box, measurement build): the launch as shipped, without the
stores (ablate 8), and with every store redirected into a 2 MB window that stays in L2 (ablate 128: issue + acknowledgement, no HBM write-back)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402  (measurement build of the library)
from easy_vitpose_amd import _capi as capi

lib = capi.load_library()
M, D = 49152, 768
AB, REV = 4, 8
for name, K, fl in (('proj', D, 0), ('fc2', 4 * D, AB | REV)):
    row = []
    for label, abl in (('shipped', 0), ('no stores', 8), ('stores into an L2-resident 2 MB window', 128), ('no operand DMA', 1), ('no DMA, no stores', 9)):
        ms = C.c_float()
        rc = lib.vp_dbg_gemm_bench2(0, 0, 6, 17 | (abl << 8), 2, fl, M, D, K, 12, C.byref(ms))
        row.append(f'{label}: ' + ('err' if rc else f'{ms.value * 1e3:6.1f} us'))
    print(f'{name:4s} ' + ' | '.join(row), flush=True)
