#!/usr/bin/env python3
"""Start stagger of the 8-phase kernel's workgroups (tools build, env VP_G8_STAGGER = n: XCD x starts x n 1024 cycles late; 100 + n:
workgroup j of every XCD starts j n 1024 cycles late): time per launch of the production qkv / fc1 / fc2 launches.
    for n in 0 1 2 4 101 102; do VP_G8_STAGGER=$n python tools/stagger_probe.py; done"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402
from easy_vitpose_amd import _capi as capi

lib = capi.load_library()
M = 49152
out = []
for name, epi, variant, group, flags, N, K in (('qkv', 0, 16, 4, 16, 2304, 768), ('fc1', 1, 16, 8, 16 | 2, 3072, 768), ('fc2', 6, 17, 2, 4 | 8, 768, 3072)):
    best = 1e9
    for _ in range(3):
        ms = C.c_float()
        rc = lib.vp_dbg_gemm_bench2(0, 0, epi, variant, group, flags, M, N, K, 40, C.byref(ms))
        assert rc == 0, capi.last_error()
        best = min(best, ms.value * 1e3)
    out.append(f'{name} {best:7.1f} us')
print(f'VP_G8_STAGGER={os.environ.get("VP_G8_STAGGER", "-"):>4s}: ' + '   '.join(out), flush=True)
