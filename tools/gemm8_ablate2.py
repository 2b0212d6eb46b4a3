#!/usr/bin/env python3
"""Where a K-tile of the 8-phase kernel goes: the same launch with pieces removed.  Run-time switches (GemmArgs::ablate: 1 = no
operand DMA after the prologue, 8 = no epilogue stores) x diagnosis BUILDS of gemm8.hip (-DVP_G8_ABL: 1 = no fragment reads,
2 = no main-loop barriers, 4 = no MFMAs; results are garbage, timing is the point).   VP_HIP_LIB=<build> python tools/gemm8_ablate2.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402  (measurement build of the library)
from easy_vitpose_amd import _capi as capi
lib = capi.load_library()
M = 49152
for shape, epi, v, N, K in [('qkv', 0, 16, 2304, 768), ('fc1', 1, 16, 3072, 768), ('fc2', 6, 17, 768, 3072)]:
    out = []
    for ab in (0, 1, 8, 9):
        ms = C.c_float()
        rc = lib.vp_dbg_gemm_bench(0, 0, epi, v | (ab << 8), 4, M, N, K, 10, C.byref(ms))
        out.append(f'abl{ab}={ms.value * 1e3:7.1f}us' + ('' if rc == 0 else f'(rc={rc})'))
    print(f'{os.path.basename(os.environ.get("VP_HIP_LIB", "default")):10s} {shape:4s}', ' '.join(out), flush=True)
