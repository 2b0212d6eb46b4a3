#!/usr/bin/env python3
"""Two cheap ablations of the GEMM (flags in GemmArgs::ablate): 1 = no operand loads after the prologue,
8 = no epilogue stores.  Shows how much of a launch is operand delivery / epilogue / the bare MFMA loop."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402  (measurement build of the library)
from easy_vitpose_amd import _capi as capi
lib = capi.load_library()
M = 49152
names = {0: 'full', 1: 'no operand loads', 2: 'A rows always L2-resident', 8: 'no epilogue stores', 10: 'A L2-resident + no epilogue stores', 9: 'neither (MFMA + LDS reads + barriers)'}
for shape, epi, N, K in [('qkv', 0, 2304, 768), ('fc1', 1, 3072, 768), ('proj', 2, 768, 768), ('fc2', 2, 768, 3072)]:
    for v in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '8').split(',')]:
        for ab in (0, 2, 8, 10, 1, 9):
            ms = C.c_float()
            rc = lib.vp_dbg_gemm_bench(0, 0, epi, v | (ab << 8), 8, M, N, K, 5, C.byref(ms))
            tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
            print(f'{shape:4s} cfg{v} {names[ab]:40s} {ms.value*1e3:8.1f} us  ({tf:7.1f} TF/s-equivalent) rc={rc}', flush=True)
