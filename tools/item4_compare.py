#!/usr/bin/env python3
"""VERDICT r2 item 4, the measurements it asks for (GPU box only):

 (a) the 8-phase kernel's main loop beside the guide's 256^2 8-phase template: gemm8_kernel with the PLAIN epilogue (+bias, 16-bit
     out) at 4096^3 and 8192^3 on uniform random operands -- the template's quoted numbers are ~1320-1340 TF at 4096^3 and ~1470 TF at
     8192^3 on the same kind of data (cdna_hip_programming.md, "The 256^2 8-phase template");
 (b) the encoder's wide GEMMs (qkv, fc1 at batch 256, production epilogues: LayerNorm fold, GELU, blocked output) on the 8-phase
     kernel (one 512-thread workgroup per CU) against the existing TWO-workgroups-per-CU kernels of the same library -- 192x128 tiles as
     4 waves of 96x64 (cfg8, persistent ring) and as 8 waves of 48x64 = 4 waves per SIMD (cfg11): the geometry item 4 proposes (two
     workgroups per CU overlapping each other's epilogues, 4 waves per SIMD) measured on this code base's own kernels.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402  (measurement build of the library)
from easy_vitpose_amd import _capi as capi

lib = capi.load_library()


def bench2(dtype, epi, variant, gm, flags, M, N, K, iters=10):
    ms = C.c_float()
    rc = lib.vp_dbg_gemm_bench2(0, dtype, epi, variant, gm, flags, M, N, K, iters, C.byref(ms))
    return None if rc else ms.value


print('== (a) plain epilogue, square shapes, uniform random operands in [-1, 1): TFLOP/s (us)')
for dtype, name in ((1, 'bf16'), (0, 'fp16')):
    for S in (4096, 8192):
        row = []
        for label, v, fl in (('gemm8 256x256 (8-phase)', 16, 0), ('cfg3 256x256 2-phase', 3, 0), ('cfg8 192x128 2 wg/CU', 8, 0)):
            if S % 192 and v == 8:
                continue
            ms = bench2(dtype, 0, v, 8, fl, S, S, S)
            row.append(f'{label}: ' + ('err' if ms is None else f'{2.0 * S ** 3 / (ms * 1e-3) / 1e12:7.1f} TF ({ms * 1e3:.0f} us)'))
        print(f'{name} {S}^3  ' + ' | '.join(row), flush=True)
print('   guide template (bf16, same data): ~1320-1340 TF at 4096^3, ~1470 TF at 8192^3')
for K in (768, 4096):
    ms = bench2(0, 0, 16, 8, 0, 49152, 3072, K)
    print(f'fp16 M=49152 N=3072 K={K}: gemm8 plain epilogue {2.0 * 49152 * 3072 * K / (ms * 1e-3) / 1e12:7.1f} TF ({ms * 1e3:.0f} us)', flush=True)

print('== (b) production wide GEMMs at batch 256 (M = 49152, K = 768), fp16, production epilogues: us per launch')
M, D = 49152, 768
FOLD, OUTB, PERS = 16, 2, 1
for name, epi, N, fl in (('qkv', 0, 3 * D, FOLD), ('fc1', 1, 4 * D, FOLD | OUTB)):
    for label, v, gm, f2 in (('gemm8 256x256, 1 wg/CU, 8 waves of 128x64 (shipped)', 16, 4 if name == 'qkv' else 8, 0),
                             ('cfg8 192x128, 2 wg/CU, 4 waves of 96x64, persistent ring', 8, 8, PERS),
                             ('cfg8 192x128, 2 wg/CU, 4 waves of 96x64, one tile per wg', 8, 8, 0),
                             ('cfg11 192x128, 2 wg/CU, 8 waves of 48x64 = 4 waves/SIMD', 11, 8, 0),
                             ('cfg3 256x256, 1 wg/CU, 2-phase', 3, 8, 0)):
        ms = bench2(0, epi, v, gm, fl | f2, M, N, D)
        if ms is None:
            print(f'{name} {label}: not supported')
            continue
        print(f'{name} {label}: {ms * 1e3:7.1f} us = {2.0 * M * N * D / (ms * 1e-3) / 1e12:6.1f} TF', flush=True)
