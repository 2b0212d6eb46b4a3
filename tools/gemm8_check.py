#!/usr/bin/env python3
"""8-phase GEMM (gemm8.hip, variants 16 / 17) against the 2-phase kernels of gemm.hip on the production shapes (GPU box only):
  * bit identity of every output element and row statistic on the same random operands (same accumulation order by
    construction, so ANY difference is a schedule / race bug), repeated `--reps` times;
  * time per launch of both, plus ablations of the new kernel (1 = no operand DMA after the prologue, 8 = no stores).
    python tools/gemm8_check.py [--batch 256] [--variant b] [--reps 3] [--no-compare] [--no-bench]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402  (measurement build of the library)
from easy_vitpose_amd import _capi as capi
from easy_vitpose_amd.configs import VARIANTS

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--variant', default='b')
ap.add_argument('--dtype', default='fp16')
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--iters', type=int, default=8)
ap.add_argument('--no-compare', action='store_true')
ap.add_argument('--no-bench', action='store_true')
ap.add_argument('--ablate', action='store_true')
args = ap.parse_args()
lib = capi.load_library()
dt = capi.DTYPES[args.dtype]
D = VARIANTS[args.variant][0]
M = args.batch * 192
LN, OUTB, AB, REV, PERS = 16, 2, 4, 8, 1
bn_res = 17 if (D % 192 == 0 and (M // 256) * (D // 192) % 256 == 0) else (16 if D % 256 == 0 else 17)
bn_wide = lambda n: 16 if n % 256 == 0 else 17
# name, epi, N, K, new variant, new flags, old variant, old group, old flags
cases = [
    ('qkv', 0, 3 * D, D, bn_wide(3 * D), LN, 8, 8, LN | PERS),
    ('fc1', 1, 4 * D, D, bn_wide(4 * D), LN | OUTB, 8, 8, LN | OUTB | PERS),
    ('proj', 6, D, D, bn_res, 0, 11, 0, 0),
    ('fc2', 6, D, 4 * D, bn_res, AB | REV, 11, 0, AB | REV),
]
ok = True
print(f'# ViTPose-{args.variant.upper()} batch {args.batch}: M={M} D={D} dtype={args.dtype}', flush=True)
if not args.no_compare:
    for name, epi, N, K, nv, nf, ov, og, of in cases:
        if M % 256 or (nv == 16 and N % 256) or (nv == 17 and N % 192):
            print(f'{name}: shape not supported by gemm8, skipped')
            continue
        nm, md = C.c_uint64(), C.c_double()
        if of & PERS and (M // 192) * (N // 128) < 1024:
            of &= ~PERS
        rc = lib.vp_dbg_gemm_compare(0, dt, epi, nv, 8, nf, ov, og, of, M, N, K, args.reps, C.byref(nm), C.byref(md))
        good = rc == 0 and nm.value == 0
        ok &= good
        print(f'compare {name:4s} N={N:5d} K={K:5d} variant {nv} vs cfg{ov}: rc={rc} mismatches={nm.value} max|d|={md.value:.3e} '
              f'{"IDENTICAL" if good else "MISMATCH " + capi.last_error()}', flush=True)
if not args.no_bench:
    for name, epi, N, K, nv, nf, ov, og, of in cases:
        rows = [('old cfg%d' % ov, ov, og, of), ('gemm8 v%d' % nv, nv, 8, nf)]
        if args.ablate:
            rows += [('gemm8 no operand DMA', nv | (1 << 8), 8, nf), ('gemm8 no stores', nv | (8 << 8), 8, nf), ('gemm8 neither', nv | (9 << 8), 8, nf)]
        for label, v, gm, fl in rows:
            ms = C.c_float()
            if fl & PERS and (M // 192) * (N // 128) < 1024:
                fl &= ~PERS
            rc = lib.vp_dbg_gemm_bench2(0, dt, epi, v, gm, fl, M, N, K, args.iters, C.byref(ms))
            if rc:
                print(f'bench {name:4s} {label:24s}: rc={rc} {capi.last_error()}', flush=True)
                continue
            tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
            print(f'bench {name:4s} N={N:5d} K={K:5d} {label:24s} {ms.value * 1e3:8.1f} us  {tf:7.1f} TF/s', flush=True)
sys.exit(0 if ok else 1)
