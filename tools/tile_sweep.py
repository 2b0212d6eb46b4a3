#!/usr/bin/env python3
"""Which tile should fc1 / fc2 run on at a given batch size?  Times every candidate configuration of the two MLP GEMMs in isolation (vp_dbg_gemm_bench2: production
epilogues and layouts, random operands) over a list of batch sizes and prints, per shape, the launch time of each candidate and what the selection rule of
tile_rules.hip (vp_dbg_gemm8_pick) would pick -- the data behind the rule's thresholds (profiles/tile_sweep_r4.txt).  GPU box only.

    python tools/tile_sweep.py [--variant b] [--batches 40,52,...]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402
from easy_vitpose_amd import _capi as capi
from easy_vitpose_amd.configs import VARIANTS

ap = argparse.ArgumentParser()
ap.add_argument('--variant', default='b')
ap.add_argument('--batches', default='40,52,64,76,85,88,100,112,128,140,152,172,192,200,230')
ap.add_argument('--iters', type=int, default=8)
args = ap.parse_args()
lib = capi.load_library()
lib.vp_dbg_gemm_bench2.argtypes = [C.c_int32] * 10 + [C.c_void_p]
D = VARIANTS[args.variant][0]
OUTB, AB, REV, LNF, PERS = 2, 4, 8, 16, 1


def t_us(epi, variant, gm, flags, M, N, K):
    ms = C.c_float()
    rc = lib.vp_dbg_gemm_bench2(0, capi.DTYPES['fp16'], epi, variant, gm, flags, M, N, K, args.iters, C.byref(ms))
    return ms.value * 1e3 if rc == 0 else None


def fmt(x):
    return '   --  ' if x is None else f'{x:7.1f}'


print(f'# ViTPose-{args.variant.upper()} (D = {D}); launch time in us of ONE GEMM in a loop of {args.iters} (isolated: caches warm with its own operands)')
print('# fc2 (bias + residual planes + row statistics, K = 4 D):  2-phase 192x128 | 256x256 | 256x192 | 192x256 || rule picks')
print('# fc1 (LayerNorm fold + bias + GELU, blocked output, K = D): 2-phase 192x128 (persistent from 1024 tiles) | 256x256 | 192x256 || rule picks')
names = {0: '2-phase', 16: '256x256', 17: '256x192', 18: '192x256'}
for n in [int(x) for x in args.batches.split(',')]:
    M = 192 * n
    tiles = C.c_int32()
    r = [t_us(6, 11, 0, AB | REV, M, D, 4 * D)] + [t_us(6, v, 2, AB | REV, M, D, 4 * D) for v in (16, 17, 18)]
    pick = lib.vp_dbg_gemm8_pick(M, D, 0, 3, C.byref(tiles))
    cands = dict(zip((0, 16, 17, 18), r))
    best = min((v for v in cands.items() if v[1] is not None), key=lambda kv: kv[1])
    print(f'fc2 n={n:4d} M={M:6d} ' + ' '.join(fmt(x) for x in r) + f' || rule: {names[pick]:8s} ({tiles.value:4d} tiles) best: {names[best[0]]:8s} '
          f'{"" if best[0] == pick else f"  <-- rule loses {100 * (cands[pick] / best[1] - 1):.1f} %"}')
    t2 = (M // 192) * (4 * D // 128)
    r = [t_us(1, 8, 8, OUTB | LNF | (PERS if t2 >= 1024 else 0), M, 4 * D, D)] + [t_us(1, v, 8, OUTB | LNF, M, 4 * D, D) for v in (16, 18)]
    pick = lib.vp_dbg_gemm8_pick(M, 4 * D, 1, 3, C.byref(tiles))
    cands = dict(zip((0, 16, 18), r))
    best = min((v for v in cands.items() if v[1] is not None), key=lambda kv: kv[1])
    print(f'fc1 n={n:4d} M={M:6d} ' + ' '.join(fmt(x) for x in r) + f'         || rule: {names[pick]:8s} ({tiles.value:4d} tiles) best: {names[best[0]]:8s} '
          f'{"" if best[0] == pick else f"  <-- rule loses {100 * (cands[pick] / best[1] - 1):.1f} %"}')
