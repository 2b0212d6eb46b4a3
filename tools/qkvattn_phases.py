#!/usr/bin/env python3
"""Where a launch of the fused qkv + attention kernel (csrc/qkvattn.hip) spends its time: the whole kernel and the kernel with one phase
compiled out (measurement build; random operands; ViTPose-B shape, 256 crops -> 128 pairs x 12 heads = 1536 tiles = 6 per CU)."""
import ctypes as C
import sys
import _toolslib  # noqa: F401
from easy_vitpose_amd import _capi as capi

lib = capi.load_library()
lib.vp_dbg_qkvattn_bench.argtypes = [C.c_int32] * 6 + [C.POINTER(C.c_float)]
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
D, heads = (int(sys.argv[2]), int(sys.argv[2]) // 64) if len(sys.argv) > 2 else (768, 12)
if len(sys.argv) > 3:   # head dim 80 (ViTPose-H: 1280 16): gemm8.hip EPI_QKV_ATTN, one crop x one head per 192 x 256 tile; ablate 2 = no fold / LDS hand-over, 16 = no attention phase, 8 = no y stores
    heads = int(sys.argv[3])
    for name, abl in [('whole kernel', 0), ('no attention phase', 16), ('no fold + hand-over, no attention phase', 18), ('no y stores', 8), ('no operand DMA behind the ring start', 1), ('whole kernel', 0)]:
        ms = C.c_float()
        capi.check(lib.vp_dbg_qkvattn_bench(0, npairs, D, heads, 30, abl, C.byref(ms)))
        print(f'{name:60s} {1e3 * ms.value:8.1f} us')
    sys.exit(0)
for name, abl in [('whole kernel', 0), ('no attention phase', 1), ('no epilogue + no attention', 3), ('K-loop of 4 K-tiles only', 4), ('4 K-tiles, no epilogue, no attention (restart + boundary)', 7),
                  ('whole kernel', 0)]:
    ms = C.c_float()
    capi.check(lib.vp_dbg_qkvattn_bench(0, npairs, D, heads, 30, abl, C.byref(ms)))
    print(f'{name:60s} {1e3 * ms.value:8.1f} us')
