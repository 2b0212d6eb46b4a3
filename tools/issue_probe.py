#!/usr/bin/env python3
"""What one global store / LDS-DMA load / block of VALU work per 16 MFMAs costs a wave whose matrix pipe is saturated
(elementwise.hip issue_probe_kernel), with one and with two waves per SIMD.  Output: ns per round of 16 MFMAs."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402  (measurement build of the library)
from easy_vitpose_amd import _capi as capi
lib = capi.load_library()
names = {0: 'MFMA only', 1: '+1 store', 2: '+1 DMA', 3: '+1 store +1 DMA', 4: '+48 VALU', 5: '+1 store +48 VALU', 6: '+1 DMA +48 VALU',
         7: '+1 store +1 DMA +48 VALU', 9: '+4 stores', 11: '+4 stores +1 DMA', 15: '+4 stores +1 DMA +48 VALU',
         17: '+1 store (L2-resident)', 25: '+4 stores (L2-resident)', 18: '+1 DMA (L2-resident)', 27: '+4 stores +1 DMA (L2-resident)',
         31: '+4 stores +1 DMA +48 VALU (L2-res.)'}
for two in (0, 32):
    base = None
    for mode, name in names.items():
        r = C.c_double()
        rc = lib.vp_dbg_peak(0, 100 + mode + two, C.byref(r))
        base = base or r.value
        print(f'{2 if two else 1} wave(s)/SIMD  {name:32s}: {r.value:8.1f} ns/round  (+{r.value - base:6.1f})  rc={rc}', flush=True)

for two in (0, 4):
    for dep, name in ((0, '8 accumulators'), (1, '4 accumulators'), (2, '2 accumulators')):
        r = C.c_double()
        rc = lib.vp_dbg_peak(0, 170 + dep + two, C.byref(r))
        print(f'{2 if two else 1} wave(s)/SIMD  32x32x16 MFMA only, {name:20s}: {r.value:8.1f} ns/round (8 MFMAs = the flops of 16 16x16x32)  rc={rc}', flush=True)
