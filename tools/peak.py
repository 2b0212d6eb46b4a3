#!/usr/bin/env python3
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402  (measurement build of the library)
from easy_vitpose_amd import _capi as capi
lib = capi.load_library()
for kind, name in [(0, 'MFMA f16 16x16x32 only (TFLOP/s)'), (1, 'MFMA f16 32x32x16 only (TFLOP/s)'), (2, 'float4 copy read+write (TB/s)'),
                   (3, 'LDS-DMA stream, 32 MiB L2/MALL-resident source, 2 blocks/CU (TB/s into LDS)'),
                   (4, 'LDS-DMA stream, 1 GiB source (TB/s into LDS)'),
                   (5, 'LDS-DMA stream, 2 MiB source = L2-resident per XCD (TB/s into LDS)'),
                   (6, 'LDS-DMA stream, 256 KiB source (TB/s into LDS)')] + [(7 + i, f'LDS-DMA stream, {m} MiB source (TB/s into LDS): memory-side cache?') for i, m in enumerate((64, 96, 128, 192, 256, 384))]:
    r = C.c_double()
    rc = lib.vp_dbg_peak(0, kind, C.byref(r))
    print(f'{name}: {r.value:.1f} rc={rc}', flush=True)
