#!/usr/bin/env python3
"""Where a gemm8 workgroup's time goes (GPU box only): cycle stamps of waves 0 and 4 of every workgroup of one qkv / fc1 launch.
    python tools/gemm8_timeline.py [--ablate N] (N: 8 = no stores, 64 = non-temporal stores)   env VP_G8_STAGGER=n"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402  (measurement build of the library)
from easy_vitpose_amd import _capi as capi

ap = argparse.ArgumentParser()
ap.add_argument('--ablate', type=int, default=0)
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--resid', action='store_true', help='the residual GEMMs (attn.proj, mlp.fc2): main loop / drain / passes / statistics / ring restart per tile')
args = ap.parse_args()
lib = capi.load_library()
M, D = args.batch * 192, 768
if args.resid:
    for name, N, K, variant, flags in (('proj', D, D, 17, 0), ('fc2', D, 4 * D, 17, 4 | 8)):
        st = np.zeros((256, 2, 16, 8), dtype=np.uint64)
        rc = lib.vp_dbg_gemm8_timeline(0, 0, 6, variant, flags, args.ablate, M, N, K, st.ctypes.data_as(C.POINTER(C.c_uint64)), 256)
        if rc:
            print(name, 'rc', rc, capi.last_error())
            continue
        s = st.astype(np.int64)
        ntile = int((s[0, 0, :, 0] > 0).sum())
        t0 = s[:, :, :ntile, :]
        names = ['main loop', 'drain + re-align', 'staging passes', 'statistics flush', 'ring restart']
        print(f'{name}: {ntile} tiles per workgroup, medians over 256 workgroups (shader cycles)')
        for grp in (0, 1):
            d = np.diff(t0[:, grp, :, :6], axis=2)
            print(f'  group {grp}: ' + '; '.join(f'{n} {np.median(d[:, :, i], 0).astype(int).tolist()}' for i, n in enumerate(names)))
        print(f'  launch span {int(t0[:, 0, ntile - 1, 5].max() - t0[:, 0, 0, 0].min())} cycles')
    sys.exit(0)
for name, epi, N, flags in (('qkv', 0, 3 * D, 16), ('fc1', 1, 4 * D, 16 | 2)):
    st = np.zeros((256, 2, 16, 8), dtype=np.uint64)
    rc = lib.vp_dbg_gemm8_timeline(0, 0, epi, 16, flags, args.ablate, M, N, D, st.ctypes.data_as(C.POINTER(C.c_uint64)), 256)
    if rc:
        print(name, 'rc', rc, capi.last_error())
        continue
    s = st.astype(np.int64)
    ntile = int((s[0, 0, :, 0] > 0).sum())
    t0 = s[:, :, :ntile, :]
    loop = t0[..., 1] - t0[..., 0]
    epi_t = t0[..., 2] - t0[..., 1]
    start = t0[:, 0, 0, 0].min()
    print(f'{name}: {ntile} tiles per workgroup (stamps of wave 0 / wave 4, medians over 256 workgroups, shader cycles)')
    for grp in (0, 1):
        print(f'  group {grp}: main loop {np.median(loop[:, grp], 0).astype(int).tolist()}')
        print(f'           epilogue  {np.median(epi_t[:, grp], 0).astype(int).tolist()}')
    gap = t0[:, 0, 1:, 0] - t0[:, 0, :-1, 2]
    print(f'  gap epilogue end -> next loop begin {np.median(gap, 0).astype(int).tolist()}')
    tot = t0[:, 0, ntile - 1, 2].max() - start
    print(f'  launch span {tot} cycles; epilogue-begin spread per tile (max - min over workgroups): {(t0[:, 0, :, 1].max(0) - t0[:, 0, :, 1].min(0)).tolist()}')
    sec = s[:, :, 8:10, :].reshape(256, 2, 16)[:, :, :12]
    if (sec[:, 0, 1] > 0).any():   # diagnosis build (-DVP_G8_ABL=16): section stamps of K-tile 4 of each workgroup's second tile
        names = ['LA reads', 'LA dma X1', 'LA waits', 'bar', 'MA mfma', 'bar', 'LB reads + dma X0 W0 W1', 'LB waits', 'bar', 'MB mfma', 'bar']
        for grp in (0, 1):
            d = np.diff(sec[:, grp, :], axis=1)
            ok = (sec[:, grp, 1] > 0)
            med = np.median(d[ok], 0).astype(int)
            print(f'  group {grp} sections of one K-tile (median cycles): ' + ', '.join(f'{n}={v}' for n, v in zip(names, med)) + f'  total {int(med.sum())}')
