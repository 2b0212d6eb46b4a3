#!/usr/bin/env python3
"""How exact is the accumulation inside v_mfma_f32_16x16x128_f8f6f4?  Per-row deviation of the fp8 probe GEMM from the fp64 product
of the SAME e4m3 codes (GPU box only; the product library's vp_dbg_fp8_gemm)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easy_vitpose_amd import _capi as capi

lib = capi.load_library()
F8 = torch.float8_e4m3fn
rng = np.random.default_rng(0)
M, N, K = 64, 64, 768
for case in ('normal', 'one outlier x1e5 in every row', 'one outlier x100', 'constant magnitude, random sign', 'K = 128 (one instruction)'):
    k = 128 if case.startswith('K = 128') else K
    A = rng.standard_normal((M, k)).astype(np.float32)
    W = rng.standard_normal((N, k)).astype(np.float32)
    if 'x1e5' in case:
        A[:, 17] = 1e5
    if 'x100' in case:
        A[:, 17] = 100.0
    if case.startswith('constant'):
        A = np.sign(A).astype(np.float32)
        W = np.sign(W).astype(np.float32)
    sa = (np.abs(A).max(1) / 448).astype(np.float32)
    sw = (np.abs(W).max(1) / 448).astype(np.float32)
    out = np.empty((M, N), np.float32)
    ca = np.empty((M, k), np.uint8)
    cw = np.empty((N, k), np.uint8)
    rc = lib.vp_dbg_fp8_gemm(0, M, N, k, A.ctypes.data, sa.ctypes.data, W.ctypes.data, sw.ctypes.data, out.ctypes.data, ca.ctypes.data, cw.ctypes.data)
    assert rc == 0, rc
    a8 = torch.from_numpy(ca).view(F8).double().numpy() * sa[:, None].astype(np.float64)
    w8 = torch.from_numpy(cw).view(F8).double().numpy() * sw[:, None].astype(np.float64)
    exact = a8 @ w8.T
    scale = np.abs(a8) @ np.abs(w8).T
    pmax = (np.abs(a8)[:, None, :] * np.abs(w8)[None, :, :]).max(-1)      # largest single product per output
    e = np.abs(out - exact)
    print(f'{case:45s} max err / sum|a||w| = {(e / scale).max():.3e}   max err / largest product = {(e / pmax).max():.3e}   '
          f'median err / largest product = {np.median(e / pmax):.3e}', flush=True)
