#!/usr/bin/env python3
"""Which activation block scale does v_mfma_scale_f32_16x16x128_f8f6f4 apply to which k?  (development aid of the fp8 mode, round 4)

Activations: every element equals its block's amax 2^(2 kb) (code 128, scale byte by block); weights: 1 inside ONE 16-k half-block
h0 (of 8 per 128-k tile), 0 elsewhere: out = 16 * 2^(2 kb_used).  The table shows, for every half block of k, which block's scale the
hardware applied to it."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easy_vitpose_amd import _capi as capi

lib = capi.load_library()
M, N, K = 64, 16, 128
kb = np.arange(K // 32)[None, :] + 0 * np.arange(M)[:, None]
A = np.repeat(np.exp2(2 * kb).astype(np.float32), 32, axis=1)
ws = np.full(N, 1.0 / 448.0, np.float32)
for h0 in range(8):
    W = np.zeros((N, K), np.float32)
    W[:, h0 * 16:(h0 + 1) * 16] = 1.0
    out = np.empty((M, N), np.float32)
    capi.check(lib.vp_dbg_mx_gemm(0, M, N, K, A.ctypes.data, W.ctypes.data, ws.ctypes.data, out.ctypes.data, None, None, None))
    v = out / 16.0
    used = np.log2(v) / 2
    print(f'weights nonzero in k [{h0 * 16}, {h0 * 16 + 16}) (block {h0 // 2}): scale of block {np.unique(used).tolist()} applied (uniform over rows/cols: {np.allclose(v, v[0, 0])})')
# and per 4-byte group inside one lane's 32 bytes
print('--- weights nonzero in ONE k:')
for k0 in [0, 1, 4, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64, 96, 127]:
    W = np.zeros((N, K), np.float32)
    W[:, k0] = 1.0
    out = np.empty((M, N), np.float32)
    capi.check(lib.vp_dbg_mx_gemm(0, M, N, K, A.ctypes.data, W.ctypes.data, ws.ctypes.data, out.ctypes.data, None, None, None))
    print(f'  k = {k0:3d} (block {k0 // 32}): scale of block {np.unique(np.log2(out) / 2).tolist()}')
