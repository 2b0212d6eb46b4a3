import os, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from easy_vitpose_amd import _capi as capi
from helpers import round_to
lib = capi.load_library()
npairs, D, heads = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 768, 12
M = npairs * 384
rng = np.random.default_rng(1)
x = round_to(rng.standard_normal((M, D)).astype(np.float32), 'fp16')
W = round_to((rng.standard_normal((3 * D, D)) * 0.04).astype(np.float32), 'fp16')
b = (rng.standard_normal(3 * D) * 0.1).astype(np.float32)
out = np.empty((M, D), np.float32)
capi.check(lib.vp_dbg_qkvattn(0, 0, npairs, D, heads, x.ctypes.data, W.ctypes.data, b.ctypes.data, out.ctypes.data))
qkv = np.empty((M, 3 * D), np.float32)
capi.check(lib.vp_dbg_gemm(0, 0, 0, M, 3 * D, D, x.ctypes.data, W.ctypes.data, b.ctypes.data, None, qkv.ctypes.data))
ref = np.empty((M, D), np.float32)
capi.check(lib.vp_dbg_attention(0, 0, M // 192, D, heads, qkv.ctypes.data, ref.ctypes.data))
d = out != ref
print('elements differing:', int(d.sum()), 'of', d.size, ' max abs diff', float(np.abs(out - ref).max()), 'ref scale', float(np.abs(ref).max()))
if d.any():
    rows, cols = np.nonzero(d)
    print('rows (mod 192) histogram of differing rows:', np.bincount(rows % 192, minlength=192).tolist())
    print('heads:', np.bincount(cols // 64, minlength=heads).tolist(), ' d within head:', np.bincount(cols % 64, minlength=64).tolist())
    print('crops:', np.bincount(rows // 192, minlength=M // 192).tolist())
# fp64 reference
import torch
t = torch.from_numpy(qkv).double().reshape(M // 192, 192, 3, heads, 64).permute(2, 0, 3, 1, 4)
r64 = (((t[0] * 0.125) @ t[1].transpose(-2, -1)).softmax(-1) @ t[2]).transpose(1, 2).reshape(M, D).numpy()
print('vs fp64: fused max err', float(np.abs(out - r64).max()), ' unfused max err', float(np.abs(ref - r64).max()))
if d.any():
    pairs_ = sorted({(int(r // 192), int(c // 64)) for r, c in zip(rows, cols)})
    info = []
    for crop, h in pairs_:
        tile = (crop // 2) * heads + h
        nt = npairs * heads
        q8, r8 = nt >> 3, nt & 7
        # which xcd range contains tile
        for xcd in range(8):
            base = xcd * (q8 + 1) if xcd < r8 else r8 * (q8 + 1) + (xcd - r8) * q8
            cnt = q8 + (1 if xcd < r8 else 0)
            if base <= tile < base + cnt:
                loc = tile - base
                info.append((crop, crop & 1, h, tile, xcd, loc, loc // 32))
    print('(crop, A/B, head, tile, xcd, local index, round):', info)
    qrows = sorted({(int(r // 192), int(r % 192), int(c // 64)) for r, c in zip(rows, cols)})
    print('(crop, query row, head) with differences:', qrows[:40])
