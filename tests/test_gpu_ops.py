"""GPU: each HIP kernel alone (through the C ABI's parity taps) against a plain
PyTorch fp32 reference of the same op on the same (dtype-rounded) operands."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from easy_vitpose_amd import _capi as capi
from helpers import round_to

pytestmark = pytest.mark.gpu
DT = {'fp16': capi.VP_DTYPE_F16, 'bf16': capi.VP_DTYPE_BF16}
# one rounding of the fp32 result to the storage type
OUT_EPS = {'fp16': 2.0 ** -11, 'bf16': 2.0 ** -8}


def _ptr(a):
    return None if a is None else a.ctypes.data


def _gemm(dtype, epi, A, W, bias, aux):
    lib = capi.load_library()
    M, K = A.shape
    N = W.shape[0]
    out = np.empty((M, N), np.float32)
    capi.check(lib.vp_dbg_gemm(0, DT[dtype], epi, M, N, K, _ptr(A), _ptr(W), _ptr(bias), _ptr(aux), _ptr(out)))
    return out


@pytest.mark.parametrize('dtype', ['fp16', 'bf16'])
@pytest.mark.parametrize('epi', [0, 1, 2, 3])
@pytest.mark.parametrize('M,N,K', [(192, 384, 384), (1000, 256, 64), (384, 2304, 768), (2 * 192 + 7, 132, 256)])
def test_gemm_epilogues(dtype, epi, M, N, K):
    rng = np.random.default_rng(M * 7 + N * 3 + K + epi)
    A = round_to(rng.standard_normal((M, K), dtype=np.float32), dtype)
    W = round_to(rng.standard_normal((N, K), dtype=np.float32) * (K ** -0.5), dtype)
    # asymmetric structure so that an m<->n transpose or a k-permutation cannot cancel
    A[:, 0] += np.arange(M, dtype=np.float32) % 5
    W[:, 1] += (np.arange(N, dtype=np.float32) % 3) * 0.25
    A, W = round_to(A, dtype), round_to(W, dtype)
    bias = rng.standard_normal(N, dtype=np.float32)
    aux = rng.standard_normal((M if epi == 2 else 192, N), dtype=np.float32) if epi >= 2 else None
    got = _gemm(dtype, epi, A, W, bias, aux)
    ref = torch.from_numpy(A).double() @ torch.from_numpy(W).double().T
    if epi != 3:
        ref = ref + torch.from_numpy(bias).double()
    if epi == 1:
        ref = F.gelu(ref)
    if epi == 2:
        ref = ref + torch.from_numpy(aux).double()
    if epi == 3:
        ref = ref + torch.from_numpy(aux).double()[torch.arange(M) % 192]
    ref = ref.numpy()
    scale = np.abs(ref).max()
    tol = 2e-5 * scale + (OUT_EPS[dtype] * scale if epi < 2 else 0)   # fp32 accumulate (+ one output rounding)
    err = np.abs(got - ref).max()
    assert err <= tol, f'gemm epi={epi} {M}x{N}x{K} {dtype}: max err {err:.3e} > {tol:.3e}'


@pytest.mark.parametrize('dtype', ['fp16', 'bf16'])
@pytest.mark.parametrize('D,heads', [(384, 12), (768, 12), (1280, 16)])
def test_attention(dtype, D, heads):
    B, hd = 3, D // heads
    rng = np.random.default_rng(D + heads)
    qkv = rng.standard_normal((B * 192, 3 * D), dtype=np.float32)
    qkv[:, :D] *= 1.5        # non-uniform softmax
    qkv[5, :D] *= 4.0        # one spiky query row
    qkv = round_to(qkv, dtype)
    out = np.empty((B * 192, D), np.float32)
    lib = capi.load_library()
    capi.check(lib.vp_dbg_attention(0, DT[dtype], B, D, heads, _ptr(qkv), _ptr(out)))
    t = torch.from_numpy(qkv).double().reshape(B, 192, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    ref = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(-1) @ v
    ref = ref.transpose(1, 2).reshape(B * 192, D).numpy()
    # P is rounded to the operand type before the PV product, the output once more
    tol = 3 * OUT_EPS[dtype] * np.abs(ref).max()
    err = np.abs(out - ref).max()
    assert err <= tol, f'attention D={D} {dtype}: max err {err:.3e} > {tol:.3e}'


@pytest.mark.parametrize('dtype', ['fp16', 'bf16'])
@pytest.mark.parametrize('M,D', [(7, 384), (192, 768), (33, 1024), (5, 1280)])
def test_layernorm(dtype, M, D):
    rng = np.random.default_rng(M + D)
    x = (rng.standard_normal((M, D), dtype=np.float32) * 3 + 1.5).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    b = (0.05 * rng.standard_normal(D)).astype(np.float32)
    o16, o32 = np.empty_like(x), np.empty_like(x)
    lib = capi.load_library()
    capi.check(lib.vp_dbg_layernorm(0, DT[dtype], M, D, _ptr(x), _ptr(g), _ptr(b), _ptr(o16), _ptr(o32)))
    ref = F.layer_norm(torch.from_numpy(x), (D,), torch.from_numpy(g), torch.from_numpy(b), eps=1e-6).numpy()
    assert np.abs(o32 - ref).max() < 2e-5
    assert np.abs(o16 - ref).max() <= OUT_EPS[dtype] * np.abs(ref).max() + 2e-5


@pytest.mark.parametrize('dtype', ['fp16', 'bf16'])
@pytest.mark.parametrize('B,Hin,Win,Cin', [(2, 16, 12, 384), (1, 32, 24, 256), (3, 16, 12, 768)])
def test_deconv_bn_relu(dtype, B, Hin, Win, Cin):
    rng = np.random.default_rng(B + Hin + Cin)
    x = round_to(rng.standard_normal((B, Hin, Win, Cin), dtype=np.float32), dtype)      # NHWC
    w = (rng.standard_normal((Cin, 256, 4, 4), dtype=np.float32) * (2.0 / (4 * Cin)) ** 0.5).astype(np.float32)
    bn = dict(weight=(1 + 0.1 * rng.standard_normal(256)).astype(np.float32), bias=(0.1 * rng.standard_normal(256)).astype(np.float32),
              running_mean=(0.1 * rng.standard_normal(256)).astype(np.float32), running_var=rng.uniform(0.5, 1.5, 256).astype(np.float32))
    tensors = {'keypoint_head.deconv_layers.0.weight': w}
    tensors.update({f'keypoint_head.deconv_layers.1.{k}': v for k, v in bn.items()})
    keep = {k: np.ascontiguousarray(v) for k, v in tensors.items()}
    descs = (capi.vp_tensor_desc * len(keep))(*[capi.vp_tensor_desc(k.encode(), v.ctypes.data_as(C.POINTER(C.c_float)), v.size)
                                                 for k, v in keep.items()])
    out = np.empty((B, 2 * Hin, 2 * Win, 256), np.float32)
    lib = capi.load_library()
    capi.check(lib.vp_dbg_deconv(0, DT[dtype], B, Hin, Win, Cin, _ptr(x), descs, len(keep), _ptr(out)))
    # reference with the same folding the packer does (fold, THEN round the weights)
    sc = bn['weight'] / np.sqrt(bn['running_var'] + 1e-5)
    wf = round_to(w * sc[None, :, None, None], dtype)
    bf = bn['bias'] - bn['running_mean'] * sc
    ref = F.relu(F.conv_transpose2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(), torch.from_numpy(wf).double(),
                                    torch.from_numpy(bf).double(), stride=2, padding=1)).permute(0, 2, 3, 1).numpy()
    tol = (OUT_EPS[dtype] + 2e-5) * np.abs(ref).max()
    err = np.abs(out - ref).max()
    assert err <= tol, f'deconv {dtype}: max err {err:.3e} > {tol:.3e}'
    # and against the un-folded fp32 module semantics (ConvTranspose2d -> BatchNorm2d(eval) -> ReLU)
    y = F.conv_transpose2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w), None, stride=2, padding=1)
    y = F.relu(F.batch_norm(y, torch.from_numpy(bn['running_mean']), torch.from_numpy(bn['running_var']),
                            torch.from_numpy(bn['weight']), torch.from_numpy(bn['bias']), training=False, eps=1e-5))
    err2 = np.abs(out - y.permute(0, 2, 3, 1).numpy()).max()
    assert err2 <= 4 * OUT_EPS[dtype] * np.abs(ref).max() + 1e-4, f'deconv vs module: {err2:.3e}'
