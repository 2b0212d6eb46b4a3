"""GPU: the OPT-IN fp8 mode (vp_config.dtype = VP_DTYPE_FP8, BASELINE configs[4]: ViTPose-B / AP-10K, fp8 operands on the CDNA4 fp8 MFMA).

qkv / fc1 / fc2 run on MXFP8 operands (OCP e4m3 codes + one power-of-two scale per 32 k, csrc/mx8.h) through the block-scaled
v_mfma_scale_f32_16x16x128_f8f6f4; everything else is the fp16 path.  e4m3 carries 3 mantissa bits: THIS MODE DOES NOT MEET THE
NORTH_STAR'S 1e-3 ON CONFIDENCES, it is never a default and never counts toward parity.  What is asserted here:

* the GEMM kernel itself is EXACT for what it is given: against fp64 arithmetic on the very codes and scales it multiplied (operands
  returned de-quantised by the tap), for all three epilogues incl. the MXFP8 output of fc1;
* end to end on the peaked AP-10K checkpoint: coordinates within the north_star's +-0.5 px on EVERY joint, confidence error measured
  and asserted at its real bound (several 1e-3 -- printed);
* mode properties at BASELINE configs[4]'s batch (512): finite, run-to-run identical, crop i of 512 == crop i alone.
"""
import numpy as np
import pytest
import torch

from easy_vitpose_amd import VitPoseHip
from easy_vitpose_amd import _capi as capi
from easy_vitpose_amd.configs import model_shape
from easy_vitpose_amd.synth import synthetic_crops, synthetic_state_dict
from helpers import CONF_TOL, KP_TOL_PX, oracle_heatmaps, weights
from oracle import vitpose_cpu as O

pytestmark = pytest.mark.gpu
F8 = torch.float8_e4m3fn

# measured bounds of the mode (round 4, MI355X), asserted with ~1.5 x headroom; the north_star's own are KP_TOL_PX / CONF_TOL
FP8_CONF_ERR_PEAKED = 5e-3         # peaked AP-10K checkpoint, every joint: measured 3.08e-3 (rms 1.0e-3, 68 % of the joints within 1e-3; with attn.proj on fp16: 2.79e-3); fp16 path 6.0e-4
FP8_HM_RMS_NOISE = 3.5e-2          # random-weight (noise-like) heatmaps, std 0.345: measured rms 2.18e-2, max 1.1e-1 (confidence max err 7.0e-2)


def _gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def _case(epi, A, W, bias, aux=None):
    lib = capi.load_library()
    M, K = A.shape
    N = W.shape[0]
    out = np.empty((M, N), np.float32)
    stats = np.empty((M, N // 64, 2), np.float32) if epi == 6 else None
    a_deq = np.empty((M, K), np.float32)
    w_deq = np.empty((N, K), np.float32)
    keep = [np.ascontiguousarray(a, dtype=np.float32) if a is not None else None for a in (A, W, bias, aux)]
    rc = lib.vp_dbg_gemm_fp8_case(0, epi, M, N, K, *[None if a is None else a.ctypes.data for a in keep], out.ctypes.data,
                                  None if stats is None else stats.ctypes.data, a_deq.ctypes.data, w_deq.ctypes.data)
    assert rc == 0, capi.last_error()
    return out, stats, a_deq, w_deq


def _operands(M, N, K, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((M, K)).astype(np.float32)
    A *= np.repeat(np.exp2(rng.integers(-3, 4, size=(M, K // 32))).astype(np.float32), 32, axis=1)     # blocks in different binades
    A[:, 3] += (np.arange(M) % 7).astype(np.float32)                                                   # asymmetric in m
    A[5, 40] = 300.0                                                                                   # an outlier inside a block
    W = (rng.standard_normal((N, K)) * 0.04).astype(np.float32)
    W[:, 1] += 0.1 * (np.arange(N) % 5)                                                                # asymmetric in n
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    return rng, A, W, bias


@pytest.mark.parametrize('M,N,K', [(512, 2304, 768), (1536, 2304, 768), (256, 3072, 1024), (2048, 3840, 1280)])
def test_fp8_gemm_qkv_epilogue(M, N, K):
    """acc * w_scale + bias -> fp16, straight from registers (split column layout): exact product of the codes, one output rounding."""
    _, A, W, bias = _operands(M, N, K, M + N)
    out, _, a_deq, w_deq = _case(0, A, W, bias)
    ref = (torch.from_numpy(a_deq).double() @ torch.from_numpy(w_deq).double().T).numpy() + bias
    mag = (torch.from_numpy(np.abs(a_deq)).double() @ torch.from_numpy(np.abs(w_deq)).double().T).numpy()
    err = np.abs(out - ref)
    tol = 2.0 ** -10 * np.abs(ref) + 1e-4 * mag + 1e-4          # fp16 rounding of the result + the instruction's internal alignment (probe: < 5e-5 of sum|a||w|)
    assert (err <= tol).all(), f'qkv {M}x{N}x{K}: max err {err.max():.3e}, worst ratio {(err / tol).max():.2f}'
    # and the operand quantisation is what the format promises: relative to the full-precision product
    full = (torch.from_numpy(A).double() @ torch.from_numpy(W).double().T).numpy() + bias
    q = np.sqrt(((out - full) ** 2).mean()) / full.std()
    print(f'[fp8 gemm qkv {M}x{N}x{K}] kernel vs fp64 of its operands: max err ratio {(err / tol).max():.2f}; MXFP8 operand error {q:.3e} of the output scale')
    assert q < 6e-2


def _mx_qdq(x):
    """quantise-dequantise rows of x [M, N] to MXFP8 exactly as csrc/mx8.h does (block of 32 columns, amax -> [128, 256))"""
    M, N = x.shape
    xb = x.astype(np.float32).reshape(M, N // 32, 32)
    amax = np.abs(xb).max(-1)
    ex = (amax.view(np.uint32) >> 23) & 0xff
    E = np.where(ex > 7, ex - 7, 0).astype(np.int32)
    inv = np.exp2(127.0 - E).astype(np.float32)[..., None]
    codes = torch.from_numpy((xb * inv).reshape(M, N)).to(F8)
    return (codes.float().numpy().reshape(M, N // 32, 32) * np.exp2(E - 127.0).astype(np.float32)[..., None]).reshape(M, N), np.exp2(E - 127.0)


@pytest.mark.parametrize('M,N,K', [(512, 3072, 768), (1280, 3072, 768), (256, 4096, 1024)])
def test_fp8_gemm_fc1_epilogue_writes_mxfp8(M, N, K):
    """gelu(acc * w_scale + bias) quantised to MXFP8 by the epilogue (a 32-column block = two lanes): the returned values must be the
    quantise-dequantise of the fp64 reference -- identical except where the fp32 value sits within rounding of a code boundary."""
    _, A, W, bias = _operands(M, N, K, M * 3 + N)
    out, _, a_deq, w_deq = _case(1, A, W, bias)
    ref = _gelu((torch.from_numpy(a_deq).double() @ torch.from_numpy(w_deq).double().T).numpy() + bias)
    ref_q, blk = _mx_qdq(ref.astype(np.float32))
    step = np.repeat(blk, 32, axis=1) * 16.0                    # largest code spacing inside a block (values in [128, 256) are 16 apart)
    diff = np.abs(out - ref_q)
    frac = (diff > 0).mean()
    print(f'[fp8 gemm fc1 {M}x{N}x{K}] MXFP8 output: {frac:.2e} of the elements differ from quantise(fp64 reference), max difference {(diff / step).max():.2f} block steps')
    assert frac < 5e-3 and (diff <= step * 1.001).all()
    assert np.abs(out - ref).max() <= (np.abs(ref) * 2.0 ** -4 + np.repeat(blk, 32, axis=1) * 2.0 ** -2).max() * 1.01      # never worse than the format


@pytest.mark.parametrize('M,N,K', [(512, 768, 3072), (1280, 768, 3072), (512, 1024, 4096), (512, 1280, 5120), (512, 768, 768), (1024, 768, 768)])
def test_fp8_gemm_residual_epilogue(M, N, K):
    """acc * w_scale + bias + two-plane residual, row statistics: 256 x 192 tiles through LDS (N = 768) and 256 x 256 tiles straight
    from registers (N = 1024 / 1280), against fp64 of the same operands."""
    rng, A, W, bias = _operands(M, N, K, M + 2 * N)
    A *= 0.5
    resid = (rng.standard_normal((M, N)) * 2.0).astype(np.float32)
    out, st, a_deq, w_deq = _case(6, A, W, bias, aux=resid)
    hi = torch.from_numpy(resid).half().float().numpy()
    x0 = hi.astype(np.float64) + torch.from_numpy(resid - hi).half().double().numpy()
    ref = (torch.from_numpy(a_deq).double() @ torch.from_numpy(w_deq).double().T).numpy() + bias + x0
    mag = (torch.from_numpy(np.abs(a_deq)).double() @ torch.from_numpy(np.abs(w_deq)).double().T).numpy()
    err = np.abs(out - ref)
    tol = 2e-5 * np.maximum(1.0, np.abs(ref)) + 3e-4 * mag     # the instruction aligns its products to the largest one (row 5 carries a 300 among O(1) values)
    assert (err <= tol).all(), f'fc2 {M}x{N}x{K}: max err {err.max():.3e}, worst ratio {(err / tol).max():.2f}'
    g = out.astype(np.float64).reshape(M, N // 64, 64)
    assert np.abs(st[..., 0] - g.sum(-1)).max() < 2e-3 * max(1.0, np.abs(g.sum(-1)).max() / 64)
    m2 = ((g - g.mean(-1, keepdims=True)) ** 2).sum(-1)
    assert np.abs(st[..., 1] - m2).max() < 2e-3 * max(1.0, m2.max())
    print(f'[fp8 gemm fc2 {M}x{N}x{K}] kernel vs fp64 of its operands: worst ratio {(err / tol).max():.2f}')


@pytest.mark.parametrize('proj16', ['0', '1'])
def test_fp8_mode_end_to_end_peaked_ap10k(monkeypatch, proj16):
    """BASELINE configs[4]'s model (ViTPose-B / AP-10K) with the peaked checkpoint: coordinates of EVERY joint within the north_star's
    +-0.5 px of the fp32 oracle; the confidence error is measured and asserted at its real bound -- it does NOT meet the north_star's 1e-3.
    proj16 = 1 (VP_FP8_PROJ16): attn.proj on the fp16 kernels (the attention core writes fp16); default: the attention core writes
    MXFP8 and attn.proj is the fourth GEMM on the fp8 kernel."""
    from cases import peaked_crops
    if proj16 == '1':
        monkeypatch.setenv('VP_FP8_PROJ16', '1')
    shp = model_shape('b', 'ap10k')
    sd = synthetic_state_dict(shp, 0, peaked=True)
    crops = peaked_crops(8)
    eng = VitPoseHip(shp, sd, dtype='fp8', device_id=0, max_batch=8)
    kp = eng.infer(crops)
    kernels = {f: eng.profile_kernel(f) for f in ('gemm_qkv', 'gemm_fc1', 'gemm_fc2', 'gemm_proj')}
    assert all('gemm8f_kernel' in kernels[f] for f in ('gemm_qkv', 'gemm_fc1', 'gemm_fc2')), kernels
    assert ('gemm8f_kernel' in kernels['gemm_proj']) == (proj16 == '0'), kernels
    assert np.array_equal(eng.infer(crops), kp)                                           # run-to-run
    assert np.array_equal(np.concatenate([eng.infer(crops[i:i + 1]) for i in (0, 3, 7)]), kp[[0, 3, 7]])   # crop i of the batch == crop i alone
    eng.close()
    ref16 = VitPoseHip(shp, sd, dtype='fp16', device_id=0, max_batch=8)
    kp16 = ref16.infer(crops)
    ref16.close()
    sdt = O.to_torch_state_dict(sd)
    ref = np.concatenate([O.inference_torch(sdt, shp.depth, shp.num_heads, c) for c in crops])
    dpx = np.abs(kp[..., :2] - ref[..., :2]).max(-1)
    dcf = np.abs(kp[..., 2] - ref[..., 2])
    d16 = np.abs(kp16[..., 2] - ref[..., 2])
    print(f'[fp8 mode, b/ap10k peaked, proj16={proj16}] {dpx.size} joints: coordinate max err {dpx.max():.4f} px (mean {dpx.mean():.4f}); confidence max err {dcf.max():.3e} '
          f'rms {np.sqrt((dcf ** 2).mean()):.3e}, {(dcf < CONF_TOL).mean():.3f} of the joints within 1e-3  [fp16 path: max {d16.max():.3e}]   kernels: {kernels}')
    assert np.isfinite(kp).all()
    assert dpx.max() < KP_TOL_PX, 'fp8 mode: coordinates must stay inside the north_star tolerance on peaked maps'
    assert dcf.max() < FP8_CONF_ERR_PEAKED, 'fp8 mode: confidence error above the bound measured in round 4'
    assert d16.max() < CONF_TOL


@pytest.mark.parametrize('variant,dataset,conf_bound', [('l', 'coco_25', 6e-3), ('h', 'wholebody', 8e-3)])
def test_fp8_mode_large_models_vs_reference_golden(golden_dir, variant, dataset, conf_bound):
    """The mode on the other models it accepts, against the reference's own peaked-checkpoint keypoints: ViTPose-L (D = 1024: attn.proj / fc2 on
    256 x 256 tiles with the register-direct residual epilogue, head dim 64 -> MXFP8 attention output) and ViTPose-H (D = 1280, head dim 80:
    attn.proj stays on fp16, K = 133).  Coordinates inside the north_star's +-0.5 px on every joint; confidences at the mode's measured bound."""
    import os
    from cases import peaked_crops
    z = np.load(os.path.join(golden_dir, f'peaked_{variant}_{dataset}.npz'))
    n = int(z['n'])
    shp = model_shape(variant, dataset)
    eng = VitPoseHip(shp, synthetic_state_dict(shp, 0, peaked=True), dtype='fp8', device_id=0, max_batch=n)
    kp = eng.infer(peaked_crops(n))
    kernels = {f: eng.profile_kernel(f) for f in ('gemm_qkv', 'gemm_fc1', 'gemm_fc2', 'gemm_proj')}
    eng.close()
    ref = z['keypoints']
    dpx = np.abs(kp[..., :2] - ref[..., :2]).max(-1)
    dcf = np.abs(kp[..., 2] - ref[..., 2])
    print(f'[fp8 mode, {variant}/{dataset} peaked vs reference golden] {dpx.size} joints: coordinate max err {dpx.max():.4f} px; confidence max err {dcf.max():.3e} '
          f'rms {np.sqrt((dcf ** 2).mean()):.3e}, {(dcf < CONF_TOL).mean():.3f} within 1e-3; kernels {kernels}')
    assert np.isfinite(kp).all() and all('gemm8f_kernel' in kernels[f] for f in ('gemm_qkv', 'gemm_fc1', 'gemm_fc2'))
    assert dpx.max() < KP_TOL_PX
    assert dcf.max() < conf_bound


def test_fp8_mode_config5_batch512():
    """BASELINE configs[4] as written: ViTPose-B / AP-10K, batch 512, fp8 operands.  Full-batch properties + the heatmap error of the
    mode on the bench's random-weight checkpoint against the fp32 oracle (noise-like maps: the worst case for 3-bit operands)."""
    shp, sd, _ = weights('b', 'ap10k')
    crops = synthetic_crops(512, 5, 'noise')
    crops[:16] = synthetic_crops(16, 6, 'blobs')
    eng = VitPoseHip(shp, sd, dtype='fp8', device_id=0, max_batch=512)
    out = eng.infer(crops)
    assert out.shape == (512, 17, 3) and np.isfinite(out).all()
    assert np.array_equal(eng.infer(crops), out)
    idx = [0, 7, 255, 256, 300, 511]
    assert np.array_equal(np.concatenate([eng.infer(crops[i:i + 1]) for i in idx]), out[idx])
    hm = eng.heatmaps(crops[idx])
    eng.close()
    ref_hm = oracle_heatmaps('b', 'ap10k', crops[idx])
    err = hm - ref_hm
    rms = float(np.sqrt((err ** 2).mean()))
    cerr = np.abs(out[idx][..., 2] - O.decode_per_crop(ref_hm)[..., 2])
    print(f'[fp8 mode, b/ap10k @512, random weights] heatmap std {ref_hm.std():.3f}: error rms {rms:.3e} max {np.abs(err).max():.3e}; confidence max err {cerr.max():.3e}')
    assert rms < FP8_HM_RMS_NOISE


def test_fp8_mode_config5_full_batch_against_the_reference(golden_dir):
    """BASELINE configs[4] as written (ViTPose-B / AP-10K, 512 crops in one call, fp8 operands), peaked checkpoint, EVERY crop and joint against the
    keypoints the reference produced crop by crop (tests/golden/full_b_ap10k_512.npz): coordinates inside the north_star's +-0.5 px, confidences
    at the mode's own measured bound (NOT the north_star's 1e-3)."""
    import os
    from cases import fullbatch_crops
    z = np.load(os.path.join(golden_dir, 'full_b_ap10k_512.npz'))
    shp = model_shape('b', 'ap10k')
    eng = VitPoseHip(shp, synthetic_state_dict(shp, 0, peaked=True), dtype='fp8', device_id=0, max_batch=512)
    kp = eng.infer(fullbatch_crops(512))
    eng.close()
    ref = z['keypoints']
    dpx = np.abs(kp[..., :2] - ref[..., :2]).max(-1)
    dcf = np.abs(kp[..., 2] - ref[..., 2])
    print(f'[fp8 mode, b/ap10k x 512 vs the reference] {dpx.size} joints: coordinate max err {dpx.max():.4f} px; confidence max err {dcf.max():.3e} '
          f'rms {np.sqrt((dcf ** 2).mean()):.3e}, {(dcf < CONF_TOL).mean():.3f} within 1e-3')
    assert np.isfinite(kp).all() and dpx.max() < KP_TOL_PX
    assert dcf.max() < FP8_CONF_ERR_PEAKED


def test_fp8_mode_rejects_what_it_does_not_support():
    shp, sd, _ = weights('s', 'coco')
    with pytest.raises(Exception):
        VitPoseHip(shp, sd, dtype='fp8', device_id=0, max_batch=4)                        # embed_dim 384: three K-tiles of 128
