"""Deterministic synthetic checkpoints and crops.

No real ``.pth`` exists offline (``models/download.sh`` needs network), so every
parity fixture and every benchmark uses a seeded state dict with the reference's
schema (keys/shapes as printed from ``ViTPose(cfg).state_dict()``, see SURVEY.md
section 8a).  Pure numpy ``Generator(PCG64(seed))`` so that the build container
and the GPU box produce bit-identical tensors.

Per-tensor distributions (documented so goldens are reproducible):

=============================== =====================================
tensor                          distribution
=============================== =====================================
pos_embed                       N(0, 0.02)
patch_embed.proj.weight / bias  N(0, 0.02) / N(0, 0.02)
norm*.weight / bias             1 + N(0, 0.1) / N(0, 0.05)
attn.qkv.weight / bias          N(0, 0.04) / N(0, 0.02)
attn.proj, mlp.fc1, mlp.fc2     N(0, 0.02) weight and bias
deconv weights                  N(0, sqrt(2 / (4*Cin)))
BN weight/bias/mean/var         1+N(0,.1) / N(0,.1) / N(0,.1) / U(.5,1.5)
final_layer.weight / bias       N(0, 0.3/16) / N(0, 0.02)
=============================== =====================================
"""
from __future__ import annotations

import numpy as np

from .configs import IMG_H, IMG_W, ModelShape


# LayerNorm output of the position-code channel of the peaked checkpoints (measured once with the CPU oracle, blobs + noise
# crops; +-10 % with token and content, which is what makes the peak heights crop dependent)
_CODE_NOMINAL = {(384, 12): 18.4, (768, 12): 23.6, (1024, 24): 24.7, (1280, 32): 24.3}


# the same with ``outliers=True`` (the massive channels dominate every row's variance, so the code's LayerNorm output is much smaller;
# ViTPose-S: calibrated on the peak heights instead -- measured code 1.31 -- so that its confidences stay in 0.25 .. 1.25 as well)
_CODE_NOMINAL_OUT = {(384, 12): 1.75, (768, 12): 1.866, (1024, 24): 3.474, (1280, 32): 4.898}


def synthetic_state_dict(shape: ModelShape, seed: int = 0, peaked: bool = False, outliers: bool = False) -> "dict[str, np.ndarray]":
    """Return ``{name: float32 ndarray}`` with the reference's key names/shapes.

    ``peaked=False``: every tensor random (table above) -- heatmaps are noise-like (std ~0.3): right for throughput
    and for tensor-level error budgets, useless for coordinate parity (arg-max and DARK step are ill-conditioned).

    ``peaked=True``: the same random tensors plus a small designed signal path that makes the heatmaps look like a
    trained model's -- one Gaussian-like blob per joint (sigma 2.2-3.2 px, peak 0.3-1.2) on a low background --
    so that +-0.5 px / 1e-3 can be asserted on EVERY joint.  Construction (deterministic, numpy only):

    * ``pos_embed[1 + t, t] += A``: token t carries a one-hot position code in channel t (A = 2.5 sqrt(D L / 12)); it rides the
      residual stream through all L random blocks (LayerNorm keeps it at ~0.7-0.9 sqrt(D), modulated by the content);
    * both deconvs get ``+ bilinear 2x kernel`` on the diagonal of the 192 code channels (ConvTranspose2d(4, 2, 1) with
      [1 3 3 1]/4 (x) [1 3 3 1]/4 IS bilinear upsampling), their random part is scaled by 0.35 (rows fed by the big code
      channels additionally by 1 / code), so d2[c] ~ the tent function of token c;
    * ``final_layer.weight[k, c] = amp_k G_k(token c) / gain_c`` for c < 192 (G_k a Gaussian of sigma 0.55-0.8 tokens around a
      seeded sub-token centre, gain_c = code x the two BatchNorm scales) + a small random part: heatmap k = the bilinear
      interpolation of G_k sampled on the token grid, times the content-dependent LayerNorm gain, plus noise.

    All GEMMs, LayerNorms, the attention and the head still see full-scale random operands; only the read-out is designed.

    ``outliers=True`` (with or without ``peaked``): the activation statistics of TRAINED ViTs that seeded random tensors lack --
    "massive activations" in a few residual channels and one attention head with large logits (``_make_outliers``)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    D, L, K = shape.embed_dim, shape.depth, shape.num_keypoints
    sd: dict[str, np.ndarray] = {}

    def normal(shp, std, mean=0.0):
        return (rng.standard_normal(shp, dtype=np.float32) * np.float32(std) + np.float32(mean)).astype(np.float32)

    sd['backbone.pos_embed'] = normal((1, 193, D), 0.02)
    sd['backbone.patch_embed.proj.weight'] = normal((D, 3, 16, 16), 0.02)
    sd['backbone.patch_embed.proj.bias'] = normal((D,), 0.02)
    for i in range(L):
        p = f'backbone.blocks.{i}.'
        sd[p + 'norm1.weight'] = normal((D,), 0.1, 1.0)
        sd[p + 'norm1.bias'] = normal((D,), 0.05)
        sd[p + 'attn.qkv.weight'] = normal((3 * D, D), 0.04)
        sd[p + 'attn.qkv.bias'] = normal((3 * D,), 0.02)
        sd[p + 'attn.proj.weight'] = normal((D, D), 0.02)
        sd[p + 'attn.proj.bias'] = normal((D,), 0.02)
        sd[p + 'norm2.weight'] = normal((D,), 0.1, 1.0)
        sd[p + 'norm2.bias'] = normal((D,), 0.05)
        sd[p + 'mlp.fc1.weight'] = normal((4 * D, D), 0.02)
        sd[p + 'mlp.fc1.bias'] = normal((4 * D,), 0.02)
        sd[p + 'mlp.fc2.weight'] = normal((D, 4 * D), 0.02)
        sd[p + 'mlp.fc2.bias'] = normal((D,), 0.02)
    sd['backbone.last_norm.weight'] = normal((D,), 0.1, 1.0)
    sd['backbone.last_norm.bias'] = normal((D,), 0.05)
    cin = D
    for j, idx in enumerate((0, 3)):
        h = 'keypoint_head.deconv_layers.'
        sd[f'{h}{idx}.weight'] = normal((cin, 256, 4, 4), float(np.sqrt(2.0 / (4 * cin))))
        sd[f'{h}{idx + 1}.weight'] = normal((256,), 0.1, 1.0)
        sd[f'{h}{idx + 1}.bias'] = normal((256,), 0.1)
        sd[f'{h}{idx + 1}.running_mean'] = normal((256,), 0.1)
        sd[f'{h}{idx + 1}.running_var'] = rng.uniform(0.5, 1.5, size=(256,)).astype(np.float32)
        sd[f'{h}{idx + 1}.num_batches_tracked'] = np.array(0, dtype=np.int64)
        cin = 256
    sd['keypoint_head.final_layer.weight'] = normal((K, 256, 1, 1), 0.3 / 16.0)
    sd['keypoint_head.final_layer.bias'] = normal((K,), 0.02)
    if outliers:
        _make_outliers(sd, shape, seed)
    if peaked:
        _make_peaked(sd, shape, seed, outliers)
    return sd


OUTLIER_BLOCK = 3          # the massive channels appear in the residual stream at the end of this block and stay
OUTLIER_CONST = (800.0, -600.0)   # two channels with a token-independent offset (mlp.fc2 bias)
OUTLIER_ROW_GAIN = 3.35e5  # / D: two channels driven by the token content (mlp.fc2 weight rows x gain: std ~150, |x| up to ~600)
LOGIT_TARGET = 45.0        # q and k rows of head 0 of two blocks scaled so that block 1's logits reach this magnitude (>= 30)


def outlier_channels(D: int):
    """Residual channels made massive (outside the 192 position-code channels of the peaked read-out)."""
    return [D - 5, D - 17, D - 40, D - 77]


def _make_outliers(sd, shape: ModelShape, seed: int) -> None:
    """Trained-ViT-like activation outliers on top of the random tensors (typical residual scale here: 0.3-1):

    * block ``OUTLIER_BLOCK``'s mlp.fc2 writes two channels with a constant +800 / -600 (bias) and two channels whose fc2 weight
      rows are scaled up (token-dependent values of std ~150, maxima ~600): 100-1000 x the rms of the other channels (0.5-6,
      measured with the oracle), the regime in which fp16 hi-plane operands, the ``rstd (acc - mean s)`` fold and the row
      statistics have to stay exact;
    * head 0 of blocks 1 and ``OUTLIER_BLOCK + 2`` has its q and k projections (weight rows and biases) scaled by the same
      factor each, chosen so that block 1's logits reach magnitudes of ~45 (>= 30; spread over the keys ~8): a close-to-one-hot
      softmax for the exp2-softmax and its 16-bit probabilities (the later block, fed by outlier-dominated LayerNorm outputs,
      reaches 8-20)."""
    D, L, h = shape.embed_dim, shape.depth, shape.num_heads
    hd = D // h
    lo = min(OUTLIER_BLOCK, L - 1)
    c = outlier_channels(D)
    p = f'backbone.blocks.{lo}.'
    amp = min(1.0, float(np.sqrt(D / 768.0)))        # ViTPose-S (D = 384): x 0.71, the same share of a row's variance as in -B
    sd[p + 'mlp.fc2.bias'][c[0]] += np.float32(OUTLIER_CONST[0] * amp)
    sd[p + 'mlp.fc2.bias'][c[1]] += np.float32(OUTLIER_CONST[1] * amp)
    sd[p + 'mlp.fc2.weight'][c[2]] *= np.float32(OUTLIER_ROW_GAIN / D * amp)
    sd[p + 'mlp.fc2.weight'][c[3]] *= np.float32(OUTLIER_ROW_GAIN / D * amp)
    # trained models keep such channels out of the read-out with a small last_norm gain; here it keeps the four massive values
    # (+-20-27 after LayerNorm) from drowning the position code in the deconv head (the blocks' norm1 / norm2 keep gain ~1: the
    # qkv / fc1 GEMMs DO see them)
    sd['backbone.last_norm.weight'][c] = np.float32(0.0)
    sd['backbone.last_norm.bias'][c] = np.float32(0.0)
    gain = np.float32(np.sqrt(LOGIT_TARGET / (0.009 * D)))   # un-scaled logits of block 1 reach ~0.009 D with these random tensors
    for blk in sorted({1, min(lo + 2, L - 1)}):
        p = f'backbone.blocks.{blk}.'
        for part in (0, 1):                          # q rows, k rows of head 0 (vit.py:166-167: rows [q | k | v] x head x hd)
            r = slice(part * D, part * D + hd)
            sd[p + 'attn.qkv.weight'][r] *= gain
            sd[p + 'attn.qkv.bias'][r] *= gain


def _make_peaked(sd, shape: ModelShape, seed: int, outliers: bool = False) -> None:
    D, L, K = shape.embed_dim, shape.depth, shape.num_keypoints
    rng = np.random.Generator(np.random.PCG64(seed + 7777))
    code = (_CODE_NOMINAL_OUT if outliers else _CODE_NOMINAL).get((D, L), 0.8 * float(np.sqrt(D)))
    pos = sd['backbone.pos_embed']
    a = np.float32(2.5 * np.sqrt(D) * np.sqrt(L / 12.0))
    for t in range(192):
        pos[0, 1 + t, t] += a
    bil = np.array([0.25, 0.75, 0.75, 0.25], dtype=np.float32)
    k2 = np.outer(bil, bil).astype(np.float32)
    h = 'keypoint_head.deconv_layers.'
    for idx in (0, 3):
        w = sd[f'{h}{idx}.weight'] * np.float32(0.35)
        w[:192] /= np.float32(code)
        for c in range(192):
            w[c, c] += k2
        sd[f'{h}{idx}.weight'] = np.ascontiguousarray(w, dtype=np.float32)
    s1 = sd[h + '1.weight'] / np.sqrt(sd[h + '1.running_var'] + np.float32(1e-5))
    s2 = sd[h + '4.weight'] / np.sqrt(sd[h + '4.running_var'] + np.float32(1e-5))
    gain = (code * s1[:192].astype(np.float64) * s2[:192].astype(np.float64)) * 0.67   # 0.67: peak of the interpolated blob / amp
    ty, tx = np.mgrid[0:16, 0:12].astype(np.float64)
    w = (rng.standard_normal((K, 256)) * 0.02).astype(np.float32)
    w[:, :192] /= np.float32(code)
    for k in range(K):
        cy, cx = rng.uniform(1.0, 14.0), rng.uniform(1.0, 10.0)
        sg, amp = rng.uniform(0.55, 0.8), rng.uniform(0.35, 0.95)
        g = amp * np.exp(-((ty - cy) ** 2 + (tx - cx) ** 2) / (2.0 * sg * sg))
        w[k, :192] += (g.reshape(-1) / gain).astype(np.float32)
    sd['keypoint_head.final_layer.weight'] = np.ascontiguousarray(w.reshape(K, 256, 1, 1))


def synthetic_crops(n: int, seed: int = 0, kind: str = 'noise') -> np.ndarray:
    """uint8 RGB crops ``[n, 256, 192, 3]``.

    ``noise``: uniform [0,255] (the benchmark input).
    ``blobs``: dark background with a few bright gaussian blobs -- gives the
    random-weight network spatially structured features, i.e. larger argmax margins.
    """
    rng = np.random.default_rng(seed)
    if kind == 'noise':
        return rng.integers(0, 256, size=(n, IMG_H, IMG_W, 3), dtype=np.uint8)
    if kind == 'blobs':
        yy, xx = np.mgrid[0:IMG_H, 0:IMG_W].astype(np.float32)
        out = np.empty((n, IMG_H, IMG_W, 3), dtype=np.uint8)
        for i in range(n):
            img = rng.uniform(20, 60, size=(IMG_H, IMG_W, 3)).astype(np.float32)
            for _ in range(int(rng.integers(2, 6))):
                cy, cx = rng.uniform(16, IMG_H - 16), rng.uniform(16, IMG_W - 16)
                s = rng.uniform(6, 20)
                col = rng.uniform(80, 195, size=3).astype(np.float32)
                g = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
                img += g[..., None] * col
            out[i] = np.clip(img, 0, 255).astype(np.uint8)
        return out
    raise ValueError(kind)
