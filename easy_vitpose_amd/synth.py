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


def synthetic_state_dict(shape: ModelShape, seed: int = 0) -> "dict[str, np.ndarray]":
    """Return ``{name: float32 ndarray}`` with the reference's key names/shapes."""
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
    return sd


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
