"""Shared helpers for the parity tests (CPU side only uses numpy/torch + oracle)."""
from __future__ import annotations

import functools

import numpy as np

from easy_vitpose_amd.configs import model_shape
from easy_vitpose_amd.synth import synthetic_crops, synthetic_state_dict
from oracle import vitpose_cpu as O

# Tolerances (north_star: keypoints +-0.5 px, confidences 1e-3 vs the torch-CPU reference).
KP_TOL_PX = 0.5
CONF_TOL = 1e-3


@functools.lru_cache(maxsize=8)
def weights(variant: str, dataset: str, seed: int = 0):
    shp = model_shape(variant, dataset)
    sd = synthetic_state_dict(shp, seed)
    return shp, sd, O.to_torch_state_dict(sd)


def oracle_heatmaps(variant, dataset, crops_u8, seed=0, chunk=8):
    shp, _, sdt = weights(variant, dataset, seed)
    outs = []
    for i in range(0, len(crops_u8), chunk):
        x = np.concatenate([O.pre_img(c)[0] for c in crops_u8[i:i + chunk]])
        outs.append(O.model_forward(sdt, x, shp.depth, shp.num_heads))
    return np.concatenate(outs)


def round_to(x: np.ndarray, dtype: str) -> np.ndarray:
    """Round fp32 values to the GEMM operand type and back (what the device stores)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    dt = torch.float16 if dtype in ('fp16', 'f16') else torch.bfloat16
    return t.to(dt).float().numpy()


def dark_offset_px(kp_yx: np.ndarray, heatmaps: np.ndarray, org_wh=None) -> np.ndarray:
    """|DARK refinement step| in heatmap pixels per joint: distance between the decoded
    sub-pixel location and the raw arg-max.  Real (Gaussian-like) peaks give < 1 px; on
    the noise-like maps of random weights the 2x2 Hessian is often near-singular and the
    Newton step explodes -- those joints are ill-conditioned in the REFERENCE itself
    (1e-7 perturbations of the blurred samples move them by tenths of a pixel), so
    coordinate parity is asserted only where this offset is small."""
    preds, _ = O.get_max_preds(heatmaps)
    n = heatmaps.shape[0]
    w = np.full(n, 192.0) if org_wh is None else np.asarray(org_wh, dtype=np.float64)[:, 0]
    h = np.full(n, 256.0) if org_wh is None else np.asarray(org_wh, dtype=np.float64)[:, 1]
    # invert transform_preds to heatmap pixels
    x_hm = (kp_yx[..., 1] - ((w // 2) - w * 0.5)[:, None]) / (w / 47.0)[:, None]
    y_hm = (kp_yx[..., 0] - ((h // 2) - h * 0.5)[:, None]) / (h / 63.0)[:, None]
    return np.hypot(x_hm - preds[..., 0], y_hm - preds[..., 1])


def argmax_margin(heatmaps: np.ndarray, radius: int = 2) -> np.ndarray:
    """max - (largest value further than `radius` px from the arg-max), per joint."""
    n, k, H, W = heatmaps.shape
    flat = heatmaps.reshape(n, k, -1)
    idx = flat.argmax(-1)
    yy, xx = np.mgrid[0:H, 0:W]
    out = np.empty((n, k), dtype=np.float64)
    for i in range(n):
        for j in range(k):
            y0, x0 = divmod(int(idx[i, j]), W)
            mask = (np.abs(yy - y0) > radius) | (np.abs(xx - x0) > radius)
            out[i, j] = flat[i, j, idx[i, j]] - heatmaps[i, j][mask].max()
    return out


def dark_conditioned(heatmaps: np.ndarray, vmin: float = 0.05) -> np.ndarray:
    """Joints whose DARK refinement is numerically well-posed in the REFERENCE:
    all 7 blurred samples >= vmin (the log amplifies an absolute heatmap error dv to
    dv/v; below the 0.001 clip the map is flat), and the log-map has a proper local
    maximum at the arg-max (negative-definite Hessian with a clear margin).  A trained
    model's peaks satisfy this; most noise-like maps of random weights do not."""
    n, k, H, W = heatmaps.shape
    preds, _ = O.get_max_preds(heatmaps)
    out = np.zeros((n, k), dtype=bool)
    for i in range(n):
        for j in range(k):
            x, y = int(preds[i, j, 0]), int(preds[i, j, 1])
            if x < 0:
                continue
            bp = np.pad(O.gaussian_blur(heatmaps[i, j]), 1, mode='edge')
            px, py = x + 1, y + 1
            s = np.array([bp[py, px], bp[py, px + 1], bp[py, px - 1], bp[py + 1, px], bp[py - 1, px],
                          bp[py + 1, px + 1], bp[py - 1, px - 1]], dtype=np.float64)
            if s.min() < vmin:
                continue
            l = np.log(s)
            dxx, dyy = l[1] - 2 * l[0] + l[2], l[3] - 2 * l[0] + l[4]
            dxy = 0.5 * (l[5] - l[1] - l[3] + 2 * l[0] - l[2] - l[4] + l[6])
            out[i, j] = dxx < -0.02 and dyy < -0.02 and dxx * dyy - dxy * dxy > 0.25 * dxx * dyy
    return out
