"""Seeded input generators shared by the golden generator and the tests.

Inputs are regenerated from seeds (bit-identical numpy PCG64 streams); only the
reference's OUTPUTS are stored in the ``.npz`` fixtures next to this file.
"""
from __future__ import annotations

import numpy as np

HM_H, HM_W = 64, 48


def peaked_heatmaps(n: int, k: int, seed: int) -> np.ndarray:
    """Heatmaps ``[n, k, 64, 48]`` float32 that look like a trained model's output:
    gaussian blobs (sigma 2, amplitude 0.3..0.95) + N(0, 0.01) noise, with sub-pixel
    centres.  Deliberate edge cases:

    * joints k%6==1: centre within 0..2 px of a border / in a corner,
    * joint (crop 1, k=0) and (crop 2, k=3): all values <= 0  (coords become -1),
    * joint (crop 3, k=2): two exactly tied maxima (first index must win),
    * joint (crop 4, k=5): flat constant positive map.
    """
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:HM_H, 0:HM_W].astype(np.float32)
    hm = np.empty((n, k, HM_H, HM_W), dtype=np.float32)
    for i in range(n):
        for j in range(k):
            if j % 6 == 1:
                side = int(rng.integers(0, 6))
                cx = rng.uniform(0, 2) if side in (0, 4) else (HM_W - 1 - rng.uniform(0, 2) if side in (1, 5) else rng.uniform(0, HM_W - 1))
                cy = rng.uniform(0, 2) if side in (2, 4) else (HM_H - 1 - rng.uniform(0, 2) if side in (3, 5) else rng.uniform(0, HM_H - 1))
            else:
                cx, cy = rng.uniform(3, HM_W - 4), rng.uniform(3, HM_H - 4)
            amp = rng.uniform(0.3, 0.95)
            g = amp * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * 2.0 ** 2))
            hm[i, j] = g + rng.normal(0, 0.01, size=g.shape).astype(np.float32)
    if n > 1:
        hm[1, 0] = -np.abs(hm[1, 0]) - 0.01
    if n > 2 and k > 3:
        hm[2, 3] = -np.abs(hm[2, 3])
        hm[2, 3, 10, 10] = 0.0          # max == 0 exactly -> still "<= 0"
    if n > 3 and k > 2:
        m = float(hm[3, 2].max()) + 0.05
        hm[3, 2, 20, 30] = m
        hm[3, 2, 40, 7] = m
    if n > 4 and k > 5:
        hm[4, 5] = 0.25
    return hm


def org_sizes(n: int, seed: int) -> np.ndarray:
    """(org_w, org_h) per crop: crop 0 is the native 192x256, the rest are arbitrary
    3:4-ish sizes incl. odd ones (exercises the integer-floor centre of postprocess)."""
    rng = np.random.default_rng(seed + 1000)
    wh = np.stack([rng.integers(31, 900, size=n), rng.integers(41, 1200, size=n)], axis=1).astype(np.int32)
    wh[0] = (192, 256)
    return wh


def pad_shapes():
    """(h, w) of the crops fed to the reference's pad_image (taller, wider and exactly 3:4)."""
    return [(256, 192), (300, 100), (100, 300), (257, 193), (1, 1), (64, 47), (63, 48), (480, 640), (333, 250), (10, 7)]


def frame_case():
    """Synthetic 480x640 frame + detector output [n, 5] (x1, y1, x2, y2, conf) for the caller-level golden: a box
    whose padded crop is exactly 192x256, a wide one, a tall one touching the frame border, and one below the
    detector threshold (dropped by the caller)."""
    import numpy as np
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:480, 0:640]
    for cx, cy, r in [(150, 200, 40), (420, 260, 60), (600, 100, 30)]:
        blob = 120.0 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2.0 * r * r))
        frame = np.clip(frame.astype(np.float64) * 0.5 + blob[..., None], 0, 255).astype(np.uint8)
    boxes = np.array([[60.2, 110.4, 231.6, 345.7, 0.91],
                      [300.0, 180.0, 560.0, 330.0, 0.80],
                      [575.5, 20.0, 639.0, 300.0, 0.55],
                      [10.0, 10.0, 50.0, 50.0, 0.20]], dtype=np.float64)
    return frame, boxes


def coco_flip_pairs():
    """Mirror joint pairs of the COCO-17 layout (left/right eye, ear, shoulder, elbow, wrist, hip, knee, ankle)."""
    return [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]


def peaked_plan():
    """(variant, dataset, crops) of the peaked-checkpoint goldens: 136 / 136 / 100 / 1064 joints."""
    return [('s', 'coco', 8), ('b', 'coco', 8), ('l', 'coco_25', 4), ('h', 'wholebody', 8)]


def peaked_crops(n: int):
    """Half blob crops, half uniform-noise crops (seeds 21 / 22)."""
    from easy_vitpose_amd.synth import synthetic_crops
    return np.concatenate([synthetic_crops(n // 2, 21, 'blobs'), synthetic_crops(n - n // 2, 22, 'noise')])


def fullbatch_plan():
    """(variant, dataset, crops) of the full-batch goldens = the BASELINE configurations at their own batch sizes:
    configs[1] b/coco 256, configs[2] h/wholebody 128, configs[3] l/coco_25 64 (one frame of 64 persons), configs[4] b/ap10k 512, and
    s/coco at 256 (configs[0]'s model at a production batch)."""
    # (ascending keypoint count: the reference's config modules share one dict, a K = 17 model built after the K = 133 one keeps 133)
    return [('s', 'coco', 256), ('b', 'coco', 256), ('b', 'ap10k', 512), ('l', 'coco_25', 64), ('h', 'wholebody', 128)]


def fullbatch_crops(n: int):
    """Half blob crops, half uniform-noise crops (seeds 41 / 42), interleaved so that every chunk of the batch holds both kinds."""
    from easy_vitpose_amd.synth import synthetic_crops
    a, b = synthetic_crops(n // 2, 41, 'blobs'), synthetic_crops(n - n // 2, 42, 'noise')
    out = np.empty((n,) + a.shape[1:], a.dtype)
    out[0::2], out[1::2] = b, a
    return out


def content_plan():
    """(variant, dataset, crops) of the CONTENT-DEPENDENT goldens (round 6): 64 crops x every BASELINE model (ascending keypoint count, see fullbatch_plan)."""
    return [('s', 'coco', 64), ('b', 'coco', 64), ('b', 'ap10k', 64), ('l', 'coco_25', 64), ('h', 'wholebody', 64)]


def content_crops(n: int, seed: int = 61):
    """Crops whose CONTENT places the keypoints: one red, one green and one blue Gaussian blob (sigma 8-12 px, amplitude 150-200 on its colour
    channel) at seeded positions on a dark noisy background.  Returns (uint8 crops [n, 256, 192, 3], blobs [n, 3, 3] = (cy, cx, sigma) in crop
    pixels per colour).  With the content read-out (`content_state_dict`) joint k peaks at the centre of blob k % 3."""
    rng = np.random.default_rng(seed)
    H, W = 256, 192
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    out = np.empty((n, H, W, 3), np.uint8)
    blobs = np.empty((n, 3, 3), np.float64)
    for i in range(n):
        img = rng.uniform(15, 45, size=(H, W, 3)).astype(np.float32)
        for c in range(3):
            cy, cx = rng.uniform(20, H - 20), rng.uniform(20, W - 20)
            s = rng.uniform(8, 12)
            g = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
            img[..., c] += g * np.float32(rng.uniform(150, 200))
            blobs[i, c] = (cy, cx, s)
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return out, blobs


def content_state_dict(variant: str, dataset: str):
    """The CONTENT-DEPENDENT parity checkpoint: the seeded random checkpoint of `synthetic_state_dict(shape, 0)` -- every
    tensor of the backbone and of both deconvs random -- with `keypoint_head.final_layer` replaced by a ridge fit (tests/golden/
    fit_content_readout.py, run once in the build container with the CPU oracle; the fitted [K, 256] weights + [K] bias are the fixture
    tests/golden/content_readout_<variant>_<dataset>.npz) that makes joint k's heatmap a blob at the centre of colour blob k % 3 of the crop.
    The read-out is linear on the 256 head features, which are functions of the crop THROUGH all L encoder blocks: unlike the `peaked`
    checkpoint (positions designed into pos_embed / final_layer, content only modulating the peak height), an encoder error here moves
    coordinates."""
    import os
    from easy_vitpose_amd.configs import model_shape
    from easy_vitpose_amd.synth import synthetic_state_dict
    shp = model_shape(variant, dataset)
    sd = synthetic_state_dict(shp, 0)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), f'content_readout_{variant}_{dataset}.npz'))
    K = shp.num_keypoints
    assert z['weight'].shape == (K, 256) and z['bias'].shape == (K,)
    sd['keypoint_head.final_layer.weight'] = np.ascontiguousarray(z['weight'].reshape(K, 256, 1, 1).astype(np.float32))
    sd['keypoint_head.final_layer.bias'] = np.ascontiguousarray(z['bias'].astype(np.float32))
    return shp, sd


CONTENT_TIE = 3e-3   # a pixel closer than this to the maximum of the REFERENCE's heatmap is a tie: the device's heatmaps may differ from the reference's by
                     # up to ~1.6e-3 per value (fp16 operands; confidences within 1e-3 is a statement about the maximum), so either peak is a correct arg-max


CONTENT_COND_PX = 0.25   # the reference's own keypoint moves by more than this under a 3e-4 (rms) perturbation of its heatmap: ill-posed in the reference itself


def content_coordinate_error(got_yx: np.ndarray, z, lo: int = 0) -> "tuple[np.ndarray, np.ndarray]":
    """Per joint: distance (max over y / x, crop pixels) between the device's keypoint and the reference's -- where the reference's heatmap has further pixels
    within CONTENT_TIE of its maximum (neighbours of the arg-max or a second peak), to the NEAREST of the reference's answers: its own keypoint or the keypoint
    its decode gives when started from one of those pixels (full_content_*.npz: alt_yx / alt_margin = the four best pixels behind the arg-max, decoded by
    make_golden.py with the reference's post_dark_udp + transform_preds).  No joint is exempt.
    Returns (error [n, K], index of the matched answer [n, K]: 0 = the reference's keypoint, 1 .. 4 = an alternate)."""
    n = len(got_yx)
    ref = z['keypoints'][lo:lo + n, :, :2]
    cand = np.concatenate([ref[:, :, None], z['alt_yx'][lo:lo + n]], 2)                       # [n, K, 3, 2]
    ok = np.concatenate([np.ones(ref.shape[:2] + (1,), bool), z['alt_margin'][lo:lo + n] < CONTENT_TIE], 2)
    d = np.abs(got_yx[:, :, None, :] - cand).max(-1)
    d = np.where(ok, d, np.inf)
    return d.min(-1), d.argmin(-1)


def tracker_sequence():
    """Detections [n, 5] (x1, y1, x2, y2, score) per frame for the tracker golden: three people walking (one of them missed by
    the detector on two frames), a fourth entering at frame 6, two crossing paths, and detector-skipped (empty) frames as
    `yolo_step > 1` produces them."""
    rng = np.random.default_rng(77)
    frames = []
    for f in range(24):
        dets = []
        people = [(100 + 6 * f, 200 + 1 * f, 80, 200, 0.90), (500 - 5 * f, 180, 90, 220, 0.80), (300 + 2 * f, 100 + 4 * f, 60, 150, 0.70)]
        if f >= 6:
            people.append((50 + 9 * (f - 6), 300, 70, 180, 0.60))
        for i, (x, y, w, h, s) in enumerate(people):
            if i == 1 and f in (9, 10):
                continue                                   # missed detections
            j = rng.normal(0, 1.5, size=4)
            dets.append([x + j[0], y + j[1], x + w + j[2], y + h + j[3], s + 0.001 * f])
        if f in (4, 13, 14, 19):
            dets = []                                      # the detector did not run
        frames.append(np.asarray(dets, dtype=np.float64).reshape(-1, 5))
    return frames


def tracker_sequence_step(step: int):
    """The same people as `tracker_sequence`, with the detector run exactly when `VitInference.inference` runs it for
    `yolo_step = step` (easy_ViTPose/inference.py:235-236: `frame_counter % yolo_step == 0 or frame_counter < 3`) and empty
    input on every other frame -- the sequence the tracker of a `yolo_step > 1` video sees (constructed with
    `max_age = step, min_hits = 1`, inference.py:179-184)."""
    full = tracker_sequence()
    return [d if (f % step == 0 or f < 3) else np.empty((0, 5)) for f, d in enumerate(full)]
