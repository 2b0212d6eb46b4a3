"""CPU ORACLE for the ViTPose hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  The product path (``easy_vitpose_amd``) never does: it
raises when the HIP library is missing.

This is a from-spec restatement of the reference torch-CPU path
``VitInference._inference_torch`` (``easy_ViTPose/inference.py:320-328``):

    pre_img -> ViT backbone -> TopdownHeatmapSimpleHead -> keypoints_from_heatmaps
    (unbiased=True, use_udp=True) -> (y, x, conf)

written as plain functional torch fp32 ops for the model (it is a floating-point
kernel, so a torch fp32 reference is the right oracle) and plain numpy for the
decode.  Each function cites the reference lines it follows.

PINNING STATUS
* model + decode arithmetic: pinned -- checked in the build container against the
  reference itself (imported from /root/reference with stubs for the absent
  third-party modules) by ``tests/golden/make_golden.py``; its outputs are committed
  as fixtures under ``tests/golden/`` and ``tests/test_oracle_golden.py`` re-checks the
  oracle against them on every run (<=1e-5 on heatmaps, <=1e-4 px on keypoints).
* ``cv2.GaussianBlur`` / ``cv2.resize`` (opencv-python==4.8.0.76, ``requirements.txt:25``)
  are third-party and absent from /root/reference and from this image:
  **parity unpinned** at that boundary.  The blur is restated from OpenCV's
  published algorithm (separable 11-tap, sigma = 0.3*((k-1)/2-1)+0.8 = 2.0, float32
  kernel normalised to sum 1, BORDER_REFLECT_101); the golden generator uses the
  same restatement as its cv2 shim.  ``cv2.resize`` is only ever the identity here
  (all BASELINE configs feed 256x192 crops) and the oracle asserts that.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

MEAN = np.array([0.485, 0.456, 0.406])  # easy_ViTPose/inference.py:32
STD = np.array([0.229, 0.224, 0.225])   # easy_ViTPose/inference.py:33
IMG_W, IMG_H = 192, 256                 # configs/ViTPose_common.py:30
HM_W, HM_H = 48, 64                     # configs/ViTPose_common.py:31


# --------------------------------------------------------------------------- a1
def pre_img(img_u8: np.ndarray):
    """``VitInference.pre_img`` (easy_ViTPose/inference.py:314-318).

    ``cv2.resize(img, (192,256), INTER_LINEAR) / 255`` in float64, normalise,
    HWC->CHW, cast to float32.  The resize is restated only as the identity
    (INTER_LINEAR at equal size returns the input)."""
    org_h, org_w = img_u8.shape[:2]
    assert (org_h, org_w) == (IMG_H, IMG_W), \
        'oracle restates cv2.resize only as identity: crops must be 256x192'
    x = img_u8.astype(np.float64) / 255
    x = ((x - MEAN) / STD).transpose(2, 0, 1)[None].astype(np.float32)
    return x, org_h, org_w


# --------------------------------------------------------------------- a2..a10
def _t(sd, name):
    v = sd[name]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


@torch.no_grad()
def backbone_forward(sd, x: torch.Tensor, depth: int, num_heads: int) -> torch.Tensor:
    """``ViT.forward`` (vit_models/backbone/vit.py:375-389) in eval mode.

    Returns the token matrix ``[B, 192, D]`` AFTER last_norm (the reference then
    permutes it to ``[B, D, 16, 12]``, vit.py:388)."""
    # PatchEmbed: Conv2d(3, D, k=16, s=16, padding=4+2*(1//2-1)=2)  vit.py:222-228
    w = _t(sd, 'backbone.patch_embed.proj.weight')
    b = _t(sd, 'backbone.patch_embed.proj.bias')
    x = F.conv2d(x, w, b, stride=16, padding=2)
    B, D, Hp, Wp = x.shape
    x = x.view(B, D, Hp * Wp).transpose(1, 2)
    # pos embed, cls-token row broadcast onto every token   vit.py:379-382
    pos = _t(sd, 'backbone.pos_embed')
    x = x + pos[:, 1:] + pos[:, :1]
    hd = D // num_heads
    scale = hd ** -0.5  # vit.py:156
    for i in range(depth):
        p = f'backbone.blocks.{i}.'
        # x = x + attn(norm1(x))   vit.py:203  (DropPath is identity in eval, vit.py:29-30)
        y = F.layer_norm(x, (D,), _t(sd, p + 'norm1.weight'), _t(sd, p + 'norm1.bias'), eps=1e-6)
        qkv = F.linear(y, _t(sd, p + 'attn.qkv.weight'), _t(sd, p + 'attn.qkv.bias'))  # vit.py:166
        qkv = qkv.reshape(B, Hp * Wp, 3, num_heads, hd).permute(2, 0, 3, 1, 4)         # vit.py:167
        q, k, v = qkv[0], qkv[1], qkv[2]
        q = q * scale                                                                   # vit.py:170
        attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)                                # vit.py:171-173
        y = (attn @ v).transpose(1, 2).reshape(B, Hp * Wp, D)                           # vit.py:176
        y = F.linear(y, _t(sd, p + 'attn.proj.weight'), _t(sd, p + 'attn.proj.bias'))  # vit.py:177
        x = x + y
        # x = x + mlp(norm2(x))    vit.py:204 ; Mlp vit.py:136-141 ; nn.GELU exact erf
        y = F.layer_norm(x, (D,), _t(sd, p + 'norm2.weight'), _t(sd, p + 'norm2.bias'), eps=1e-6)
        y = F.linear(y, _t(sd, p + 'mlp.fc1.weight'), _t(sd, p + 'mlp.fc1.bias'))
        y = F.gelu(y)
        y = F.linear(y, _t(sd, p + 'mlp.fc2.weight'), _t(sd, p + 'mlp.fc2.bias'))
        x = x + y
    x = F.layer_norm(x, (D,), _t(sd, 'backbone.last_norm.weight'), _t(sd, 'backbone.last_norm.bias'), eps=1e-6)
    return x


@torch.no_grad()
def head_forward(sd, tokens: torch.Tensor) -> torch.Tensor:
    """``TopdownHeatmapSimpleHead.forward`` (topdown_heatmap_simple_head.py:188-193):
    2 x [ConvTranspose2d(k=4,s=2,p=1,bias=False) + BatchNorm2d(eval, eps 1e-5) + ReLU]
    (:291-321, cfg topdown_heatmap_base_head.py:105-120) then Conv2d(256, K, 1) (:124-130)."""
    B, T, D = tokens.shape
    x = tokens.permute(0, 2, 1).reshape(B, D, 16, 12)  # vit.py:388
    h = 'keypoint_head.deconv_layers.'
    for idx in (0, 3):
        x = F.conv_transpose2d(x, _t(sd, f'{h}{idx}.weight'), None, stride=2, padding=1, output_padding=0)
        x = F.batch_norm(x, _t(sd, f'{h}{idx + 1}.running_mean'), _t(sd, f'{h}{idx + 1}.running_var'),
                         _t(sd, f'{h}{idx + 1}.weight'), _t(sd, f'{h}{idx + 1}.bias'),
                         training=False, eps=1e-5)
        x = F.relu(x)
    x = F.conv2d(x, _t(sd, 'keypoint_head.final_layer.weight'), _t(sd, 'keypoint_head.final_layer.bias'))
    return x


@torch.no_grad()
def model_forward(sd, x, depth: int, num_heads: int) -> np.ndarray:
    """``ViTPose.forward`` (vit_models/model.py:23-24): heatmaps ``[B, K, 64, 48]`` fp32."""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return head_forward(sd, backbone_forward(sd, x.float(), depth, num_heads)).numpy()


# -------------------------------------------------------------------- a13..a15
def get_max_preds(heatmaps: np.ndarray):
    """``_get_max_preds`` (vit_utils/top_down_eval.py:82-114): first-index argmax,
    x = idx % W, y = idx // W as float32, coords = -1 where maxval <= 0."""
    N, K, _, W = heatmaps.shape
    flat = heatmaps.reshape((N, K, -1))
    idx = np.argmax(flat, 2).reshape((N, K, 1))
    maxvals = np.amax(flat, 2).reshape((N, K, 1))
    preds = np.tile(idx, (1, 1, 2)).astype(np.float32)
    preds[:, :, 0] = preds[:, :, 0] % W
    preds[:, :, 1] = preds[:, :, 1] // W
    preds = np.where(np.tile(maxvals, (1, 1, 2)) > 0.0, preds, -1)
    return preds, maxvals


def gaussian_kernel_1d(ksize: int = 11) -> np.ndarray:
    """OpenCV ``getGaussianKernel(ksize, sigma<=0)`` for float32 images:
    sigma = 0.3*((ksize-1)*0.5 - 1) + 0.8 ; w_i = exp(-(i-c)^2/(2 sigma^2)) ; sum -> 1.
    (ksize 11 is not one of the fixed small kernels {1,3,5,7}.)"""
    sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    c = (ksize - 1) * 0.5
    i = np.arange(ksize, dtype=np.float64)
    w = np.exp(-((i - c) ** 2) / (2.0 * sigma * sigma))
    w = w / w.sum()
    return w.astype(np.float32)


def gaussian_blur(hm: np.ndarray, ksize: int = 11) -> np.ndarray:
    """``cv2.GaussianBlur(hm, (k,k), 0)`` restated: separable, rows then columns,
    float32 accumulation, BORDER_REFLECT_101 (OpenCV default).  PARITY UNPINNED vs
    the real OpenCV binary (absent here); see module docstring."""
    g = gaussian_kernel_1d(ksize)
    r = ksize // 2
    H, W = hm.shape
    src = hm.astype(np.float32)
    pad = np.pad(src, ((0, 0), (r, r)), mode='reflect')  # numpy 'reflect' == REFLECT_101
    tmp = np.zeros((H, W), dtype=np.float32)
    for t in range(ksize):
        tmp += g[t] * pad[:, t:t + W]
    pad = np.pad(tmp, ((r, r), (0, 0)), mode='reflect')
    out = np.zeros((H, W), dtype=np.float32)
    for t in range(ksize):
        out += g[t] * pad[t:t + H, :]
    return out


def post_dark_udp(coords: np.ndarray, batch_heatmaps: np.ndarray, kernel: int = 11) -> np.ndarray:
    """``post_dark_udp`` (vit_utils/top_down_eval.py:354-415), statement for statement,
    including the flat index arithmetic into the edge-padded map (so that coords of
    -1 wrap exactly as the reference's fancy indexing does).  Must be called per
    crop (N == 1): the reference's float32 index arithmetic (:393-395) is wrong for
    N*K*(H+2)*(W+2) > 2**24 and VitInference always calls it with N == 1."""
    B, K, H, W = batch_heatmaps.shape
    N = coords.shape[0]
    assert (B == 1 or B == N)
    for heatmaps in batch_heatmaps:
        for i in range(K):
            heatmaps[i] = gaussian_blur(heatmaps[i], kernel)        # :383-385
    np.clip(batch_heatmaps, 0.001, 50, batch_heatmaps)              # :386
    np.log(batch_heatmaps, batch_heatmaps)                          # :387
    pad = np.pad(batch_heatmaps, ((0, 0), (0, 0), (1, 1), (1, 1)), mode='edge').flatten()  # :389-391
    index = coords[..., 0] + 1 + (coords[..., 1] + 1) * (W + 2)     # :393 (float32 arithmetic)
    index += (W + 2) * (H + 2) * np.arange(0, B * K).reshape(-1, K)  # :394
    index = index.astype(int).reshape(-1, 1)                        # :395
    i_ = pad[index]
    ix1 = pad[index + 1]
    iy1 = pad[index + W + 2]
    ix1y1 = pad[index + W + 3]
    ix1_y1_ = pad[index - W - 3]
    ix1_ = pad[index - 1]
    iy1_ = pad[index - 2 - W]
    dx = 0.5 * (ix1 - ix1_)
    dy = 0.5 * (iy1 - iy1_)
    derivative = np.concatenate([dx, dy], axis=1).reshape(N, K, 2, 1)
    dxx = ix1 - 2 * i_ + ix1_
    dyy = iy1 - 2 * i_ + iy1_
    dxy = 0.5 * (ix1y1 - ix1 - iy1 + i_ + i_ - ix1_ - iy1_ + ix1_y1_)
    hessian = np.concatenate([dxx, dxy, dxy, dyy], axis=1).reshape(N, K, 2, 2)
    hessian = np.linalg.inv(hessian + np.finfo(np.float32).eps * np.eye(2))  # float64  :411-413
    coords -= np.einsum('ijmn,ijnk->ijmk', hessian, derivative).squeeze()    # :414
    return coords


def transform_preds_udp(coords: np.ndarray, center, scale, output_size) -> np.ndarray:
    """``transform_preds(..., use_udp=True)`` (post_processing/post_transforms.py:150-194)."""
    scale_x = scale[0] / (output_size[0] - 1.0)
    scale_y = scale[1] / (output_size[1] - 1.0)
    target = np.ones_like(coords)
    target[:, 0] = coords[:, 0] * scale_x + center[0] - scale[0] * 0.5
    target[:, 1] = coords[:, 1] * scale_y + center[1] - scale[1] * 0.5
    return target


def keypoints_from_heatmaps(heatmaps: np.ndarray, center, scale, kernel: int = 11):
    """``keypoints_from_heatmaps(unbiased=True, use_udp=True)`` path
    (vit_utils/top_down_eval.py:493-641): copy (:545), _get_max_preds + post_dark_udp
    (:586-589), transform_preds per person (:634-636), maxvals unchanged."""
    heatmaps = heatmaps.copy()
    N, K, H, W = heatmaps.shape
    preds, maxvals = get_max_preds(heatmaps)
    preds = post_dark_udp(preds, heatmaps, kernel=kernel)
    for i in range(N):
        preds[i] = transform_preds_udp(preds[i], center[i], scale[i], [W, H])
    return preds, maxvals


def postprocess(heatmaps: np.ndarray, org_w: int, org_h: int) -> np.ndarray:
    """``VitInference.postprocess`` (easy_ViTPose/inference.py:187-205): integer-floor
    centre, returns ``[N, K, 3]`` = (y, x, conf)."""
    points, prob = keypoints_from_heatmaps(heatmaps,
                                           center=np.array([[org_w // 2, org_h // 2]]),
                                           scale=np.array([[org_w, org_h]]))
    return np.concatenate([points[:, :, ::-1], prob], axis=2)


def decode_per_crop(heatmaps: np.ndarray, org_wh=None) -> np.ndarray:
    """Decode a batch the way VitInference does: one ``postprocess`` call per crop."""
    N = heatmaps.shape[0]
    out = []
    for n in range(N):
        w, h = (IMG_W, IMG_H) if org_wh is None else (int(org_wh[n][0]), int(org_wh[n][1]))
        out.append(postprocess(heatmaps[n:n + 1].astype(np.float32), w, h))
    return np.concatenate(out, axis=0).astype(np.float32)


# ------------------------------------------------------------------------ a10
def inference_torch(sd, depth: int, num_heads: int, img_u8: np.ndarray) -> np.ndarray:
    """``VitInference._inference_torch`` (easy_ViTPose/inference.py:320-328) for one crop:
    returns ``[1, K, 3]`` float32 (y, x, conf) in crop pixels."""
    x, org_h, org_w = pre_img(img_u8)
    hm = model_forward(sd, x, depth, num_heads)
    return postprocess(hm, org_w, org_h).astype(np.float32)


def to_torch_state_dict(sd_np) -> "dict[str, torch.Tensor]":
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}


def flip_back(output_flipped: np.ndarray, flip_pairs) -> np.ndarray:
    """``flip_back(..., target_type='GaussianHeatmap')`` (vit_utils/post_processing/post_transforms.py:110-147):
    swap the mirrored joint channels, then reverse the width axis."""
    assert output_flipped.ndim == 4
    back = output_flipped.copy()
    for left, right in flip_pairs:
        back[:, left] = output_flipped[:, right]
        back[:, right] = output_flipped[:, left]
    return back[..., ::-1]


def flip_test_heatmaps(sd, x: np.ndarray, depth: int, num_heads: int, flip_pairs, shift_heatmap: bool = False) -> np.ndarray:
    """Flip-test: heatmaps of the crops averaged with the flipped-back heatmaps of their mirror images
    (head ``inference_model``, topdown_heatmap_simple_head.py:195-218 incl. the optional one-pixel shift :213-215;
    the 0.5 (a + b) average is what the flip-test consumer computes)."""
    a = model_forward(sd, x, depth, num_heads)
    b = flip_back(model_forward(sd, np.ascontiguousarray(x[..., ::-1]), depth, num_heads), flip_pairs).copy()
    if shift_heatmap:
        b[:, :, :, 1:] = b.copy()[:, :, :, :-1]
    return (0.5 * (a + b)).astype(np.float32)
