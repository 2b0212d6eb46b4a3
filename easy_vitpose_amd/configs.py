"""Shape contract of the ViTPose hot path.

Mirrors (as a static table, not as importable python-dict configs) what the
reference spreads over ``easy_ViTPose/configs/ViTPose_common.py:65-195``
(``model_small/base/large/huge``: embed_dim / depth / num_heads),
``ViTPose_common.py:29-40`` (``data_cfg``: image 192x256, heatmap 48x64) and the
per-dataset ``out_channels`` (``configs/ViTPose_coco.py:4-18``,
``ViTPose_coco_25.py:4-20``, ``ViTPose_wholebody.py:4-20``, ``ViTPose_ap10k.py:4-22``,
``ViTPose_mpii.py``, ``ViTPose_aic.py``, ``ViTPose_apt36k.py``).

Name lookup follows ``vit_utils/util.py:20-41`` (``MODEL_ABBR_MAP``,
``infer_dataset_by_path``, ``dyn_model_import``).
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass

# input crop (W, H) and heatmap (W, H): ViTPose_common.py:29-31
IMAGE_SIZE = (192, 256)
HEATMAP_SIZE = (48, 64)
IMG_H, IMG_W = 256, 192
HM_H, HM_W = 64, 48
PATCH = 16
TOKENS = 192  # 16 x 12 patches
GRID_H, GRID_W = 16, 12
DECONV_CH = 256  # num_deconv_filters=(256, 256)

MODEL_ABBR_MAP = {'s': 'small', 'b': 'base', 'l': 'large', 'h': 'huge'}

# (embed_dim, depth, num_heads)   ViTPose_common.py:65-195
VARIANTS = {
    's': (384, 12, 12),
    'b': (768, 12, 12),
    'l': (1024, 24, 16),
    'h': (1280, 32, 16),
}

# dataset -> number of keypoints (head out_channels)
DATASET_KEYPOINTS = {
    'coco': 17,
    'coco_25': 25,
    'wholebody': 133,
    'mpii': 16,
    'aic': 14,
    'ap10k': 17,
    'apt36k': 17,
}


@dataclass(frozen=True)
class ModelShape:
    """Static shape of one (variant, dataset) model."""
    variant: str
    embed_dim: int
    depth: int
    num_heads: int
    num_keypoints: int

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def mlp_dim(self) -> int:
        return 4 * self.embed_dim

    def gflop_per_person(self) -> float:
        """Algorithmic FLOPs (2*M*N*K of every matmul/conv incl. attention core),
        the formula of SURVEY.md section 8(d)."""
        D, L, K = self.embed_dim, self.depth, self.num_keypoints
        return (L * (4608 * D * D + 147456 * D) + 294912 * D + 1572864 * D
                + 1610612736 + 1572864 * K) / 1e9


def model_shape(variant: str, dataset: str | None = None, num_keypoints: int | None = None) -> ModelShape:
    assert variant in VARIANTS, f'The model name {variant} is not valid'
    if num_keypoints is None:
        assert dataset in DATASET_KEYPOINTS, 'The specified dataset is not valid'
        num_keypoints = DATASET_KEYPOINTS[dataset]
    D, L, h = VARIANTS[variant]
    return ModelShape(variant, D, L, h, int(num_keypoints))


def infer_dataset_by_path(model_path: str) -> str:
    """Same filename convention as ``vit_utils/util.py:28-34``:
    ``vitpose-b-coco_25.pth`` -> ``coco_25``. Raises ValueError like the reference."""
    model = os.path.basename(model_path)
    m = re.search(r'-([a-zA-Z0-9_]+)\.[pth, onnx, engine]', model)
    if not m:
        raise ValueError('Could not infer the dataset from ckpt name, specify it')
    return m.group(1)


def infer_variant_from_state_dict(sd) -> str:
    """The reference needs ``model_name`` for .pth files; the shapes also tell."""
    D = int(sd['backbone.pos_embed'].shape[-1])
    for v, (d, _, _) in VARIANTS.items():
        if d == D:
            return v
    raise ValueError(f'unknown embed_dim {D}')
