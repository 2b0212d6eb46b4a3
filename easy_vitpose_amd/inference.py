"""``VitInference`` -- the reference's public class (easy_ViTPose/inference.py:52-336)
with its backend slot filled by the MI355X-native HIP path.

Kept from the reference: constructor signature (:81-90), ``reset`` (:174-185),
``postprocess`` (:187-205), ``inference(img) -> {id: (K,3) (y,x,score)}`` (:221-281),
``pre_img`` (:314-318), the state attributes (:112-116, :274-279) and the exception
types.  Changed on purpose: the per-box Python loop that ran the model once per
crop (:259-272) becomes ONE batched call into the C ABI (``_inference_batch``);
``_inference`` (single crop, :207-219) is still there and returns ``[1, K, 3]``.

Out of scope here: the YOLO detector and the SORT tracker
are third-party / CPU-side; ``yolo`` may be a path (needs ``ultralytics``) or any
callable ``img_rgb -> ndarray[n, 5] (x1, y1, x2, y2, conf)``, ``tracker`` any object
with the SORT ``update`` interface.  ``draw`` is not provided.
"""
from __future__ import annotations

import os
import typing
from typing import Optional

import numpy as np

from .configs import IMG_H, IMG_W, infer_dataset_by_path, infer_variant_from_state_dict, model_shape
from .cropprep import crop_params, resize_linear_u8
from .engine import VitPoseHip, decode_heatmaps

__all__ = ['VitInference']

MEAN = [0.485, 0.456, 0.406]  # inference.py:32
STD = [0.229, 0.224, 0.225]   # inference.py:33

DETC_TO_YOLO_YOLOC = {        # inference.py:36-48
    'human': [0], 'cat': [15], 'dog': [16], 'horse': [17], 'sheep': [18], 'cow': [19],
    'elephant': [20], 'bear': [21], 'zebra': [22], 'giraffe': [23],
    'animals': [15, 16, 17, 18, 19, 20, 21, 22, 23],
}


def pad_image(image: np.ndarray, aspect_ratio: float):
    """Zero-pad a crop to the given W/H aspect ratio; returns (padded, (left_pad, top_pad)).
    Same contract as ``vit_utils/inference.py:41-70``."""
    h, w = image.shape[:2]
    left = top = 0
    if w / h < aspect_ratio:
        tw = int(aspect_ratio * h)
        left = (tw - w) // 2
        out = np.zeros((h, tw) + image.shape[2:], dtype=image.dtype)
        out[:, left:left + w] = image
    else:
        th = int(w / aspect_ratio)
        top = (th - h) // 2
        out = np.zeros((th, w) + image.shape[2:], dtype=image.dtype)
        out[top:top + h] = image
    return out, (left, top)


class VitInference:
    """ViTPose inference with the MI355X HIP backend (see module docstring)."""

    def __init__(self, model,
                 yolo,
                 model_name: Optional[str] = None,
                 det_class: Optional[str] = None,
                 dataset: Optional[str] = None,
                 yolo_size: Optional[int] = 320,
                 device: Optional[str] = None,
                 is_video: Optional[bool] = False,
                 single_pose: Optional[bool] = False,
                 yolo_step: Optional[int] = 1,
                 *, dtype: str = 'fp16', max_batch: int = 64, tracker=None):
        state_dict = None
        if isinstance(model, (str, os.PathLike)):
            assert os.path.isfile(model), f'The model file {model} does not exist'
            assert not str(model).endswith(('.onnx', '.engine')), \
                'the HIP backend loads .pth checkpoints only (no ONNX / TensorRT dispatch)'
            if dataset is None:
                dataset = infer_dataset_by_path(str(model))
        else:  # an in-memory state dict (extension; used by tests / benchmarks)
            state_dict = model
            assert dataset is not None, 'dataset must be given with an in-memory state dict'
        if callable(yolo):
            self.yolo = yolo
        else:
            assert os.path.isfile(yolo), f'The YOLOv8 model {yolo} does not exist'
            try:
                from ultralytics import YOLO
            except ModuleNotFoundError as e:
                raise ModuleNotFoundError('ultralytics is not installed: pass a callable detector as `yolo`') from e
            self._yolo_model = YOLO(yolo, task='detect')
            self.yolo = self._call_ultralytics

        if device is None:
            device = 'cuda'
        assert str(device).startswith('cuda'), 'the HIP backend runs on an AMD GPU only (no CPU fallback)'
        self.device = device
        self.yolo_size = yolo_size
        self.yolo_step = yolo_step
        self.is_video = is_video
        self.single_pose = single_pose
        self._tracker_factory = tracker
        self.reset()

        self.save_state = True
        self._img = None
        self._yolo_res = None
        self._tracker_res = None
        self._keypoints = None
        self._scores_bbox = None

        assert dataset in ['mpii', 'coco', 'coco_25', 'wholebody', 'aic', 'ap10k', 'apt36k', 'custom'], \
            'The specified dataset is not valid'
        self.dataset = dataset
        if det_class is None:
            det_class = 'animals' if dataset in ['ap10k', 'apt36k'] else 'human'
        self.yolo_classes = DETC_TO_YOLO_YOLOC[det_class]
        assert model_name in [None, 's', 'b', 'l', 'h'], f'The model name {model_name} is not valid'

        if state_dict is None:
            import torch
            ckpt = torch.load(model, map_location='cpu', weights_only=True)
            state_dict = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt
        if model_name is None:
            model_name = infer_variant_from_state_dict(state_dict)
        nk = None
        if dataset == 'custom':
            nk = int(np.asarray(state_dict['keypoint_head.final_layer.bias']).shape[0])
        self.target_size = [IMG_W, IMG_H]  # data_cfg['image_size'], ViTPose_common.py:30
        dev_id = int(str(device).split(':')[1]) if ':' in str(device) else 0
        self._vit_pose = VitPoseHip(model_shape(model_name, None if nk else dataset, nk), state_dict,
                                    dtype=dtype, device_id=dev_id, max_batch=max_batch)
        self._inference = self._inference_hip

    # ----------------------------------------------------------------- glue
    def _call_ultralytics(self, img_rgb):
        results = self._yolo_model(img_rgb[..., ::-1], verbose=False, imgsz=self.yolo_size,
                                   device=self.device if self.device != 'cuda' else 0,   # inference.py:238
                                   classes=self.yolo_classes)[0]
        self._yolo_res = results
        return results.boxes.data.cpu().numpy()[:, :5]

    def reset(self):
        """Ready for a new video (inference.py:174-185)."""
        use_tracker = self.is_video and not self.single_pose
        self.tracker = None
        if use_tracker:
            if self._tracker_factory is None:   # the reference's own choice and parameters (inference.py:179-184):
                from .tracker import Sort       # with detector-skipped frames (yolo_step > 1) a coasting track's hit streak restarts at every
                min_hits = 3 if self.yolo_step == 1 else 1   # re-match, so min_hits must be 1 there or no pose is reported on detector frames
                self.tracker = Sort(max_age=self.yolo_step, min_hits=min_hits, iou_threshold=0.3)
            else:
                self.tracker = self._tracker_factory()
        self.frame_counter = 0

    @classmethod
    def postprocess(cls, heatmaps, org_w, org_h):
        """Heatmaps ``[N,K,64,48]`` -> ``[N,K,3]`` (y, x, conf); GPU decode kernel with the
        reference's arguments (inference.py:187-205)."""
        n = heatmaps.shape[0]
        wh = np.tile(np.array([[org_w, org_h]], dtype=np.int32), (n, 1))
        return decode_heatmaps(np.asarray(heatmaps, dtype=np.float32), wh)

    def pre_img(self, img):
        """inference.py:314-318 -- kept for API compatibility (host float path)."""
        org_h, org_w = img.shape[:2]
        img_input = resize_linear_u8(img, self.target_size) / 255
        img_input = ((img_input - MEAN) / STD).transpose(2, 0, 1)[None].astype(np.float32)
        return img_input, org_h, org_w

    def _inference_batch(self, crops: "list[np.ndarray]") -> np.ndarray:
        """N crops (uint8 RGB, any size, already padded to 3:4) -> ``[N, K, 3]``: OpenCV-style 8-bit bilinear
        resize on the host (cropprep.resize_linear_u8; identity for 256x192), then ONE call into the HIP
        library (normalisation is on device).  `inference(img)` does not use this: it hands the whole frame
        to the device crop kernel (`vp_infer_frame`)."""
        if len(crops) == 0:
            return np.empty((0, self._vit_pose.K, 3), dtype=np.float32)
        wh = np.array([[c.shape[1], c.shape[0]] for c in crops], dtype=np.int32)
        batch = np.stack([resize_linear_u8(np.ascontiguousarray(c), self.target_size) for c in crops])
        return self._vit_pose.infer(batch, wh)

    def _inference_hip(self, img: np.ndarray) -> np.ndarray:
        """Drop-in for ``_inference_torch`` (inference.py:320-328): one crop -> ``[1, K, 3]``."""
        return self._inference_batch([img])

    # ------------------------------------------------------------ inference
    def inference(self, img: np.ndarray) -> "dict[typing.Any, typing.Any]":
        """inference.py:221-281 with the crop loop batched."""
        res_pd = np.empty((0, 5))
        if (self.tracker is None or (self.frame_counter % self.yolo_step == 0 or self.frame_counter < 3)):
            det = np.asarray(self.yolo(img), dtype=np.float64).reshape((-1, 5))
            res_pd = det[det[:, 4] > 0.35].reshape((-1, 5))
        self.frame_counter += 1

        ids = None
        if self.tracker is not None:
            res_pd = self.tracker.update(res_pd)
            ids = res_pd[:, 5].astype(int).tolist()
        bboxes = res_pd[:, :4].round().astype(int)
        scores = res_pd[:, 4].tolist()
        pad_bbox = 10
        if ids is None:
            ids = range(len(bboxes))

        # crop + zero-pad to 3:4 + resize + normalise all happen on device from ONE copy of the frame
        params = crop_params(bboxes, img.shape[:2], pad_bbox) if len(bboxes) else np.zeros((0, 8), np.int32)
        for i in range(len(bboxes)):                       # keep the reference's in-place box update (:261-262)
            bboxes[i] = (params[i, 0], params[i, 1], params[i, 0] + params[i, 2], params[i, 1] + params[i, 3])
        offsets = [np.array([p[1] - p[5], p[0] - p[4]]) for p in params]   # bbox[:2][::-1] - [top_pad, left_pad]
        kps = self._vit_pose.infer_frame(np.ascontiguousarray(img), params)

        frame_keypoints, scores_bbox = {}, {}
        for i, (id_, score) in enumerate(zip(ids, scores)):
            k = kps[i]
            k[:, :2] += offsets[i]
            frame_keypoints[id_] = k
            scores_bbox[id_] = score
        if self.save_state:
            self._img = img
            self._tracker_res = (bboxes, ids, scores)
            self._keypoints = frame_keypoints
            self._scores_bbox = scores_bbox
        return frame_keypoints

    def draw(self, show_yolo=True, show_raw_yolo=False, confidence_threshold=0.5):
        raise NotImplementedError('drawing (cv2/matplotlib) is outside the HIP hot path; use the keypoint dict')
