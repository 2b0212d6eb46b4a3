#!/usr/bin/env python3
"""Command line driver over ``VitInference`` -- the frame loop of the reference's top-level ``inference.py:19-145`` (same option
names) for the HIP path: read an image or a frame stack, detect (ultralytics if it is installed, else boxes from a file), run the
pose path, track across frames (``is_video``), report FPS and write the ``--save-json`` file in the reference's wire format.

    python -m easy_vitpose_amd.cli --input frame.png --model vitpose-b-coco.pth --yolo yolov8s.pt --output-path out --save-json
    python -m easy_vitpose_amd.cli --input clip.npy --synthetic b --boxes boxes.json --output-path out --save-json

Not rebuilt (outside the hot path): drawing / preview windows (``--show``, ``--save-img``: OpenCV) and video
decoding -- a video is accepted as a ``.npy`` stack ``[frames, H, W, 3]`` uint8 RGB, or as a directory of image files.
``--boxes`` (JSON: one ``[[x1, y1, x2, y2, conf], ...]`` list per frame, or a single list used for every frame) replaces the
detector when ultralytics is not installed (there is no network in the build image to fetch it or its weights).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np


def _read_frames(path: str, rotate: int):
    if os.path.isdir(path):
        from PIL import Image
        files = sorted(f for f in os.listdir(path) if f.lower().rsplit('.', 1)[-1] in ('png', 'jpg', 'jpeg', 'bmp'))
        return [np.array(Image.open(os.path.join(path, f)).convert('RGB').rotate(rotate)) for f in files], True
    ext = path[path.rfind('.') + 1:].lower()
    if ext == 'npy':
        arr = np.load(path)
        assert arr.dtype == np.uint8 and arr.ndim == 4 and arr.shape[3] == 3, 'frame stack must be uint8 [frames, H, W, 3] RGB'
        if rotate:
            arr = np.rot90(arr, k=(rotate // 90) % 4, axes=(1, 2))
        return list(np.ascontiguousarray(arr)), True
    assert ext not in ('avi', 'mp4', 'mov'), 'video decoding needs OpenCV: convert the clip to a .npy frame stack or a directory of images'
    from PIL import Image
    return [np.array(Image.open(path).convert('RGB').rotate(rotate))], False


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--input', required=True, help='image file, .npy frame stack, or directory of images')
    ap.add_argument('--output-path', default='', help='output directory (required by --save-json)')
    ap.add_argument('--model', default=None, help='ViTPose checkpoint (.pth)')
    ap.add_argument('--synthetic', default=None, choices=['s', 'b', 'l', 'h'], help='seeded peaked synthetic checkpoint of this size instead of --model')
    ap.add_argument('--yolo', default=None, help='ultralytics detector weights')
    ap.add_argument('--boxes', default=None, help='JSON file with detector boxes (replaces --yolo)')
    ap.add_argument('--dataset', default=None)
    ap.add_argument('--det-class', default=None)
    ap.add_argument('--model-name', default=None, choices=['s', 'b', 'l', 'h'])
    ap.add_argument('--yolo-size', type=int, default=320)
    ap.add_argument('--conf-threshold', type=float, default=0.5)
    ap.add_argument('--rotate', type=int, default=0, choices=[0, 90, 180, 270])
    ap.add_argument('--yolo-step', type=int, default=1)
    ap.add_argument('--single-pose', action='store_true')
    ap.add_argument('--save-json', action='store_true')
    ap.add_argument('--show', action='store_true')
    ap.add_argument('--save-img', action='store_true')
    ap.add_argument('--max-batch', type=int, default=64)
    ap.add_argument('--dtype', default='fp16', choices=['fp16', 'bf16'])
    args = ap.parse_args(argv)
    assert not (args.show or args.save_img), 'drawing / preview (OpenCV) is outside the HIP hot path: use --save-json'
    assert not args.save_json or args.output_path, 'Specify an output path if using save-img or save-json flags'
    assert (args.model is None) != (args.synthetic is None), 'give exactly one of --model / --synthetic'

    from easy_vitpose_amd import VitInference
    from easy_vitpose_amd.configs import infer_dataset_by_path, model_shape
    from easy_vitpose_amd.jsonio import COCO17_JOINTS, save_json
    frames, is_video = _read_frames(args.input, args.rotate)
    dataset = args.dataset or (infer_dataset_by_path(args.model) if args.model else 'coco')

    detector = args.yolo
    if args.boxes is not None:
        boxes = json.load(open(args.boxes))
        per_frame = bool(boxes) and isinstance(boxes[0], list) and bool(boxes[0]) and isinstance(boxes[0][0], list)
        state = {'i': 0}

        def detector(img):   # noqa: F811 -- one call per detector frame, in order
            b = boxes[min(state['i'], len(boxes) - 1)] if per_frame else boxes
            state['i'] += 1
            return np.asarray(b, dtype=np.float64).reshape(-1, 5)
    assert detector is not None, 'give --yolo (ultralytics weights) or --boxes'

    state_dict = None
    if args.synthetic:
        from easy_vitpose_amd.synth import synthetic_state_dict
        state_dict = synthetic_state_dict(model_shape(args.synthetic, dataset), 0, peaked=True)
    model = VitInference(state_dict if state_dict is not None else args.model, detector, args.model_name or args.synthetic,
                         args.det_class, dataset, args.yolo_size, is_video=is_video, single_pose=args.single_pose,
                         yolo_step=args.yolo_step, dtype=args.dtype, max_batch=args.max_batch)
    print(f'>>> Model loaded: {args.model or "synthetic ViTPose-" + args.synthetic.upper()}')
    print(f'>>> Running inference on {args.input}')
    keypoints, dts = [], []
    for img in frames:
        t0 = time.time()
        keypoints.append(model.inference(img))
        dts.append(time.time() - t0)
    if is_video:
        tot = sum(len(k) for k in keypoints)
        print(f'>>> Mean inference FPS: {1 / np.mean(dts):.2f}')
        print(f'>>> Total poses predicted: {tot} mean per frame: {tot / len(frames):.2f}')
        print(f'>>> Mean FPS per pose: {tot / max(sum(dts), 1e-9):.2f}')
    if args.save_json:
        out_dir = os.path.join(args.output_path, os.path.basename(args.input.rstrip('/')))
        os.makedirs(out_dir, exist_ok=True)
        base = os.path.basename(args.input.rstrip('/'))
        stem = base[:base.rfind('.')] if '.' in base else base
        path = os.path.join(out_dir, stem + '_result.json')
        print('>>> Saving output json')
        save_json(path, keypoints, COCO17_JOINTS if model._vit_pose.K == 17 and dataset == 'coco' else None)
    return 0


if __name__ == '__main__':
    sys.exit(main())
