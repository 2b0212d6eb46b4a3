"""Wire format of the reference's ``--save-json`` output (top-level ``inference.py:136-141``):

    {"keypoints": [ {person_id: [[y, x, score], ...K rows]}, ... one dict per frame ],
     "skeleton":  {joint_index: joint_name}}

``VitInference.inference(frame)`` returns exactly one such per-frame dict.  Host-side only; the joint
name table of the reference (``vit_utils/visualization.py::joints_dict``) is dataset metadata the caller passes in --
only the 17 standard COCO names are built in.
"""
from __future__ import annotations

import json
from typing import Mapping, Optional, Sequence

import numpy as np

COCO17_JOINTS = {0: 'nose', 1: 'left_eye', 2: 'right_eye', 3: 'left_ear', 4: 'right_ear', 5: 'left_shoulder',
                 6: 'right_shoulder', 7: 'left_elbow', 8: 'right_elbow', 9: 'left_wrist', 10: 'right_wrist',
                 11: 'left_hip', 12: 'right_hip', 13: 'left_knee', 14: 'right_knee', 15: 'left_ankle', 16: 'right_ankle'}


class _NumpyEncoder(json.JSONEncoder):
    def default(self, o):
        if isinstance(o, np.ndarray):
            return o.tolist()
        if isinstance(o, np.generic):
            return o.item()
        return super().default(o)


def frames_to_json(frames: Sequence[Mapping], joint_names: Optional[Mapping[int, str]] = None) -> str:
    """``frames``: one ``{person_id: ndarray[K, 3] (y, x, score)}`` per frame, as returned by ``VitInference.inference``."""
    out = {'keypoints': [{str(k) if not isinstance(k, (str, int)) else k: v for k, v in f.items()} for f in frames],
           'skeleton': dict(joint_names) if joint_names is not None else {}}
    return json.dumps(out, cls=_NumpyEncoder)


def save_json(path: str, frames: Sequence[Mapping], joint_names: Optional[Mapping[int, str]] = None) -> None:
    with open(path, 'w') as f:
        f.write(frames_to_json(frames, joint_names))
