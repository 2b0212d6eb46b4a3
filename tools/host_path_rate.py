#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (`vp_infer`): H2D of the crop batch + compute + D2H of the
keypoints, from pageable numpy memory.  Never the `value` of bench.py (that one starts with the crops in HBM).  GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easy_vitpose_amd.configs import model_shape
from easy_vitpose_amd.engine import VitPoseHip
from easy_vitpose_amd.inference import MEAN, STD
from easy_vitpose_amd.synth import synthetic_crops, synthetic_state_dict

shp = model_shape('b', 'coco')
eng = VitPoseHip(shp, synthetic_state_dict(shp, 0), dtype='fp16', max_batch=256)
u8 = synthetic_crops(256, 0, 'noise')
f32 = np.ascontiguousarray((((u8 / 255.0) - MEAN) / STD).transpose(0, 3, 1, 2).astype(np.float32))
for name, x in (('u8 NHWC host buffer', u8), ('f32 NCHW host buffer', f32)):
    for _ in range(3):
        eng.infer(x)
    t = time.perf_counter()
    n = 10
    for _ in range(n):
        eng.infer(x)
    dt = (time.perf_counter() - t) / n
    print(f'{name}: {256 / dt:.0f} persons/s, {dt * 1e3:.2f} ms per 256 crops (H2D + compute + D2H, pageable numpy memory)')
eng.close()
