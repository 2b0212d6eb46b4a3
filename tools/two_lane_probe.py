#!/usr/bin/env python3
"""Can two half-batches on two streams fill each other's low-power phases?  Two handles (own stream each) of max_batch 128 run the
whole path concurrently from two threads; with VP_G8_WGS=128 (tools build) every 8-phase launch takes 128 workgroups, so the two
handles' persistent kernels fit on the chip side by side.  Aggregate persons/s against ONE handle at batch 256.
    python tools/two_lane_probe.py            (run under VP_G8_WGS=128 and without)"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402  (device memory)
import _toolslib  # noqa: F401,E402
from easy_vitpose_amd.configs import model_shape  # noqa: E402
from easy_vitpose_amd.engine import VitPoseHip  # noqa: E402
from easy_vitpose_amd.synth import synthetic_state_dict  # noqa: E402

shp = model_shape('b', 'coco')
sd = synthetic_state_dict(shp, seed=1)
dev = torch.device('cuda:0')
steps = int(os.environ.get('STEPS', '40'))


def run(nh, batch, offset_s=0.0):
    engs = [VitPoseHip(shp, sd, dtype='fp16', max_batch=batch) for _ in range(nh)]
    crops = [torch.rand((batch, 3, 256, 192), device=dev) for _ in range(nh)]
    outs = [torch.zeros((batch, shp.num_keypoints, 3), device=dev) for _ in range(nh)]
    torch.cuda.synchronize()
    for e, c, o in zip(engs, crops, outs):
        for _ in range(3):
            e.infer_device(c, o, sync=True)

    def work(i):
        if i:
            time.sleep(offset_s)
        for _ in range(steps):
            engs[i].infer_device(crops[i], outs[i], sync=False, ordered=False)
        engs[i].synchronize()
    th = [threading.Thread(target=work, args=(i,)) for i in range(nh)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    for e in engs:
        e.close()
    return nh * batch * steps / dt


print(f'VP_G8_WGS={os.environ.get("VP_G8_WGS", "-")}:', flush=True)
print(f'  1 handle  x 256 crops: {run(1, 256):9.0f} persons/s', flush=True)
print(f'  1 handle  x 128 crops: {run(1, 128):9.0f} persons/s', flush=True)
print(f'  2 handles x 128 crops: {run(2, 128):9.0f} persons/s', flush=True)
print(f'  2 handles x 128 crops, second one starts 0.4 ms later: {run(2, 128, 0.0004):9.0f} persons/s', flush=True)
