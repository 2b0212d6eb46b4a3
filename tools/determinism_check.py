#!/usr/bin/env python3
"""Run-to-run determinism of the whole path on a FRESH handle (first forward on uninitialised workspaces vs later ones): any
difference is a read-before-write or a race.  GPU box only.   python tools/determinism_check.py [variant batch reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402  (measurement build of the library)
from easy_vitpose_amd.configs import model_shape
from easy_vitpose_amd.engine import VitPoseHip
from easy_vitpose_amd.synth import synthetic_crops, synthetic_state_dict

variant = sys.argv[1] if len(sys.argv) > 1 else 's'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
shp = model_shape(variant, 'coco')
sd = synthetic_state_dict(shp, 0)
crops = np.concatenate([synthetic_crops(batch // 4, 7, 'blobs'), synthetic_crops(batch - batch // 4, 8, 'noise')])
bad = 0
for trial in range(reps):
    eng = VitPoseHip(shp, sd, dtype='fp16', max_batch=batch)
    tok = [eng.tokens(crops) for _ in range(2)]
    hm = [eng.heatmaps(crops) for _ in range(3)]
    kp = eng.infer(crops)
    eng.close()
    dt = np.abs(tok[0] - tok[1]).max()
    d01, d12 = np.abs(hm[0] - hm[1]).max(), np.abs(hm[1] - hm[2]).max()
    dk = np.abs(kp[..., 2] - hm[2].reshape(batch, -1, 3072).max(-1)).max()
    rows = sorted(set(np.argwhere(np.abs(hm[0] - hm[1]).reshape(batch, -1).max(1) > 0).ravel().tolist()))
    print(f'trial {trial}: tokens run0-vs-1 {dt:.3e}  heatmaps 0-vs-1 {d01:.3e} 1-vs-2 {d12:.3e}  infer-vs-heatmaps {dk:.3e}  crops differing {rows}', flush=True)
    bad += (dt > 0) + (d01 > 0) + (d12 > 0) + (dk > 0)
print('DETERMINISTIC' if bad == 0 else f'NONDETERMINISTIC ({bad} differing comparisons)')
sys.exit(0 if bad == 0 else 1)
