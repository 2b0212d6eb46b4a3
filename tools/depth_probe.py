#!/usr/bin/env python3
"""Small batches: what do COLD weights cost?  The same model at depths 24 / 8 / 4 / 2 / 1 (ViTPose-L geometry, 8 crops by default): with few blocks all weights stay in the 256 MB
memory-side cache (and partly in the L2s) from one step to the next; at depth 24 (600 MB) every layer's weights come from HBM.  ms per step and us per LAYER (step minus the
depth-0 intercept of a linear fit) -- if the per-layer time falls with the depth, prefetching the next layer's weights would pay; if it does not, the small-batch GEMMs are
bound by something else (round 6: it does not).      GPU box only.      python tools/depth_probe.py [--variant l --dataset coco_25 --crops 8]"""
import argparse
import dataclasses
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from easy_vitpose_amd.configs import model_shape  # noqa: E402
from easy_vitpose_amd.engine import VitPoseHip  # noqa: E402
from easy_vitpose_amd.synth import synthetic_crops, synthetic_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--variant', default='l')
ap.add_argument('--dataset', default='coco_25')
ap.add_argument('--crops', type=int, default=8)
ap.add_argument('--iters', type=int, default=200)
args = ap.parse_args()
base = model_shape(args.variant, args.dataset)
crops = torch.from_numpy(np.ascontiguousarray(synthetic_crops(args.crops, 0, 'noise'))).cuda()
out = torch.empty((args.crops, base.num_keypoints, 3), dtype=torch.float32, device='cuda')
rows = []
for depth in (base.depth, 12, 8, 4, 2, 1):
    shp = dataclasses.replace(base, depth=depth)
    eng = VitPoseHip(shp, synthetic_state_dict(shp, 0), 'fp16', 0, max_batch=args.crops)
    for _ in range(10):
        eng.infer_device(crops, out, sync=True, ordered=False)
    t0 = time.perf_counter()
    for _ in range(args.iters):
        eng.infer_device(crops, out, sync=False, ordered=False)
    eng.synchronize()
    ms = (time.perf_counter() - t0) / args.iters * 1e3
    eng.close()
    mb = depth * 12 * shp.embed_dim * shp.embed_dim * 2 / 1e6
    rows.append((depth, ms, mb))
    print(f'depth {depth:3d}: {ms:7.3f} ms per step, encoder weights {mb:6.0f} MB', flush=True)
# intercept from the two shallowest models
(d1, m1, _), (d2, m2, _) = rows[-1], rows[-2]
per = (m2 - m1) / (d2 - d1)
icpt = m1 - per * d1
print(f'intercept (patch embed + head + decode) {icpt:.3f} ms; us per layer: ' + ', '.join(f'depth {d}: {1e3 * (m - icpt) / d:.1f}' for d, m, _ in rows))
