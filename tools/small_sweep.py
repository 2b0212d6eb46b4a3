#!/usr/bin/env python3
"""Small batches IN SITU (one GPU's share of a sharded frame: BASELINE configs[3] = ViTPose-L, 8 crops): ms per step of the whole hot path (hipGraph replay) and
us per launch of the four encoder GEMM families for tile-configuration overrides (VP_GEMM_TUNE, measurement build), each checked for bit-identical keypoints against
the default rule.  The isolated sweeps of rounds 2-3 (tools/gemm_small.py) ran with L2-resident weights; inside the step every layer's weights come from HBM.
Round 6: a set may carry a split-K override of the residual GEMMs behind a '|' (VP_SPLITK, e.g. `sk=|fc2:4:20,proj:2:12`; `off=|0`); the family columns are us per LAYER
(a split-K family is two launches: partial products + reduction).
GPU box only.      python tools/small_sweep.py [--cases l:coco_25:8,...] [--sets name=tune[|splitk];name=tune...]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
from easy_vitpose_amd.configs import model_shape  # noqa: E402
from easy_vitpose_amd.engine import VitPoseHip  # noqa: E402
from easy_vitpose_amd.synth import synthetic_crops, synthetic_state_dict  # noqa: E402

PROJ, FC1, QKV, FC2 = 0, 1, 2, 10
DEFAULT_SETS = [
    ('default', ''),
    ('wide=19', f'{QKV}:19:0,{FC1}:19:0'),
    ('wide=20', f'{QKV}:20:0,{FC1}:20:0'),
    ('wide=24', f'{QKV}:24:0,{FC1}:24:0'),
    ('wide=13', f'{QKV}:13:0,{FC1}:13:0'),
    ('proj=12', f'{PROJ}:12:0'),
    ('proj=21', f'{PROJ}:21:0'),
    ('proj=23', f'{PROJ}:23:0'),
    ('fc2=21', f'{FC2}:21:0'),
    ('fc2=22', f'{FC2}:22:0'),
    ('fc2=23', f'{FC2}:23:0'),
    ('resid=22', f'{PROJ}:22:0,{FC2}:22:0'),
]

ap = argparse.ArgumentParser()
ap.add_argument('--cases', default='l:coco_25:8')
ap.add_argument('--sets', default='')
ap.add_argument('--iters', type=int, default=100)
args = ap.parse_args()
sets = DEFAULT_SETS if not args.sets else [tuple(x.split('=', 1)) if '=' in x else (x, '') for x in args.sets.split(';')]
sets = [(n, t.replace('PROJ', str(PROJ)).replace('FC1', str(FC1)).replace('QKV', str(QKV)).replace('FC2', str(FC2))) for n, t in sets]

for case in args.cases.split(','):
    variant, dataset, n = case.split(':')
    n = int(n)
    shp = model_shape(variant, dataset)
    sd = synthetic_state_dict(shp, 0)
    crops = torch.from_numpy(np.ascontiguousarray(synthetic_crops(n, 0, 'noise'))).cuda()
    out = torch.empty((n, shp.num_keypoints, 3), dtype=torch.float32, device='cuda')
    print(f'# ViTPose-{variant.upper()} / {dataset}, {n} crops (M = {192 * n}, D = {shp.embed_dim}): ms per step (hipGraph replay, {args.iters} calls) | us per layer: qkv fc1 proj fc2 attention | kernels | keypoints vs default', flush=True)
    ref = None
    for name, tune in sets:
        tune, _, sk = tune.partition('|')
        if sk:
            os.environ['VP_SPLITK'] = sk
        else:
            os.environ.pop('VP_SPLITK', None)
        if tune:
            os.environ['VP_GEMM_TUNE'] = tune
        else:
            os.environ.pop('VP_GEMM_TUNE', None)
        try:
            eng = VitPoseHip(shp, sd, 'fp16', 0, max_batch=n)
            for _ in range(5):
                eng.infer_device(crops, out, sync=True, ordered=False)
            kp = out.cpu().numpy().copy()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                eng.infer_device(crops, out, sync=False, ordered=False)
            eng.synchronize()
            ms = (time.perf_counter() - t0) / args.iters * 1e3
            eng.set_profiling(True)
            eng.reset_profile()
            for _ in range(10):
                eng.infer_device(crops, out, sync=True, ordered=False)
            p = eng.profile()
            us = {f: 1e3 * p[f]['ms'] / (10 * shp.depth) for f in ('gemm_qkv', 'gemm_fc1', 'gemm_proj', 'gemm_fc2', 'attention')}
            kern = ' | '.join(eng.profile_kernel(f).split('TileCfg')[-1] for f in ('gemm_qkv', 'gemm_fc1', 'gemm_proj', 'gemm_fc2'))
            eng.close()
        except Exception as e:   # a configuration that does not launch for this shape
            print(f'{name:10s} failed: {str(e)[:160]}', flush=True)
            continue
        if ref is None:
            ref = kp
        same = 'identical' if np.array_equal(kp, ref) else f'differs from the first set (max {np.abs(kp - ref).max():.3g})'
        print(f'{name:10s} {ms:7.3f} ms | ' + ' '.join(f'{us[f]:6.1f}' for f in us) + f' | {kern} | {same}', flush=True)
