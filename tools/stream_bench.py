#!/usr/bin/env python3
"""BASELINE.json configs[3] as a stream: ViTPose-L / coco_25, 64 persons per 1080p frame, 30 fps target.

Every frame: the (fake) detector's 64 boxes -> this rank's shard of the boxes -> ONE host->device copy of the
frame, crop + pad + resize + normalise + model + decode on device (`vp_infer_frame`), keypoints back, and (N > 1)
one RCCL all-gather of `[64, 25, 3]` so that every rank holds the whole frame's result.  Reports sustained fps and
the per-frame latency distribution (wall clock around the whole frame, PCIe included -- this is the latency a
caller of `VitInference.inference` sees, not the HBM-resident throughput `bench.py` reports).

    python tools/stream_bench.py [--frames 200] [--persons 64] [--variant l] [--dataset coco_25]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/stream_bench.py

One process per GPU; the frame is synthetic (seeded noise with blob "persons"), the weights seeded random.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--persons', type=int, default=64)
    ap.add_argument('--variant', default='l')
    ap.add_argument('--dataset', default='coco_25')
    ap.add_argument('--dtype', default='fp16')
    ap.add_argument('--height', type=int, default=1080)
    ap.add_argument('--width', type=int, default=1920)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from easy_vitpose_amd.configs import model_shape
    from easy_vitpose_amd.cropprep import crop_params
    from easy_vitpose_amd.engine import VitPoseHip
    from easy_vitpose_amd.parallel import shard_bounds
    from easy_vitpose_amd.synth import synthetic_state_dict

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    shp = model_shape(args.variant, args.dataset)
    K = shp.num_keypoints
    per = -(-args.persons // world)
    eng = VitPoseHip(shp, synthetic_state_dict(shp, 0), dtype=args.dtype, device_id=local, max_batch=max(per, 1))

    rng = np.random.default_rng(0)
    frame = rng.integers(0, 256, size=(args.height, args.width, 3), dtype=np.uint8)
    # 64 person boxes on a grid with jittered sizes (what a detector would hand over: x1, y1, x2, y2)
    cols = int(np.ceil(np.sqrt(args.persons * args.width / args.height)))
    rows = -(-args.persons // cols)
    boxes = []
    for i in range(args.persons):
        cx = (i % cols + 0.5) * args.width / cols
        cy = (i // cols + 0.5) * args.height / rows
        w = rng.uniform(0.5, 0.9) * args.width / cols
        h = rng.uniform(0.6, 0.95) * args.height / rows
        boxes.append([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2])
    boxes = np.asarray(boxes).round().astype(int)
    params = crop_params(boxes, frame.shape[:2], 10)
    lo, hi = shard_bounds(args.persons, world, rank)
    my = params[lo:hi]
    d_local = torch.zeros((per, K, 3), dtype=torch.float32, device=dev)
    d_all = torch.empty((world * per, K, 3), dtype=torch.float32, device=dev)

    def one_frame():
        kp = eng.infer_frame(frame, my)                       # H2D frame, device crop prep + model + decode, D2H
        if world > 1:
            d_local[:hi - lo].copy_(torch.from_numpy(kp))
            dist.all_gather_into_tensor(d_all, d_local)       # every rank ends up with the whole frame
            return d_all[:args.persons]
        return kp

    for _ in range(args.warmup):
        one_frame()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    lat = []
    t_start = time.perf_counter()
    for _ in range(args.frames):
        t0 = time.perf_counter()
        out = one_frame()
        if world > 1:
            torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    total = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([total], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        total = float(t.item())
    lat = np.asarray(lat) * 1e3
    if rank == 0:
        fps = args.frames / total
        print(json.dumps({
            'workload': f'ViTPose-{args.variant.upper()} {args.dataset} K={K}, {args.persons} persons per {args.width}x{args.height} frame, '
                        f'{per} crops per GPU, frame H2D + device crop prep + model + decode + D2H'
                        + (' + RCCL all-gather' if world > 1 else ''),
            'n_gpus': world, 'frames': args.frames, 'fps': round(fps, 1), 'persons_per_sec': round(fps * args.persons, 1),
            'target_fps': 30, 'meets_target': bool(fps >= 30),
            'latency_ms': {'p10': round(float(np.percentile(lat, 10)), 3), 'p50': round(float(np.percentile(lat, 50)), 3),
                           'p90': round(float(np.percentile(lat, 90)), 3), 'max': round(float(lat.max()), 3)},
            'dtype': args.dtype, 'data': 'synthetic'}))
        assert np.isfinite(np.asarray(out.cpu() if hasattr(out, 'cpu') else out)).all()
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
