#!/usr/bin/env python3
"""Headline benchmark: persons/s of the ViTPose hot path on N MI355X (BASELINE.json).

    python bench.py [--gpus N --steps K --warmup W] [--dtype fp16|bf16] [--batch 256]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (crop batch resident in HBM -> patch embed
-> ViT encoder -> deconv head -> heatmaps -> arg-max/DARK-UDP decode -> keypoints in
HBM, + the RCCL all-gather of keypoints when N > 1) over one batch of 256 synthetic
256x192 crops per GPU (weak scaling: crops are independent, each rank owns its batch).
Workload = BASELINE.json configs[1]: ViTPose-B COCO-17, batch 256, seeded random
weights of that architecture, synthetic uniform-noise crops.

Prints ONE JSON line on rank 0 with `roofline` (the GEMM family with the largest share of the step, timed live
with HIP events on the library's stream inside the timed region), `step_ms` percentiles, `host_persons_per_sec`
(N=1: pinned host buffers -> keypoints on the host through the asynchronous double-buffered entry, PCIe included)
and `cpu_baseline` (the oracle's torch-CPU per-crop path on this box's host cores, rank 0, N=1 only, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_16BIT = 2.5e15   # dense bf16/fp16 MFMA peak, MI355X_MICROARCH.md (256 CU x 4096 FLOP/clk x 2.4 GHz)
PEAK_HBM = 8.0e12


def cpu_baseline(variant, dataset, budget_s=15.0):
    """Reference-equivalent CPU path (oracle/ restatement of _inference_torch, per crop,
    batch 1 exactly like VitInference) on a bounded sample of the same synthetic crops.

    torch's default of one thread per hardware thread is pathological for batch-1 ViT
    GEMMs on a many-core host (measured 0.03 persons/s with 256 threads), so a few
    thread counts are probed on 2 crops each and the best one is used and reported."""
    import torch
    from easy_vitpose_amd.configs import model_shape
    from easy_vitpose_amd.synth import synthetic_crops, synthetic_state_dict
    from oracle import vitpose_cpu as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    shp = model_shape(variant, dataset)
    sd = O.to_torch_state_dict(synthetic_state_dict(shp, 0))
    crops = synthetic_crops(256, 0, 'noise')
    run = lambda i: O.inference_torch(sd, shp.depth, shp.num_heads, crops[i])
    best_t, best_dt = None, None
    for t in sorted({min(avail, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(t)
        run(0)
        t0 = time.perf_counter(); run(1); run(2); dt = (time.perf_counter() - t0) / 2
        if best_dt is None or dt < best_dt:
            best_t, best_dt = t, dt
        if dt > 3.0:
            break
    torch.set_num_threads(best_t)
    n, t0 = 0, time.perf_counter()
    while n < len(crops) and (time.perf_counter() - t0 < budget_s or n < 3):
        run(n)
        n += 1
    dt = time.perf_counter() - t0
    return {'value': round(n / dt, 3), 'unit': 'persons/s', 'cores': best_t, 'kind': 'port',
            'sample': f'{n} crops of the same workload, one at a time (pre_img -> torch fp32 model -> decode), '
                      f'{dt:.1f} s, torch {torch.__version__} with {best_t} threads (best of 8/16/32/64; host has {avail} hw threads)'}


# GEMM families of the encoder: profiling-family name -> (kernel symbol prefix in the rocprofv3 trace, description)
FAMILIES = {
    'gemm_fc1': ('gemm8_kernel<{T}, 1,', 'mlp.fc1 (+LayerNorm fold +bias +GELU), 8-phase persistent kernel, 256x256 tiles'),
    'gemm_fc2': ('gemm8_kernel<{T}, 6,', 'mlp.fc2 (+bias +residual planes +LayerNorm row statistics), 8-phase kernel, 256x192 tiles'),
    'gemm_qkv': ('gemm8_kernel<{T}, 0,', 'attn.qkv (+LayerNorm fold +bias), 8-phase persistent kernel, 256x256 tiles'),
    'gemm_proj': ('gemm_kernel<{T}, 6, 0', 'attn.proj (+bias +residual planes +LayerNorm row statistics), 192x128 tiles'),
}


def kernel_label(fam, T, B, shp):
    """Name of the kernel a family runs on at this batch size.  The 8-phase kernel takes a GEMM when its row count is a multiple
    of 256 and the launch has >= 448 output tiles (vitpose_api.hip: gemm()); smaller launches run gemm_kernel's tile table."""
    pre, what = FAMILIES[fam]
    if pre.startswith('gemm8'):
        M, D = B * 192, shp.embed_dim
        N = {'gemm_fc1': 4 * D, 'gemm_qkv': 3 * D, 'gemm_fc2': D}[fam]
        best = None                                   # (fill of the last round of 256 workgroups, tiles, tile width): as vitpose_api.hip gemm()
        for bn in ((256, 192) if fam == 'gemm_fc2' else (256,)):
            if N % bn == 0 and M % 256 == 0:
                t = (M // 256) * (N // bn)
                f = t / (-(-t // 256) * 256)
                if best is None or f > best[0] + 1e-9:
                    best = (f, t, bn)
        if best is None or not (best[1] >= 448 or (best[0] >= 0.8 and best[1] >= 192)):
            return f'gemm_kernel<{T}, ...> tile table (shape or launch size outside the set of the 8-phase kernel): ' + what.split(',')[0]
        what = what.replace('256x192', f'256x{best[2]}')
    return pre.format(T=T) + ' ...>: ' + what


def strong_scaling_config4(world, rank, dev, dtype, steps=100, warmup=10):
    """BASELINE.json configs[3]: one frame of 64 crops through ViTPose-L coco_25, strong-scaled: every rank takes 64 / world crops
    (resident in its HBM), RCCL all-gather of the keypoints, max-over-ranks time.  Reported inside the single JSON line as
    `strong_scaling_config4` (the driver computes efficiency from the per-N values)."""
    import torch
    import torch.distributed as dist
    from easy_vitpose_amd import VitPoseHip
    from easy_vitpose_amd.configs import model_shape
    from easy_vitpose_amd.parallel import shard_bounds
    from easy_vitpose_amd.synth import synthetic_crops, synthetic_state_dict
    shp = model_shape('l', 'coco_25')
    N, K = 64, shp.num_keypoints
    lo, hi = shard_bounds(N, world, rank)
    per = -(-N // world)
    eng = VitPoseHip(shp, synthetic_state_dict(shp, 0), dtype=dtype, device_id=dev.index, max_batch=per)
    crops = torch.from_numpy(synthetic_crops(N, seed=4, kind='noise')[lo:hi]).to(dev)
    local = torch.zeros((per, K, 3), dtype=torch.float32, device=dev)
    gathered = torch.zeros((world * per, K, 3), dtype=torch.float32, device=dev)

    def frame():
        if hi > lo:
            eng.infer_device(crops, local[:hi - lo], sync=True)
        dist.all_gather_into_tensor(gathered, local)

    def fence():
        eng.synchronize(); torch.cuda.synchronize(); dist.barrier()

    for _ in range(warmup):
        frame()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        frame()
    fence()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    eng.close()
    dt = float(t.item())
    return {'workload': 'ViTPose-L coco_25, one frame of 64 u8 crops resident in HBM, 64/world per rank, RCCL all-gather of keypoints',
            'scaling': 'strong', 'crops_per_rank': per, 'frames': steps, 'ms_per_frame': round(dt / steps * 1e3, 4),
            'persons_per_sec': round(N * steps / dt, 1)}


def pmc_traffic(args, fam):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes of THIS command (profiles/pmc_r2.json,
    made by tools/profile.sh + tools/summarize_profile.py; FETCH_SIZE x2 on gfx950 as MI355X_MICROARCH.md prescribes).
    PMC passes cannot run inside this process, so the figure is read from the committed profile of the same configuration;
    None when the configuration differs from the profiled one."""
    path = os.path.join(ROOT, 'profiles', 'pmc_r2.json')
    if not (os.path.exists(path) and args.variant == 'b' and args.batch == 256 and args.dtype == 'fp16' and args.gpus == 1):
        return None
    pre = FAMILIES[fam][0].format(T='F16' if args.dtype == 'fp16' else 'BF16')
    try:
        for k, d in json.load(open(path)).items():
            if k.startswith(pre) and 'hbm_bytes_per_dispatch' in d:
                return d['hbm_bytes_per_dispatch']
    except Exception:
        pass
    return None


def host_path_rate(eng, crops_u8, K, seconds=1.5):
    """persons/s of the host-visible path: uint8 crops in pinned host memory -> keypoints in pinned host memory, through
    vp_infer_submit / vp_infer_wait (H2D of batch i+1 and D2H of batch i-1 under the compute of batch i)."""
    from easy_vitpose_amd import PinnedArray
    B = crops_u8.shape[0]
    pin_in = [PinnedArray(crops_u8.shape, np.uint8) for _ in range(2)]
    pin_out = [PinnedArray((B, K, 3), np.float32) for _ in range(2)]
    for p in pin_in:
        p.array[...] = crops_u8
    pending = []
    def push(i):
        pending.append(eng.submit(pin_in[i & 1].array, pin_out[i & 1].array))
    for i in range(3):          # warm
        push(i)
        if len(pending) == 2:
            eng.wait(pending.pop(0))
    while pending:
        eng.wait(pending.pop(0))
    n, t0 = 0, time.perf_counter()
    while True:
        push(n)
        n += 1
        if len(pending) == 2:
            eng.wait(pending.pop(0))
        if time.perf_counter() - t0 > seconds and n >= 20:
            break
    while pending:
        eng.wait(pending.pop(0))
    dt = time.perf_counter() - t0
    assert np.isfinite(pin_out[0].array).all()
    for p in pin_in + pin_out:
        p.free()
    return n * B / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--dtype', default='fp16', choices=['fp16', 'bf16'])
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--max-batch', type=int, default=0, help='workspace batch of the handle (< --batch: the batch is processed in chunks)')
    ap.add_argument('--variant', default='b')
    ap.add_argument('--dataset', default='coco')
    ap.add_argument('--input', default='f32', choices=['f32', 'u8'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-host-path', action='store_true')
    ap.add_argument('--breakdown', action='store_true', help='extra untimed pass with every kernel family timed (stderr)')
    ap.add_argument('--strong', action='store_true', help='also measure the strong-scaled frame of BASELINE configs[3] (default when WORLD_SIZE > 1)')
    ap.add_argument('--force-dist', action='store_true', help='run the RCCL code path (process group, all-gather, barrier) even with one rank')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from easy_vitpose_amd import VitPoseHip
    from easy_vitpose_amd.configs import model_shape
    from easy_vitpose_amd.synth import synthetic_crops, synthetic_state_dict

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    assert torch.cuda.is_available(), 'bench.py needs an AMD GPU (the HIP path has no CPU fallback)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    shp = model_shape(args.variant, args.dataset)
    B, K = args.batch, shp.num_keypoints
    eng = VitPoseHip(shp, synthetic_state_dict(shp, 0), dtype=args.dtype, device_id=local_rank, max_batch=args.max_batch or B)
    crops_u8 = synthetic_crops(B, seed=rank, kind='noise')
    if args.input == 'u8':
        d_crops = torch.from_numpy(crops_u8).to(dev)
    else:  # what pre_img hands to the model: normalised float32 NCHW
        mean = np.array([0.485, 0.456, 0.406]); std = np.array([0.229, 0.224, 0.225])
        x = ((crops_u8.astype(np.float64) / 255 - mean) / std).transpose(0, 3, 1, 2).astype(np.float32)
        d_crops = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_out = torch.zeros((B, K, 3), dtype=torch.float32, device=dev)
    d_all = torch.zeros((world * B, K, 3), dtype=torch.float32, device=dev) if use_dist else None
    torch.cuda.synchronize()

    def step():
        eng.infer_device(d_crops, d_out, sync=use_dist)      # library stream; sync hands over to torch's stream
        if use_dist:
            dist.all_gather_into_tensor(d_all, d_out)        # RCCL over xGMI, [world*B, K, 3]

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    fams = list(FAMILIES)
    eng.set_profiling(fams)      # warm-up pass with all four encoder GEMM families timed: picks the dominant one
    eng.reset_profile()
    for _ in range(args.warmup):
        step()
    fence()
    wprof = eng.profile()
    dom = max(fams, key=lambda f: wprof[f]['ms']) if args.warmup > 0 else 'gemm_fc1'
    # the four encoder GEMM families are timed live (HIP events around every launch, on the library's stream); the one with
    # the largest share of the step is reported as the dominant kernel (rocprofv3 --stats of the same command: profiles/)
    # batches of <= 16 crops replay a captured hipGraph, which event records inside the chunk would switch off: there the timed
    # region runs unprofiled and the dominant kernel's launch time is the warm-up pass's (roofline.timed = 'warm-up pass')
    live = B > 16
    eng.set_profiling([dom] if live else False)     # timed region: only the dominant family carries event records (2 per launch)
    eng.reset_profile()
    if not live:
        for _ in range(3):       # first sighting runs eagerly, the second captures the graph
            step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = eng.profile() if live else wprof
    eng.set_profiling(False)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # every rank holds every rank's keypoints, own shard bit-identical to the local result
        assert torch.equal(d_all[rank * B:(rank + 1) * B], d_out), 'all-gather result differs from the local shard'
    assert torch.isfinite(d_out).all(), 'non-finite keypoints'

    # per-step distribution (separate, untimed-for-`value` pass with a host synchronisation after every step)
    step_ms = []
    for _ in range(min(args.steps, 50)):
        fence()
        t1 = time.perf_counter()
        step()
        fence()
        step_ms.append((time.perf_counter() - t1) * 1e3)
    host_rate = None
    if rank == 0 and world == 1 and not args.no_host_path:
        host_rate = host_path_rate(eng, crops_u8, K)

    strong = None
    if use_dist and (world > 1 or args.strong):
        strong = strong_scaling_config4(world, rank, dev, args.dtype)

    breakdown = None
    if args.breakdown and rank == 0:
        eng.set_profiling(True); eng.reset_profile()
        for _ in range(3):
            eng.infer_device(d_crops, d_out, sync=True)
        p = eng.profile(); eng.set_profiling(False)
        breakdown = {k: {'ms_per_step': round(v['ms'] / 3, 4), 'launches_per_step': v['launches'] // 3,
                         'tflops': round(v['flops'] / max(v['ms'], 1e-9) / 1e9, 1),
                         'gbps': round(v['bytes'] / max(v['ms'], 1e-9) / 1e6, 1)} for k, v in p.items()}
        print(json.dumps({'breakdown': breakdown}), file=sys.stderr)

    if rank == 0:
        persons_s = world * B * args.steps / dt
        d = prof[dom]
        ach = d['flops'] / (d['ms'] * 1e-3) if d['ms'] > 0 else 0.0
        T = 'F16' if args.dtype == 'fp16' else 'BF16'
        per_family = {f: {'ms_per_step': round(wprof[f]['ms'] / max(args.warmup, 1), 4), 'avg_launch_us': round(1e3 * wprof[f]['ms'] / max(wprof[f]['launches'], 1), 2),
                          'tflops': round(wprof[f]['flops'] / max(wprof[f]['ms'], 1e-9) / 1e9, 1),
                          'algorithmic_gbps': round(wprof[f]['bytes'] / max(wprof[f]['ms'], 1e-9) / 1e6, 1)} for f in fams}   # from the warm-up pass
        line = {
            'metric': 'persons_per_sec', 'value': round(persons_s, 1), 'unit': 'persons/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'ViTPose-{args.variant.upper()} {args.dataset} K={K}, batch {B} x 256x192 crops per GPU '
                                   f'({args.input} input resident in HBM) -> (y,x,conf) keypoints in HBM'
                                   + (', RCCL all-gather of keypoints' if use_dist else ''),
                       'global_batch': world * B, 'weights': 'seeded random init (no checkpoint offline)',
                       'gflop_per_person': round(shp.gflop_per_person(), 3)},
            'model_tflops': round(persons_s * shp.gflop_per_person() / 1e3, 1),
            'model_frac_of_mfma_peak': round(persons_s * shp.gflop_per_person() * 1e9 / PEAK_MFMA_16BIT, 4),
            'roofline': {'bound': 'mfma', 'kernel': kernel_label(dom, T, B, shp),
                         'achieved': round(ach / 1e12, 2), 'peak': PEAK_MFMA_16BIT / 1e12, 'unit': 'TFLOP/s',
                         'frac': round(ach / PEAK_MFMA_16BIT, 4), 'traffic': pmc_traffic(args, dom),
                         'timed': 'live, HIP events around every launch of the timed region' if B > 16 else 'warm-up pass (the timed region replays a hipGraph)',
                         'launches': d['launches'], 'avg_launch_ms': round(d['ms'] / max(d['launches'], 1), 5),
                         'flops_per_launch': d['flops'] / max(d['launches'], 1),
                         'algorithmic_bytes_per_launch': d['bytes'] / max(d['launches'], 1)},
            'encoder_gemms': per_family,
            'step_ms': {'p10': round(float(np.percentile(step_ms, 10)), 4), 'p50': round(float(np.percentile(step_ms, 50)), 4),
                        'p90': round(float(np.percentile(step_ms, 90)), 4), 'n': len(step_ms), 'note': 'one host synchronisation per step'},
            'host_persons_per_sec': None if host_rate is None else round(host_rate, 1),
        }
        if strong is not None:
            line['strong_scaling_config4'] = strong
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.variant, args.dataset)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line), flush=True)
    eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
