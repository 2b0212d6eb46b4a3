"""CPU: the N>1 path (crop sharding + all-gather of keypoints) with world_size 2 over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from easy_vitpose_amd.parallel import ShardedPose, shard_bounds

K = 5


def fake_infer(crops, org_wh):
    """Deterministic per-crop 'keypoints' (a stand-in for VitPoseHip.infer on the CPU box)."""
    crops = np.asarray(crops, dtype=np.float64)
    out = np.zeros((len(crops), K, 3), np.float32)
    for i, c in enumerate(crops):
        s = c.sum()
        w = 1.0 if org_wh is None else float(org_wh[i][0])
        out[i] = (np.arange(K * 3).reshape(K, 3) + s * 1e-3 + w).astype(np.float32)
    return out


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(123)
        crops = rng.integers(0, 255, size=(n, 4, 3, 3)).astype(np.uint8)
        wh = rng.integers(10, 500, size=(n, 2)).astype(np.int32)
        sp = ShardedPose(fake_infer, K, device='cpu')
        full = sp.infer(crops, wh).numpy()
        lo, hi = shard_bounds(n, world, rank)
        pre = sp.infer(crops[lo:hi], wh[lo:hi], n_total=n, pre_sharded=True).numpy()
        q.put((rank, full, pre))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n', [0, 1, 2, 7, 64])
def test_sharded_equals_single_process(n):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(123)
    crops = rng.integers(0, 255, size=(n, 4, 3, 3)).astype(np.uint8)
    wh = rng.integers(10, 500, size=(n, 2)).astype(np.int32)
    ref = fake_infer(crops, wh)
    for rank, full, pre in res:
        assert full.shape == (n, K, 3)
        assert np.array_equal(full, ref), f'rank {rank}: gathered result differs from the single-process result'
        assert np.array_equal(pre, ref)
