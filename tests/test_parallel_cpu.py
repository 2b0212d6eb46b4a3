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


@pytest.mark.parametrize('n,world', [(0, 2), (1, 2), (2, 2), (7, 2), (64, 2), (7, 3), (2, 3), (64, 3)])
def test_sharded_equals_single_process(n, world):
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


# ------------------------------------------------------------------ bench.py's own multi-rank code path, with a fake engine
class FakeEngine:
    """What bench.Harness / bench.strong_scaling_config4 need from VitPoseHip: infer_device(crops, out, sync), synchronize(),
    close() -- on CPU tensors; 'keypoints' are a deterministic function of each crop, so sharded == unsharded is checkable."""
    K = 5

    def __init__(self):
        self.calls = 0

    @staticmethod
    def expected(crops: torch.Tensor) -> torch.Tensor:
        s = crops.reshape(len(crops), -1).double().sum(1)
        base = torch.arange(FakeEngine.K * 3, dtype=torch.float64).reshape(1, FakeEngine.K, 3)
        return (base + s.reshape(-1, 1, 1) * 1e-3).float()

    def infer_device(self, d_crops, d_out, sync=True):
        self.calls += 1
        d_out.copy_(self.expected(d_crops))
        return d_out

    def synchronize(self):
        pass

    def close(self):
        pass


def _frame_crops(n):
    return torch.from_numpy(np.random.default_rng(9).integers(0, 255, size=(n, 6, 4, 3)).astype(np.uint8))


def _bench_worker(rank, world, port, n_frame, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dev = torch.device('cpu')
        # (1) the weak-scaling step of main(): Harness.step / fence / timed / check_gathered / allgather_ms
        B, K = 3, FakeEngine.K
        eng = FakeEngine()
        crops = torch.from_numpy(np.random.default_rng(100 + rank).integers(0, 255, size=(B, 6, 4, 3)).astype(np.uint8))
        d_out = torch.zeros((B, K, 3))
        d_all = torch.zeros((world * B, K, 3))
        H = bench.Harness(eng, crops, d_out, d_all, dist)
        dt, per_rank = H.timed(4)
        H.check_gathered(B)
        assert eng.calls == 4 and len(per_rank) == world and abs(max(per_rank) - dt) < 1e-12 and H.allgather_ms(2) > 0
        weak = d_all.clone()
        # (2) the strong-scaled frame of BASELINE configs[3]: ShardedPose(pre_sharded) with an uneven tail
        frame = _frame_crops(n_frame)

        def factory(per, lo, hi):
            return FakeEngine(), frame[lo:hi].clone(), FakeEngine.K
        res = bench.strong_scaling_config4(world, rank, dev, 'fp16', steps=3, warmup=1, n_total=n_frame, engine_factory=factory, dist=dist)
        q.put((rank, weak.numpy(), res['keypoints'].clone().numpy(), res['crops_per_rank'], res['per_rank_ms_per_frame']))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_frame', [7, 64, 1])
def test_bench_multi_rank_code_path_with_fake_engine(n_frame):
    """bench.py's N > 1 code -- the weak-scaling Harness (step = local inference + all-gather, barrier-fenced timed loop, max over
    ranks, per-rank times) and strong_scaling_config4 (ShardedPose with pre-sharded crops, uneven tail) -- at world size 2 over
    gloo with an injected fake engine: the 8-GPU path is exercised before 8-GPU hardware shows up."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, n_frame, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    weak_ref = torch.cat([FakeEngine.expected(torch.from_numpy(np.random.default_rng(100 + r).integers(0, 255, size=(3, 6, 4, 3)).astype(np.uint8)))
                          for r in range(world)]).numpy()
    frame_ref = FakeEngine.expected(_frame_crops(n_frame)).numpy()
    for rank, weak, kp, per, per_rank_ms in res:
        assert np.array_equal(weak, weak_ref), f'rank {rank}: weak-scaling all-gather'
        assert kp.shape == (n_frame, FakeEngine.K, 3) and np.array_equal(kp, frame_ref), f'rank {rank}: strong-scaled frame'
        assert per == -(-n_frame // world) and len(per_rank_ms) == world
