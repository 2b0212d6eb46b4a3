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
from fake_bench_engine import FakeEngine, frame_crops as _frame_crops   # noqa: E402  (shared with the VP_BENCH_ENGINE hook of bench.main)


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


def _order_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        log = FakeEngine.LOG
        real_ag, real_bar = dist.all_gather_into_tensor, dist.barrier

        def ag(*a, **k):
            log.append('all_gather')
            return real_ag(*a, **k)

        def bar(*a, **k):
            log.append('barrier')
            return real_bar(*a, **k)
        dist.all_gather_into_tensor, dist.barrier = ag, bar      # ShardedPose and Harness both call through the module
        B, K = 3, FakeEngine.K
        H = bench.Harness(FakeEngine(), torch.zeros((B, 6, 4, 3), dtype=torch.uint8), torch.zeros((B, K, 3)), torch.zeros((world * B, K, 3)), dist,
                          device_sync=lambda: log.append('device_sync'))
        del log[:]
        for _ in range(3):
            H.step()
        weak = list(log)
        del log[:]
        frame = _frame_crops(7)
        bench.strong_scaling_config4(world, rank, torch.device('cpu'), 'fp16', steps=3, warmup=0, n_total=7,
                                     engine_factory=lambda per, lo, hi: (FakeEngine(), frame[lo:hi].clone(), FakeEngine.K), dist=dist,
                                     device_sync=lambda: log.append('device_sync'))
        q.put((rank, weak, list(log)))
    finally:
        dist.destroy_process_group()


def test_distributed_step_never_blocks_the_host_between_inference_and_collective():
    """the all-gather is enqueued behind the keypoints on the device's stream -- no host synchronisation between the
    inference call and the collective (bench.py's distributed step used to pass sync=True).  The fake engine and the
    patched collectives record the host-side call order at world size 2: inside a weak-scaling step and inside a strong-scaled frame
    every inference call is un-synchronised and is followed DIRECTLY by the all-gather; the host blocks only in the fences."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_order_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    blocking = ('synchronize', 'device_sync', 'barrier')
    for rank, weak, strong in res:
        assert weak == [('infer', False), 'all_gather'] * 3, f'rank {rank}: weak-scaling step order {weak}'
        # the strong-scaled frame loop: 3 frames of [infer, all_gather] back to back, then the closing fence and the timing exchange
        fence = ['synchronize', 'device_sync', 'barrier']
        assert strong[:3] == fence and strong[9:12] == fence, f'rank {rank}: opening / closing fence {strong}'
        frames = strong[3:9]
        assert frames == [('infer', False), 'all_gather'] * 3, f'rank {rank}: strong-scaling frame order {strong}'
        assert not any(e in blocking for e in frames), f'rank {rank}: the host blocked inside the frame loop: {strong}'


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


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset -- the form the driver uses for N = 1 -- must spawn its two ranks itself
    (torch.distributed.run on 127.0.0.1) and still print exactly ONE JSON line on stdout, from rank 0 (it
    used to die on an assert).  The ranks run bench.main() end to end over gloo with the fake engine (VP_BENCH_ENGINE hook): argument
    hand-through, process-group set-up, weak-scaling Harness, the strong-scaled frame, the JSON assembly."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['VP_BENCH_ENGINE'] = 'fake_bench_engine:make'
    env['PYTHONPATH'] = os.path.join(root, 'tests') + os.pathsep + root + os.pathsep + env.get('PYTHONPATH', '')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1'],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f'stdout must hold exactly one JSON line, got {len(lines)}: {r.stdout[:500]}'
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == 3 and j['warmup'] == 1 and j['scaling'] == 'weak'
    assert j['config']['global_batch'] == 2 * 3 and len(j['per_rank_ms_per_step']) == 2 and j['allgather_ms'] > 0
    assert j['value'] > 0 and j['cpu_baseline'] is None
    sc = j['strong_scaling_config4']
    assert sc['scaling'] == 'strong' and sc['crops_per_rank'] == 4 and len(sc['per_rank_ms_per_frame']) == 2
    assert 'launching' in r.stderr      # the launcher path ran (not an externally provided WORLD_SIZE)
