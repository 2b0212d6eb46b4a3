"""A stand-in for VitPoseHip on CPU tensors (tests only): what bench.Harness / bench.strong_scaling_config4 / bench.main need
from the engine -- infer_device, synchronize, the profiling calls, close.  'Keypoints' are a deterministic function of each crop,
so sharded == unsharded is checkable.  `make` is the VP_BENCH_ENGINE hook bench.main() imports in every rank of a CPU run."""
import numpy as np
import torch

FAMS = ['gemm_proj', 'gemm_fc1', 'gemm_qkv', 'gemm_patch', 'gemm_deconv', 'gemm_final', 'attention', 'layernorm', 'im2col', 'decode', 'gemm_fc2']


class FakeEngine:
    K = 5

    LOG = []          # per process: ('infer', sync) / 'synchronize' in call order, shared by every FakeEngine of the process (the
                      # host-sync test of tests/test_parallel_cpu.py interleaves it with the collectives)

    def __init__(self):
        self.calls = 0

    @staticmethod
    def expected(crops: torch.Tensor) -> torch.Tensor:
        s = crops.reshape(len(crops), -1).double().sum(1)
        base = torch.arange(FakeEngine.K * 3, dtype=torch.float64).reshape(1, FakeEngine.K, 3)
        return (base + s.reshape(-1, 1, 1) * 1e-3).float()

    def infer_device(self, d_crops, d_out, sync=True):
        self.calls += 1
        FakeEngine.LOG.append(('infer', bool(sync)))
        d_out.copy_(self.expected(d_crops))
        return d_out

    def synchronize(self):
        FakeEngine.LOG.append('synchronize')

    def close(self):
        pass

    # profiling surface of VitPoseHip (all zeros: nothing is timed on the CPU)
    def set_profiling(self, families=True):
        pass

    def reset_profile(self):
        pass

    def profile(self):
        return {f: dict(ms=0.0, flops=0.0, bytes=0.0, launches=0) for f in FAMS}

    def profile_kernel(self, family):
        return 'fake'


def frame_crops(n):
    return torch.from_numpy(np.random.default_rng(9).integers(0, 255, size=(n, 6, 4, 3)).astype(np.uint8))


def rank_crops(rank, B=3):
    return torch.from_numpy(np.random.default_rng(100 + rank).integers(0, 255, size=(B, 6, 4, 3)).astype(np.uint8))


def make(args, rank, world, local_rank):
    """VP_BENCH_ENGINE hook: bench.main() on a CPU box over gloo."""
    B = 3
    frame = frame_crops(7)

    def strong_factory(per, lo, hi):
        return FakeEngine(), frame[lo:hi].clone(), FakeEngine.K
    return dict(eng=FakeEngine(), d_crops=rank_crops(rank, B), d_out=torch.zeros((B, FakeEngine.K, 3)), K=FakeEngine.K,
                dev=torch.device('cpu'), backend='gloo', device_sync=lambda: None, strong_factory=strong_factory)
