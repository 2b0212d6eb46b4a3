"""Crop sharding across the GPUs of one node + all-gather of keypoints.

The path shards trivially: crops are independent, weights are
replicated, and the single exchange is an all-gather of ``[n_local, K, 3]`` float32
keypoints (2.4 KB per rank for 8 crops x 25 joints: latency-bound, xGMI bandwidth is
irrelevant).  One process per GPU; ``torch.distributed`` (backend ``nccl`` = RCCL over
xGMI on ROCm, ``gloo`` in the CPU tests) is plumbing only.

The reference has no inference-side parallelism (one model, one device,
easy_ViTPose/inference.py:167, crops looped at :259-272); this is the build's own
addition required by BASELINE.json's north_star.
"""
from __future__ import annotations

from typing import Callable

import numpy as np


def shard_bounds(n: int, world: int, rank: int) -> "tuple[int, int]":
    """Contiguous split of n crops: rank r takes [r*ceil(n/world), ...) clipped to n."""
    per = -(-n // world) if n > 0 else 0
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


class ShardedPose:
    """Run ``infer_local`` on this rank's shard of the crops and all-gather the keypoints.

    ``infer_local(crops_shard, org_wh_shard) -> ndarray/Tensor [n_local, K, 3]``; on GPU it
    is ``VitPoseHip.infer`` (or ``infer_device``).  Every rank passes the same full batch
    (or only its own shard with ``pre_sharded=True``) and receives the full ``[N, K, 3]``.
    """

    def __init__(self, infer_local: Callable, num_keypoints: int, device: str = 'cpu', group=None, reuse_buffers: bool = False):
        import torch.distributed as dist
        assert dist.is_initialized(), 'init_process_group first (nccl = RCCL on ROCm, gloo on CPU)'
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.infer_local = infer_local
        self.K = num_keypoints
        self.device = device
        # reuse_buffers: the padded local block and the gathered block are allocated once per shard size and reused by every call (a
        # frame loop then launches no allocation / fill kernels); the returned tensor is a VIEW that the next call overwrites
        self.reuse = reuse_buffers
        self._bufs = {}

    def infer(self, crops, org_wh=None, n_total: int | None = None, pre_sharded: bool = False):
        import torch
        if pre_sharded:
            assert n_total is not None
            n, shard, wh = n_total, crops, org_wh
            lo, hi = shard_bounds(n, self.world, self.rank)
            assert len(shard) == hi - lo, f'rank {self.rank}: shard has {len(shard)} crops, expected {hi - lo}'
        else:
            n = len(crops)
            lo, hi = shard_bounds(n, self.world, self.rank)
            shard = crops[lo:hi]
            wh = None if org_wh is None else org_wh[lo:hi]
        per = -(-n // self.world) if n > 0 else 0
        if per not in self._bufs or not self.reuse:
            self._bufs = {per: (torch.zeros((per, self.K, 3), dtype=torch.float32, device=self.device),
                                torch.empty((self.world * per, self.K, 3), dtype=torch.float32, device=self.device), [0])}
        local, gathered, last = self._bufs[per]
        if hi - lo < last[0]:
            local[hi - lo:].zero_()          # a shorter shard than last time: the padding rows are zeros again
        last[0] = hi - lo
        if hi > lo:
            res = self.infer_local(shard, wh)
            res = res if isinstance(res, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(res))
            local[:hi - lo] = res.to(self.device)
        if per > 0:
            # equal-sized (tail rank zero-padded) contributions -> one fused all-gather
            self.dist.all_gather_into_tensor(gathered, local, group=self.group)
        return gathered[:n]
