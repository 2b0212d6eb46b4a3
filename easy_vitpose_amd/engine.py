"""Python host object over the C ABI: one ``VitPoseHip`` = one model on one GPU.

The C library owns the weights, workspaces and its HIP stream; numpy / torch are
only used to hand it pointers.  There is no CPU implementation behind this class.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as capi
from .configs import HM_H, HM_W, IMG_H, IMG_W, ModelShape


def _as_f32_numpy(v) -> np.ndarray:
    if hasattr(v, 'detach'):  # torch tensor (bf16 / fp16 checkpoints have no numpy dtype: widen first)
        v = v.detach().float().cpu().numpy()
    return np.ascontiguousarray(np.asarray(v), dtype=np.float32)


def _tensor_descs(state_dict):
    """state dict -> (ctypes array of vp_tensor_desc, list keeping the float32 arrays alive)"""
    sd = state_dict['state_dict'] if 'state_dict' in state_dict else state_dict  # inference.py:163-166
    keep, descs = [], []
    for name, v in sd.items():
        if name.endswith('num_batches_tracked'):
            continue
        a = _as_f32_numpy(v)
        keep.append(a)
        descs.append(capi.vp_tensor_desc(name.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.size))
    return (capi.vp_tensor_desc * len(descs))(*descs), keep


class PinnedArray:
    """numpy view of page-locked host memory (vp_host_alloc): the buffers vp_infer_submit can copy from / to asynchronously."""

    def __init__(self, shape, dtype):
        self._lib = capi.load_library()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._ptr = self._lib.vp_host_alloc(max(n, 1))
        if not self._ptr:
            raise MemoryError(f'vp_host_alloc({n}) failed')
        buf = (C.c_char * max(n, 1)).from_address(self._ptr)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if getattr(self, '_ptr', None):
            self.array = None
            self._lib.vp_host_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class VitPoseHip:
    """ViTPose (backbone + head + decode) on one MI355X through libvitpose_hip.so."""

    def __init__(self, shape: ModelShape, state_dict, dtype: str = 'fp16', device_id: int = 0, max_batch: int = 64):
        self.lib = capi.load_library()
        self.shape = shape
        self.dtype = dtype
        self.device_id = int(device_id)
        self.max_batch = int(max_batch)
        self.K = shape.num_keypoints
        cfg = capi.vp_config(shape.embed_dim, shape.depth, shape.num_heads, shape.num_keypoints,
                             capi.DTYPES[dtype], self.device_id, self.max_batch)
        h = C.c_void_p()
        capi.check(self.lib.vp_create(C.byref(h), C.byref(cfg)))
        self._h = h
        try:
            self._load(state_dict)
        except Exception:
            self.close()
            raise

    # -------------------------------------------------------------- weights
    def _load(self, state_dict):
        arr, keep = _tensor_descs(state_dict)
        code = self.lib.vp_load_weights(self._h, arr, len(arr))
        if code == capi.VP_ERR_MISSING_TENSOR:
            raise KeyError(capi.last_error(self._h))      # load_state_dict's "Missing key(s)"
        if code == capi.VP_ERR_SHAPE:
            raise RuntimeError(capi.last_error(self._h))  # load_state_dict's "size mismatch"
        capi.check(code, self._h)

    # ------------------------------------------------------------ inference
    @staticmethod
    def _fmt(crops: np.ndarray):
        if crops.dtype == np.uint8:
            assert crops.ndim == 4 and crops.shape[1:] == (IMG_H, IMG_W, 3), \
                f'uint8 crops must be [N,{IMG_H},{IMG_W},3], got {crops.shape}'
            return capi.VP_INPUT_U8_NHWC
        assert crops.dtype == np.float32 and crops.ndim == 4 and crops.shape[1:] == (3, IMG_H, IMG_W), \
            f'float32 crops must be [N,3,{IMG_H},{IMG_W}], got {crops.dtype} {crops.shape}'
        return capi.VP_INPUT_F32_NCHW

    def infer(self, crops: np.ndarray, org_wh=None) -> np.ndarray:
        """crops (uint8 NHWC raw, or float32 NCHW normalised) -> float32 [N, K, 3] (y, x, conf)."""
        crops = np.ascontiguousarray(crops)
        fmt = self._fmt(crops)
        n = crops.shape[0]
        out = np.empty((n, self.K, 3), dtype=np.float32)
        if n == 0:
            return out
        wh = None if org_wh is None else np.ascontiguousarray(org_wh, dtype=np.int32).reshape(n, 2)
        capi.check(self.lib.vp_infer(self._h, crops.ctypes.data, fmt, n,
                                     None if wh is None else wh.ctypes.data, out.ctypes.data), self._h)
        return out

    def infer_device(self, d_crops, d_out, org_wh=None, sync: bool = True, ordered: bool = True):
        """Device-resident torch tensors in/out (no copies).  With `ordered` (default) the call goes through the stream-ordered entry
        (vp_infer_device_stream, contract in include/vitpose_hip.h): the library's kernels are ordered after everything already
        enqueued on torch's CURRENT stream (the producers of `d_crops`) and torch work enqueued afterwards waits for them -- `d_out`
        can be consumed by the next torch op without a host synchronisation.  WHERE the kernels run depends on the batch: up to 16
        crops (`VP_CALLER_STREAM`) they are launched on torch's current stream itself -- work the caller enqueues on that stream
        afterwards runs BEHIND them, not beside them, and the handle stays busy on that stream until the next call, `synchronize()` or
        `close()`; larger batches run on the library's own stream, fenced against the caller's with two events (work enqueued on the
        caller's stream afterwards waits for them too, other streams overlap freely).  `sync` additionally blocks the host until the
        result is complete.  `ordered=False`: the library's own stream with no ordering against torch's streams (vp_infer_device)."""
        import torch
        assert d_crops.is_cuda and d_out.is_cuda and d_crops.is_contiguous() and d_out.is_contiguous()
        fmt = capi.VP_INPUT_U8_NHWC if d_crops.dtype == torch.uint8 else capi.VP_INPUT_F32_NCHW
        n = d_crops.shape[0]
        assert d_out.dtype == torch.float32 and d_out.numel() == n * self.K * 3
        whp = None
        if org_wh is not None:
            assert org_wh.is_cuda and org_wh.dtype == torch.int32 and org_wh.numel() == 2 * n
            whp = org_wh.data_ptr()
        if ordered:
            cs = torch.cuda.current_stream(d_crops.device).cuda_stream
            capi.check(self.lib.vp_infer_device_stream(self._h, d_crops.data_ptr(), fmt, n, whp, d_out.data_ptr(), cs), self._h)
            if sync:
                self.synchronize()
        else:
            capi.check(self.lib.vp_infer_device(self._h, d_crops.data_ptr(), fmt, n, whp, d_out.data_ptr(), int(sync)), self._h)
        return d_out

    def submit(self, crops: np.ndarray, out: np.ndarray, org_wh=None) -> int:
        """Asynchronous host path (vp_infer_submit): enqueue one batch (<= max_batch crops) and return its slot; `crops`,
        `org_wh` and `out` must stay alive and untouched until `wait(slot)`.  Pinned buffers (PinnedArray) make the copies
        overlap the previous batch's compute; two batches may be in flight."""
        assert crops.flags['C_CONTIGUOUS'] and out.flags['C_CONTIGUOUS'] and out.dtype == np.float32
        n = crops.shape[0]
        assert out.size == n * self.K * 3
        wh = None if org_wh is None else np.ascontiguousarray(org_wh, dtype=np.int32).reshape(n, 2)
        self._inflight_wh = getattr(self, '_inflight_wh', {})
        slot = C.c_int32(-1)
        capi.check(self.lib.vp_infer_submit(self._h, crops.ctypes.data, self._fmt(crops), n, None if wh is None else wh.ctypes.data,
                                            out.ctypes.data, C.byref(slot)), self._h)
        self._inflight_wh[slot.value] = (crops, wh, out)
        return slot.value

    def wait(self, slot: int):
        capi.check(self.lib.vp_infer_wait(self._h, int(slot)), self._h)
        getattr(self, '_inflight_wh', {}).pop(int(slot), None)

    def infer_flip(self, crops: np.ndarray, flip_pairs, org_wh=None, shift_heatmap: bool = False, return_heatmaps: bool = False):
        """Flip-test inference (reference head `inference_model(x, flip_pairs)` + `flip_back`, topdown_heatmap_simple_head.py:
        195-218, post_transforms.py:110-147): average of the heatmaps of the crops and of their flipped-back mirror images,
        then the usual decode.  `flip_pairs`: the dataset's mirror joint pairs [[l, r], ...] (the reference ships none)."""
        crops = np.ascontiguousarray(crops)
        n = crops.shape[0]
        pairs = np.ascontiguousarray(np.asarray(flip_pairs, dtype=np.int32).reshape(-1, 2))
        wh = None if org_wh is None else np.ascontiguousarray(org_wh, dtype=np.int32).reshape(n, 2)
        out = np.empty((n, self.K, 3), dtype=np.float32)
        hm = np.empty((n, self.K, HM_H, HM_W), dtype=np.float32) if return_heatmaps else None
        if n:
            capi.check(self.lib.vp_infer_flip(self._h, crops.ctypes.data, self._fmt(crops), n,
                                              None if wh is None else wh.ctypes.data, pairs.ctypes.data if len(pairs) else None,
                                              len(pairs), int(bool(shift_heatmap)), out.ctypes.data,
                                              None if hm is None else hm.ctypes.data), self._h)
        return (out, hm) if return_heatmaps else out

    def infer_frame(self, frame: np.ndarray, params: np.ndarray) -> np.ndarray:
        """Whole frame + crop geometry (cropprep.crop_params) -> [n, K, 3] in padded-crop pixels (vp_infer_frame)."""
        frame = np.ascontiguousarray(frame)
        assert frame.dtype == np.uint8 and frame.ndim == 3 and frame.shape[2] == 3
        params = np.ascontiguousarray(params, dtype=np.int32).reshape(-1, 8)
        n = params.shape[0]
        out = np.empty((n, self.K, 3), dtype=np.float32)
        if n == 0:
            return out
        capi.check(self.lib.vp_infer_frame(self._h, frame.ctypes.data, frame.shape[0], frame.shape[1],
                                           params.ctypes.data, n, out.ctypes.data), self._h)
        return out

    def heatmaps(self, crops: np.ndarray) -> np.ndarray:
        crops = np.ascontiguousarray(crops)
        n = crops.shape[0]
        out = np.empty((n, self.K, HM_H, HM_W), dtype=np.float32)
        capi.check(self.lib.vp_infer_heatmaps(self._h, crops.ctypes.data, self._fmt(crops), n, out.ctypes.data), self._h)
        return out

    def tokens(self, crops: np.ndarray) -> np.ndarray:
        crops = np.ascontiguousarray(crops)
        n = crops.shape[0]
        out = np.empty((n, 192, self.shape.embed_dim), dtype=np.float32)
        capi.check(self.lib.vp_infer_tokens(self._h, crops.ctypes.data, self._fmt(crops), n, out.ctypes.data), self._h)
        return out

    # ------------------------------------------------------------ profiling
    def set_profiling(self, families=True):
        """True/False = all/none, or an iterable of family names from ``_capi.VP_PROF_NAMES``."""
        if families is True:
            mask = -1
        elif not families:
            mask = 0
        else:
            mask = 0
            for f in families:
                mask |= 1 << capi.VP_PROF_NAMES.index(f)
        capi.check(self.lib.vp_set_profiling(self._h, mask), self._h)

    def reset_profile(self):
        capi.check(self.lib.vp_reset_profile(self._h), self._h)

    def profile(self) -> dict:
        p = capi.vp_profile()
        capi.check(self.lib.vp_get_profile(self._h, C.byref(p)), self._h)
        return {name: dict(ms=p.ms[i], flops=p.flops[i], bytes=p.bytes[i], launches=p.launches[i])
                for i, name in enumerate(capi.VP_PROF_NAMES)}

    def profile_kernel(self, family: str) -> str:
        """Name of the kernel the last launch of `family` ran on, as the library's launch code recorded it (vp_profile_kernel)."""
        buf = C.create_string_buffer(256)
        capi.check(self.lib.vp_profile_kernel(self._h, capi.VP_PROF_NAMES.index(family), buf, len(buf)), self._h)
        return buf.value.decode('utf-8', 'replace')

    def synchronize(self):
        capi.check(self.lib.vp_synchronize(self._h), self._h)

    def close(self):
        if getattr(self, '_h', None):
            self.lib.vp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_heatmaps(heatmaps: np.ndarray, org_wh=None, device_id: int = 0) -> np.ndarray:
    """GPU decode of host heatmaps [N,K,64,48] -> [N,K,3] (y, x, conf) (vp_decode_only)."""
    lib = capi.load_library()
    hm = np.ascontiguousarray(heatmaps, dtype=np.float32)
    n, k, h, w = hm.shape
    assert (h, w) == (HM_H, HM_W)
    out = np.empty((n, k, 3), dtype=np.float32)
    wh = None if org_wh is None else np.ascontiguousarray(org_wh, dtype=np.int32).reshape(n, 2)
    capi.check(lib.vp_decode_only(device_id, hm.ctypes.data, n, k, None if wh is None else wh.ctypes.data, out.ctypes.data))
    return out


def crop_prep_device(frame: np.ndarray, params: np.ndarray, device_id: int = 0) -> np.ndarray:
    """The device crop/pad/resize kernel alone (vp_dbg_crop_prep): uint8 [n, 256, 192, 3]."""
    lib = capi.load_library()
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    params = np.ascontiguousarray(params, dtype=np.int32).reshape(-1, 8)
    out = np.empty((len(params), IMG_H, IMG_W, 3), dtype=np.uint8)
    capi.check(lib.vp_dbg_crop_prep(device_id, frame.ctypes.data, frame.shape[0], frame.shape[1], params.ctypes.data,
                                    len(params), out.ctypes.data))
    return out


class VitPoseGroup:
    """One process, N GPUs (vp_group_*): weights replicated, crops of a call sharded contiguously, all devices concurrent."""

    def __init__(self, shape: ModelShape, state_dict, device_ids, dtype: str = 'fp16', max_batch: int = 64):
        self.lib = capi.load_library()
        self.shape, self.K = shape, shape.num_keypoints
        self.device_ids = [int(d) for d in device_ids]
        cfg = capi.vp_config(shape.embed_dim, shape.depth, shape.num_heads, shape.num_keypoints, capi.DTYPES[dtype], 0, int(max_batch))
        ids = (C.c_int32 * len(self.device_ids))(*self.device_ids)
        g = C.c_void_p()
        code = self.lib.vp_group_create(C.byref(g), C.byref(cfg), ids, len(self.device_ids))
        if code != capi.VP_OK:
            raise capi.VpError(code, (self.lib.vp_group_last_error(None) or b'').decode('utf-8', 'replace'))
        self._g = g
        arr, keep = _tensor_descs(state_dict)
        self._check(self.lib.vp_group_load_weights(self._g, arr, len(arr)))

    def _check(self, code):
        if code != capi.VP_OK:
            raise capi.VpError(code, (self.lib.vp_group_last_error(self._g) or b'').decode('utf-8', 'replace'))

    def infer(self, crops: np.ndarray, org_wh=None) -> np.ndarray:
        crops = np.ascontiguousarray(crops)
        n = crops.shape[0]
        out = np.empty((n, self.K, 3), dtype=np.float32)
        if n == 0:
            return out
        wh = None if org_wh is None else np.ascontiguousarray(org_wh, dtype=np.int32).reshape(n, 2)
        self._check(self.lib.vp_group_infer(self._g, crops.ctypes.data, VitPoseHip._fmt(crops), n,
                                            None if wh is None else wh.ctypes.data, out.ctypes.data))
        return out

    def infer_allgather(self, crops: np.ndarray, d_all, org_wh=None):
        """`d_all`: one torch float32 tensor [n, K, 3] per device of the group; every one receives ALL keypoints (peer copies)."""
        crops = np.ascontiguousarray(crops)
        n = crops.shape[0]
        assert len(d_all) == len(self.device_ids) and all(t.is_cuda and t.is_contiguous() and t.numel() == n * self.K * 3 for t in d_all)
        ptrs = (C.c_void_p * len(d_all))(*[t.data_ptr() for t in d_all])
        out = np.empty((n, self.K, 3), dtype=np.float32)
        wh = None if org_wh is None else np.ascontiguousarray(org_wh, dtype=np.int32).reshape(n, 2)
        self._check(self.lib.vp_group_infer_allgather(self._g, crops.ctypes.data, VitPoseHip._fmt(crops), n,
                                                      None if wh is None else wh.ctypes.data, ptrs, out.ctypes.data))
        return out

    def close(self):
        if getattr(self, '_g', None):
            self.lib.vp_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
