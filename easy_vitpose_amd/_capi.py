"""ctypes binding of ``include/vitpose_hip.h`` (libvitpose_hip.so).

This is the stub a maintainer of the reference would add next to
``easy_ViTPose/inference.py``.  There is deliberately no
fallback: if the shared object is missing or cannot be loaded, importing the
compute entry points raises ``HipExtensionMissing``.
"""
from __future__ import annotations

import ctypes as C
import os

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_lib', 'libvitpose_hip.so')

VP_OK, VP_ERR_INVALID, VP_ERR_HIP, VP_ERR_STATE, VP_ERR_MISSING_TENSOR, VP_ERR_SHAPE = range(6)
VP_DTYPE_F16, VP_DTYPE_BF16, VP_DTYPE_FP8 = 0, 1, 2
VP_INPUT_F32_NCHW, VP_INPUT_U8_NHWC = 0, 1
VP_PROF_NAMES = ['gemm_proj', 'gemm_fc1', 'gemm_qkv', 'gemm_patch', 'gemm_deconv', 'gemm_final',
                 'attention', 'layernorm', 'im2col', 'decode', 'gemm_fc2']
VP_PROF_COUNT = len(VP_PROF_NAMES)
DTYPES = {'fp16': VP_DTYPE_F16, 'f16': VP_DTYPE_F16, 'bf16': VP_DTYPE_BF16, 'fp8': VP_DTYPE_FP8}

# every symbol include/vitpose_hip.h declares (tests check the .so exports all of them)
SYMBOLS = ['vp_abi_version', 'vp_create', 'vp_load_weights', 'vp_infer', 'vp_infer_device', 'vp_infer_device_stream', 'vp_host_alloc', 'vp_host_free',
           'vp_infer_submit', 'vp_infer_wait', 'vp_group_create', 'vp_group_size', 'vp_group_member', 'vp_group_load_weights', 'vp_group_infer',
           'vp_group_infer_allgather', 'vp_group_destroy', 'vp_group_last_error', 'vp_infer_frame', 'vp_infer_flip', 'vp_infer_heatmaps',
           'vp_infer_tokens', 'vp_decode_only', 'vp_stream', 'vp_synchronize', 'vp_set_profiling',
           'vp_reset_profile', 'vp_get_profile', 'vp_profile_kernel', 'vp_group_peer_access_missing', 'vp_destroy', 'vp_last_error',
           'vp_dbg_gemm', 'vp_dbg_attention', 'vp_dbg_layernorm', 'vp_dbg_deconv', 'vp_dbg_gemm_case', 'vp_dbg_crop_prep',
           'vp_dbg_group_plan', 'vp_dbg_group_trace', 'vp_dbg_gemm8_pick', 'vp_dbg_gemm2_pick', 'vp_dbg_splitk_pick', 'vp_dbg_run_batch', 'vp_dbg_fp8_gemm', 'vp_dbg_mx_gemm', 'vp_dbg_host_e4m3', 'vp_dbg_gemm_fp8_case', 'vp_dbg_qkvattn']


class HipExtensionMissing(RuntimeError):
    pass


class VpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f'[vitpose_hip status {code}] {msg}')
        self.code = code
        self.msg = msg


class vp_config(C.Structure):
    _fields_ = [('embed_dim', C.c_int32), ('depth', C.c_int32), ('num_heads', C.c_int32),
                ('num_keypoints', C.c_int32), ('dtype', C.c_int32), ('device_id', C.c_int32),
                ('max_batch', C.c_int32)]


class vp_tensor_desc(C.Structure):
    _fields_ = [('name', C.c_char_p), ('data', C.POINTER(C.c_float)), ('numel', C.c_int64)]


class vp_profile(C.Structure):
    _fields_ = [('ms', C.c_double * VP_PROF_COUNT), ('flops', C.c_double * VP_PROF_COUNT),
                ('bytes', C.c_double * VP_PROF_COUNT), ('launches', C.c_int64 * VP_PROF_COUNT)]


_lib = None


def load_library():
    """dlopen the in-tree HIP library (once).  Raises HipExtensionMissing loudly."""
    global _lib
    if _lib is not None:
        return _lib
    lib_path = os.environ.get('VP_HIP_LIB', LIB_PATH)   # kernel development only: A/B two builds on the same box
    if not os.path.exists(lib_path):
        raise HipExtensionMissing(
            f'{lib_path} not found. Build it with `python -m easy_vitpose_amd.build` '
            '(needs hipcc). There is no CPU fallback for the ViTPose path.')
    try:
        lib = C.CDLL(lib_path)
    except OSError as e:  # e.g. libamdhip64 missing
        raise HipExtensionMissing(f'cannot load {lib_path}: {e}') from e
    H = C.c_void_p
    lib.vp_abi_version.restype = C.c_int
    lib.vp_create.argtypes = [C.POINTER(H), C.POINTER(vp_config)]
    lib.vp_load_weights.argtypes = [H, C.POINTER(vp_tensor_desc), C.c_int32]
    lib.vp_infer.argtypes = [H, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.vp_infer_device.argtypes = [H, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
    lib.vp_infer_device_stream.argtypes = [H, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.vp_host_alloc.argtypes = [C.c_size_t]
    lib.vp_host_alloc.restype = C.c_void_p
    lib.vp_host_free.argtypes = [C.c_void_p]
    lib.vp_host_free.restype = None
    lib.vp_infer_submit.argtypes = [H, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    lib.vp_infer_wait.argtypes = [H, C.c_int32]
    lib.vp_group_create.argtypes = [C.POINTER(H), C.POINTER(vp_config), C.POINTER(C.c_int32), C.c_int32]
    lib.vp_group_size.argtypes = [H]
    lib.vp_group_member.argtypes = [H, C.c_int32]
    lib.vp_group_member.restype = C.c_void_p
    lib.vp_group_load_weights.argtypes = [H, C.POINTER(vp_tensor_desc), C.c_int32]
    lib.vp_group_infer.argtypes = [H, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.vp_group_infer_allgather.argtypes = [H, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]
    lib.vp_group_destroy.argtypes = [H]
    lib.vp_group_last_error.argtypes = [H]
    lib.vp_group_last_error.restype = C.c_char_p
    lib.vp_infer_frame.argtypes = [H, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    lib.vp_dbg_crop_prep.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    lib.vp_infer_heatmaps.argtypes = [H, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.vp_infer_tokens.argtypes = [H, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.vp_decode_only.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.vp_stream.argtypes = [H]
    lib.vp_stream.restype = C.c_void_p
    lib.vp_synchronize.argtypes = [H]
    lib.vp_set_profiling.argtypes = [H, C.c_int32]
    lib.vp_reset_profile.argtypes = [H]
    lib.vp_get_profile.argtypes = [H, C.POINTER(vp_profile)]
    lib.vp_destroy.argtypes = [H]
    lib.vp_last_error.argtypes = [H]
    lib.vp_last_error.restype = C.c_char_p
    lib.vp_dbg_gemm.argtypes = [C.c_int32] * 6 + [C.c_void_p] * 5
    lib.vp_dbg_attention.argtypes = [C.c_int32] * 5 + [C.c_void_p] * 2
    lib.vp_dbg_layernorm.argtypes = [C.c_int32] * 4 + [C.c_void_p] * 5
    lib.vp_dbg_deconv.argtypes = [C.c_int32] * 6 + [C.c_void_p, C.POINTER(vp_tensor_desc), C.c_int32, C.c_void_p]
    lib.vp_infer_flip.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                  C.c_void_p, C.c_void_p]
    lib.vp_dbg_gemm_case.argtypes = [C.c_int32] * 9 + [C.c_void_p] * 8
    if hasattr(lib, 'vp_dbg_gemm8_timeline'):   # the measurement build (include/vitpose_hip_tools.h; tools/ point VP_HIP_LIB at it)
        lib.vp_dbg_gemm8_timeline.argtypes = [C.c_int32] * 9 + [C.POINTER(C.c_uint64), C.c_int32]
        lib.vp_dbg_gemm_timeline.argtypes = [C.c_int32] * 6 + [C.POINTER(C.c_uint64), C.c_int32]
        lib.vp_dbg_gemm8_timeline.restype = lib.vp_dbg_gemm_timeline.restype = C.c_int
        lib.vp_dbg_gemm_bench.argtypes = [C.c_int32] * 9 + [C.POINTER(C.c_float)]
        lib.vp_dbg_gemm_bench2.argtypes = [C.c_int32] * 10 + [C.POINTER(C.c_float)]
        lib.vp_dbg_gemm_compare.argtypes = [C.c_int32] * 13 + [C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        lib.vp_dbg_peak.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_double)]
        for n_ in ('vp_dbg_gemm_bench', 'vp_dbg_gemm_bench2', 'vp_dbg_gemm_compare', 'vp_dbg_peak'):
            getattr(lib, n_).restype = C.c_int
    lib.vp_profile_kernel.argtypes = [H, C.c_int32, C.c_char_p, C.c_int32]
    lib.vp_group_peer_access_missing.argtypes = [H]
    lib.vp_dbg_group_plan.argtypes = [C.c_int32] * 3 + [C.c_void_p, C.c_void_p, C.c_int32]
    lib.vp_dbg_gemm8_pick.argtypes = [C.c_int32] * 4 + [C.c_void_p]
    lib.vp_dbg_gemm2_pick.argtypes = [C.c_int32] * 4 + [C.c_void_p]
    lib.vp_dbg_splitk_pick.argtypes = [C.c_int32] * 3 + [C.c_void_p]
    lib.vp_dbg_run_batch.argtypes = [C.c_int32] * 3
    lib.vp_dbg_group_trace.argtypes = [C.c_int32] * 3 + [C.c_void_p, C.c_int32]
    lib.vp_dbg_fp8_gemm.argtypes = [C.c_int32] * 4 + [C.c_void_p] * 7
    lib.vp_dbg_mx_gemm.argtypes = [C.c_int32] * 4 + [C.c_void_p] * 7
    lib.vp_dbg_host_e4m3.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.vp_dbg_gemm_fp8_case.argtypes = [C.c_int32] * 5 + [C.c_void_p] * 8
    lib.vp_dbg_qkvattn.argtypes = [C.c_int32] * 5 + [C.c_void_p] * 4
    for name in SYMBOLS:
        if name not in ('vp_stream', 'vp_last_error', 'vp_host_alloc', 'vp_host_free', 'vp_group_member', 'vp_group_last_error'):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def last_error(handle=None) -> str:
    lib = load_library()
    s = lib.vp_last_error(handle)
    return s.decode('utf-8', 'replace') if s else ''


def check(code: int, handle=None):
    if code != VP_OK:
        raise VpError(code, last_error(handle))
