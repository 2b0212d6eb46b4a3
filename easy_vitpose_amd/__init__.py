"""MI355X-native ViTPose inference path (HIP/CDNA4 kernels behind a C ABI).

``VitInference`` keeps the surface of ``easy_ViTPose.VitInference``
(easy_ViTPose/__init__.py:1-5); ``VitPoseHip`` is the batched engine underneath.
Importing the package never touches the GPU; the first use of the engine loads
``_lib/libvitpose_hip.so`` and raises loudly if it is missing (no CPU fallback).
"""
from .configs import ModelShape, model_shape  # noqa: F401


def __getattr__(name):
    if name == 'VitInference':
        from .inference import VitInference
        return VitInference
    if name in ('VitPoseHip', 'VitPoseGroup', 'PinnedArray', 'decode_heatmaps'):
        from . import engine
        return getattr(engine, name)
    raise AttributeError(name)


__all__ = ['VitInference', 'VitPoseHip', 'VitPoseGroup', 'PinnedArray', 'decode_heatmaps', 'ModelShape', 'model_shape']
