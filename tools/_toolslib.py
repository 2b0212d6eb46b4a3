"""Every script under tools/ imports this first: it points the ctypes binding at the MEASUREMENT build of the library
(libvitpose_hip_tools.so = the product sources compiled with -DVP_TOOLS: ablation flags, start stagger and cycle stamps inside the
GEMM kernels, the experimental tile configurations and kernel variants, the development environment switches -- see
include/vitpose_hip_tools.h).  Build it in the build container (`python -m easy_vitpose_amd.build --tools`); the .so travels to the
GPU box with the gpurun snapshot.  An explicit VP_HIP_LIB (tools/ab.sh) wins."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
_TOOLS = os.path.join(ROOT, 'easy_vitpose_amd', '_lib', 'libvitpose_hip_tools.so')
if 'VP_HIP_LIB' not in os.environ:
    if not os.path.exists(_TOOLS):
        from easy_vitpose_amd.build import build_library
        build_library(tools=True)
    os.environ['VP_HIP_LIB'] = _TOOLS
