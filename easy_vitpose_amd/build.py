"""In-tree build of libvitpose_hip.so (gfx950 only) with hipcc.

    python -m easy_vitpose_amd.build [--force] [--tools]

The shared object is written to ``easy_vitpose_amd/_lib/`` (git-ignored, but it
travels to the GPU box with the gpurun snapshot).  hipcc cross-compiles for
gfx950 without a GPU, so this also runs in the CPU-only build container.

Two libraries from the same sources:

* ``libvitpose_hip.so`` -- the PRODUCT: what ``_capi.load_library`` loads, what tests / bench / smoke run.
* ``libvitpose_hip_tools.so`` (``--tools``, ``-DVP_TOOLS``) -- the measurement build ``tools/`` load through ``VP_HIP_LIB``:
  ablation flags, start stagger and cycle stamps inside the GEMM kernels, the experimental tile configurations, the
  development environment switches (``include/vitpose_hip_tools.h``).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, '_lib')
LIB = os.path.join(LIBDIR, 'libvitpose_hip.so')
TOOLS_LIB = os.path.join(LIBDIR, 'libvitpose_hip_tools.so')
SOURCES = ['gemm.hip', 'gemm8.hip', 'gemm8f.hip', 'qkvattn.hip', 'quant8.hip', 'attention.hip', 'elementwise.hip', 'decode.hip', 'fp8_probe.hip',
           'vitpose_api.hip', 'weights.hip', 'tile_rules.hip', 'debug_taps.hip']
TOOLS_SOURCES = SOURCES
HEADERS = ['common.h', 'kernels.h', 'gemm8_common.h', 'mx8.h', 'api_internal.h', os.path.join('..', '..', 'include', 'vitpose_hip.h'),
           os.path.join('..', '..', 'include', 'vitpose_hip_tools.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden',
         '-ffp-contract=fast', '-Wno-unused-result']


def _hipcc() -> str:
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: the HIP extension cannot be built (no CPU fallback exists)')


def _stale(target: str, deps: "list[str]") -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, tools: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objs, jobs = [], []
    lib = TOOLS_LIB if tools else LIB
    extra = ['-DVP_TOOLS'] if tools else []
    for src in (TOOLS_SOURCES if tools else SOURCES):
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace('.hip', '.tools.o' if tools else '.o'))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc, *FLAGS, *extra, '-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed: {" ".join(cmd)}\n{r.stdout}\n{r.stderr}')
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(lib, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', lib])
    return lib


if __name__ == '__main__':
    path = build_library(force='--force' in sys.argv, verbose=True, tools='--tools' in sys.argv)
    print(path)
