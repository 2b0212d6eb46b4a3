#!/usr/bin/env python3
"""Shader clock and board power WHILE a kernel runs (GPU box only): a sampler thread reads the amdgpu hwmon files (freq1_input = sclk,
power1_average / power1_input) and `rocm-smi --showclocks --showpower --json` while the main thread loops one kernel for a few seconds.
Question it answers: is the chip power-capped under the 8-phase GEMM (clock well below the 2.4 GHz the 2.5 PF peak assumes)?
    python tools/clock_power_probe.py [--secs 4]"""
import argparse
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import threading
import time

try:
    import torch   # first: torch initialises HIP itself (importing it after the library's first HIP calls found no device)
    torch.cuda.init()
except Exception:   # noqa: BLE001
    torch = None

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if 'VP_HIP_LIB' not in os.environ:
    import _toolslib  # noqa: F401,E402  (measurement build of the library)
from easy_vitpose_amd import _capi as capi

ap = argparse.ArgumentParser()
ap.add_argument('--secs', type=float, default=4.0)
args = ap.parse_args()
lib = capi.load_library()


def hwmon_files():
    out = {}
    for d in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'):
        for name in ('freq1_input', 'freq2_input', 'power1_average', 'power1_input', 'power1_cap'):
            p = os.path.join(d, name)
            if os.path.exists(p):
                out.setdefault(name, p)
    return out


HW = hwmon_files()
print('hwmon files:', HW, flush=True)


def read_hw():
    r = {}
    for k, p in HW.items():
        try:
            r[k] = float(open(p).read().strip())
        except Exception:
            pass
    return r


def smi_once():
    try:
        o = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=20).stdout
        return json.loads(o)
    except Exception as e:
        return {'error': str(e)}


class Sampler(threading.Thread):
    """rocm-smi back to back; every sample carries the wall-clock interval of its call"""
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.smi = []

    def run(self):
        while not self.stop:
            t0 = time.time()
            r = smi_once()
            self.smi.append((t0, time.time(), r))


def num(s):
    return float(''.join(ch for ch in str(s) if ch.isdigit() or ch == '.'))


def loop(name, fn):
    fn()   # warm
    s = Sampler()
    s.start()
    t0 = time.time()
    vals = []
    while time.time() - t0 < args.secs:
        vals.append(fn())
    t1 = time.time()
    s.stop = True
    s.join()
    clk, pw = [], []
    for a, b, r in s.smi:
        if a < t0 + 0.7 or b > t1 or not isinstance(r, dict) or 'error' in r:
            continue
        card = next(iter(r.values()))
        clk.append(num(card.get('sclk clock speed:', '0')))
        pw.append(num(card.get('Current Socket Graphics Package Power (W)', '0')))
    vals.sort()
    m = lambda v: sum(v) / len(v) if v else float('nan')
    print(f'{name:44s} {vals[len(vals) // 2]:9.2f}   sclk {m(clk):6.0f} MHz ({min(clk, default=0):.0f}-{max(clk, default=0):.0f})   power {m(pw):6.0f} W ({min(pw, default=0):.0f}-{max(pw, default=0):.0f})   n={len(clk)}', flush=True)


def gemm(epi, variant, group, flags, M, N, K):
    def f():
        ms = C.c_float()
        rc = lib.vp_dbg_gemm_bench2(0, 0, epi, variant, group, flags, M, N, K, 50, C.byref(ms))
        assert rc == 0, capi.last_error()
        return ms.value * 1e3
    return f


def peak(kind):
    def f():
        r = C.c_double()
        rc = lib.vp_dbg_peak(0, kind, C.byref(r))
        assert rc == 0, capi.last_error()
        return r.value
    return f


M = 49152
print('idle:', json.dumps(smi_once())[:600], flush=True)
print('kernel                                        us | TF | TB/s', flush=True)


def gemm1(epi, variant, group, M, N, K):   # vp_dbg_gemm_bench: variant | ablation << 8 (1 = no operand DMA after the prologue, 8 = no stores)
    def f():
        ms = C.c_float()
        rc = lib.vp_dbg_gemm_bench(0, 0, epi, variant, group, M, N, K, 50, C.byref(ms))
        assert rc == 0, capi.last_error()
        return ms.value * 1e3
    return f


if os.environ.get('VP_PROBE_SET', 'main') == 'main':
    loop('fc1 gemm8 256x256 (LN fold, GELU)', gemm(1, 16, 8, 16 | 2, M, 3072, 768))
    loop('qkv gemm8 256x256 (LN fold)', gemm(0, 16, 4, 16, M, 2304, 768))
    loop('fc2 gemm8 256x192 (residual)', gemm(6, 17, 2, 4 | 8, M, 768, 3072))
    loop('proj cfg11 192x128', gemm(6, 11, 0, 0, M, 768, 768))
    loop('fc1 cfg8 192x128 2-phase persist', gemm(1, 8, 8, 16 | 2 | 1, M, 3072, 768))
    loop('plain gemm8 4096^3', gemm(0, 16, 8, 0, 4096, 4096, 4096))
    loop('plain gemm8 M=49152 N=3072 K=4096', gemm(0, 16, 8, 0, M, 3072, 4096))
    loop('MFMA-only 16x16x32 (TF)', peak(0))
    loop('MFMA-only 32x32x16 (TF)', peak(1))
    loop('float4 copy 1 GiB (TB/s)', peak(4))
    loop('float4 copy 32 MiB (TB/s)', peak(3))
for ab in ((0, 1, 8, 9) if os.environ.get('VP_PROBE_SET', 'main') != 'torch' else ()):   # plain-epilogue fc1 shape with pieces removed (timing AND power)
    loop(f'fc1 shape, bias epilogue, ablate {ab}', gemm1(0, 16 | (ab << 8), 8, M, 3072, 768))

if os.environ.get('VP_PROBE_SET', 'main') in ('main', 'torch'):   # the vendor library's plain GEMM (torch.mm -> hipBLASLt / rocBLAS) on the same shapes: is IT power-capped too?
    try:
        dev = torch.device('cuda:0')

        def mm(M_, N_, K_):
            a = (torch.rand((M_, K_), device=dev, dtype=torch.float32) * 2 - 1).half()
            b = (torch.rand((K_, N_), device=dev, dtype=torch.float32) * 2 - 1).half()
            c = torch.empty((M_, N_), device=dev, dtype=torch.float16)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def f():
                e0.record()
                for _ in range(50):
                    torch.mm(a, b, out=c)
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / 50 * 1e3
            return f
        loop('torch.mm fp16 49152 x 3072 x 768 (plain)', mm(M, 3072, 768))
        loop('torch.mm fp16 49152 x 3072 x 4096 (plain)', mm(M, 3072, 4096))
        loop('torch.mm fp16 8192^3 (plain)', mm(8192, 8192, 8192))
    except Exception as e:   # noqa: BLE001
        print('torch.mm probe skipped:', e)
elif os.environ.get('VP_PROBE_SET') == 'operand_bits':
    # Round 6: is mlp.fc1 bound by its schedule or by the board power limit?  The SAME kernel, the same instruction stream and the same memory traffic on
    # random operands (what every benchmark of this repository runs), on all-zero operands and on constant operands; with and without the epilogue's stores (ablation 8).
    for tag, fl in (('random operands', 0), ('all-zero operands', 64), ('constant 1.0 operands', 128)):
        loop(f'fc1 gemm8 256x256, {tag}', gemm(1, 16, 8, 16 | 2 | fl, M, 3072, 768))
    for tag, fl in (('random operands', 0), ('all-zero operands', 64)):
        loop(f'fc1 gemm8 256x256 NO STORES, {tag}', gemm(1, 16 | (8 << 8), 8, 16 | 2 | fl, M, 3072, 768))
    for tag, fl in (('random operands', 0), ('all-zero operands', 64)):
        loop(f'fc2 gemm8 192x256, {tag}', gemm(6, 18, 2, 4 | 8 | fl, M, 768, 3072))
    for tag, fl in (('random operands', 0), ('all-zero operands', 64)):
        loop(f'proj cfg11 192x128, {tag}', gemm(6, 11, 0, fl, M, 768, 768))
