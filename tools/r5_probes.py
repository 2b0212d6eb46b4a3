#!/usr/bin/env python3
"""Round-5 calibration probes (GPU box only; measurement build):
  stream   HBM ceilings: copy / read-only / write-only over 1 GiB per direction, U x 16 bytes in flight per lane, persistent grids, plain and
           non-temporal accesses (the round-1 copy kernel kept one load in flight per lane)
  valu     v_fma_f32 x 2 against v_pk_fma_f32 where no MFMA issues beside them
  hwid     where the two 512-thread workgroups of a CU land (wave slots, CU, XCD): basis of the attn.proj start stagger
  h80      go / no-go of a fused qkv + attention kernel for head dim 80: the GEMM part alone on the 192 x 256 tile (one crop x one padded head)
    python tools/r5_probes.py [stream] [valu] [hwid] [h80]
"""
import collections
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402  (measurement build of the library)
from easy_vitpose_amd import _capi as capi

lib = capi.load_library()
what = set(sys.argv[1:]) or {'stream', 'valu', 'hwid', 'h80'}


def peak(kind):
    r = C.c_double()
    rc = lib.vp_dbg_peak(0, kind, C.byref(r))
    return r.value if rc == 0 else float('nan')


if 'stream' in what:
    print('== HBM stream ceilings (TB/s of algorithmic bytes; copy = read + written), 1 GiB per direction, best of 4 timed launches', flush=True)
    print(f'round-1 copy kernel (one float4 in flight per lane, 4096 workgroups): {peak(2):.2f}', flush=True)
    gnames = ['512', '1024', '2048', '4096', '8192', '16384', 'one/chunk', '768']
    for mode, mname in ((0, 'copy'), (1, 'read'), (2, 'write')):
        for nt in (0, 1):
            for uc in (0, 1, 2, 3):
                row = []
                for gc in (0, 7, 1, 2, 3, 4, 6):
                    row.append(f'{gnames[gc]}: {peak(300 + 64 * mode + 32 * nt + 8 * uc + gc):5.2f}')
                print(f'{mname:5s} {"nt " if nt else "   "} U={1 << uc}  workgroups ' + '  '.join(row), flush=True)

if 'valu' in what:
    print('== VALU probe: ns per round of 64 multiply-adds per lane, one workgroup per CU', flush=True)
    for two in (0, 1):
        a, b = peak(500 + 2 * two), peak(501 + 2 * two)
        print(f'{8 if two else 4} waves per CU: 64 v_fma_f32 {a:7.2f} ns   32 v_pk_fma_f32 {b:7.2f} ns   ratio {a / b:.2f}', flush=True)

if 'hwid' in what and hasattr(lib, 'vp_dbg_hwid_probe'):
    lib.vp_dbg_hwid_probe.argtypes = [C.c_int32] * 5 + [C.POINTER(C.c_uint32)]
    lib.vp_dbg_hwid_probe.restype = C.c_int
    blocks = 1536
    buf = (C.c_uint32 * (blocks * 4))()
    rc = lib.vp_dbg_hwid_probe(0, blocks, 512, 80 * 1024, 64, buf)
    print(f'== HW_ID probe: {blocks} workgroups x 512 threads, 80 KiB LDS (two per CU), rc={rc}', flush=True)
    rows = []
    for b in range(blocks):
        hw, xcc, lo, hi = buf[4 * b:4 * b + 4]
        rows.append(dict(b=b, wave=hw & 15, simd=(hw >> 4) & 3, pipe=(hw >> 6) & 3, cu=(hw >> 8) & 15, sh=(hw >> 12) & 1, se=(hw >> 13) & 7,
                         xcc=xcc & 15, t=(hi << 32) | lo, raw=hw))
    t0 = min(r['t'] for r in rows)
    for r in rows[:24]:
        print(f"  wg {r['b']:4d}: xcc {r['xcc']} se {r['se']} sh {r['sh']} cu {r['cu']:2d} simd {r['simd']} wave slot {r['wave']} start +{r['t'] - t0} raw {r['raw']:#010x}")
    first = [r for r in rows if r['b'] < 512]
    per_cu = collections.defaultdict(list)
    for r in first:
        per_cu[(r['xcc'], r['se'], r['sh'], r['cu'])].append(r)
    print(f'  first 512 workgroups occupy {len(per_cu)} distinct (xcc, se, sh, cu); workgroups per CU: {collections.Counter(len(v) for v in per_cu.values())}')
    print(f'  xcc of wg b == b % 8 for {sum(r["xcc"] == r["b"] % 8 for r in rows)} of {blocks}')
    slots = collections.Counter(tuple(sorted(x['wave'] for x in v)) for v in per_cu.values())
    print(f'  wave slots (wave 0 of each workgroup) of the co-resident pairs: {dict(slots)}')
    diffs = collections.Counter((max(x['b'] for x in v) - min(x['b'] for x in v)) for v in per_cu.values() if len(v) == 2)
    print(f'  blockIdx distance of the two workgroups of a CU: {dict(diffs)}')
    late = sorted(r['t'] - t0 for r in rows if r['b'] >= 512)
    if late:
        print(f'  start of workgroups >= 512 (cycles after the first): min {late[0]} median {late[len(late) // 2]} max {late[-1]}')

if 'h80' in what:
    print('== head dim 80 fusion go / no-go: ViTPose-H at 128 crops, attn.qkv alone', flush=True)
    M, K = 128 * 192, 1280
    LN = 16
    cases = [('production: 256 x 256 tiles, N = 3840 (15 real n-tiles)', 16, 3840, 0),
             ('192 x 256 tiles, N = 3840', 18, 3840, 0),
             ('192 x 256 tiles, N = 4096 (16 heads padded 240 -> 256 columns: the one-crop fused tile)', 18, 4096, 0),
             ('192 x 256 tiles, N = 4096, no stores (what the fusion removes from the GEMM)', 18, 4096, 8),
             ('192 x 256 tiles, N = 4096, no stores, no operand DMA', 18, 4096, 9)]
    for name, v, N, abl in cases:
        ms = C.c_float()
        rc = lib.vp_dbg_gemm_bench2(0, 0, 0, v | (abl << 8), 4, LN, M, N, K, 10, C.byref(ms))
        tf = 2.0 * M * 3840 * K / (ms.value * 1e-3) / 1e12 if rc == 0 else 0
        print(f'  {name}: rc={rc} {ms.value * 1e3:7.1f} us  ({tf:6.1f} TF/s of the 3840 useful columns)', flush=True)
