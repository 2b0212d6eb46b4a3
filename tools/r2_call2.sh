#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c2; O=gpurun_out/c2
timeout 600 python tools/gemm8_check.py --reps 2 --no-bench > $O/check.txt 2>&1
timeout 200 python tools/gemm8_timeline.py > $O/timeline.txt 2>&1
VP_G8_STAGGER=1 timeout 200 python tools/gemm8_timeline.py > $O/timeline_stag1.txt 2>&1
timeout 200 python tools/gemm8_timeline.py --ablate 64 > $O/timeline_nt.txt 2>&1
for st in 0 1 2 3; do
  echo "== stagger $st" >> $O/bench_stag.txt
  VP_G8_STAGGER=$st timeout 300 python tools/gemm8_check.py --no-compare --iters 10 >> $O/bench_stag.txt 2>&1
done
python - >> $O/bench_nt.txt 2>&1 <<'PY'
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
from easy_vitpose_amd import _capi as capi
lib = capi.load_library()
M = 49152
for name, epi, N, K, v, fl in (('qkv', 0, 2304, 768, 16, 16), ('fc1', 1, 3072, 768, 16, 18)):
    for ab in (0, 64):
        ms = C.c_float()
        rc = lib.vp_dbg_gemm_bench2(0, 0, epi, v | (ab << 8), 8, fl, M, N, K, 10, C.byref(ms))
        print(name, 'ablate', ab, 'rc', rc, f'{ms.value*1e3:.1f} us', flush=True)
PY
for st in 0 1 2; do
  echo "== e2e stagger $st" >> $O/bench_e2e.txt
  VP_G8_STAGGER=$st timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --breakdown >> $O/bench_e2e.txt 2>&1
done
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt; cat $O/check.txt $O/timeline.txt $O/timeline_stag1.txt $O/bench_stag.txt $O/bench_nt.txt; grep -h "value\|==" $O/bench_e2e.txt | cut -c1-150
