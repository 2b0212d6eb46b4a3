#!/bin/bash
# split-run residual register epilogue (ViTPose-L / H): identity against the 2-phase kernels + same-box A/B of old / new product builds
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/resid2.txt; rm -f $O
for v in l h; do
  echo "== gemm8_check --variant $v --batch 128" >> $O
  timeout 300 python tools/gemm8_check.py --variant $v --batch 128 --reps 2 >> $O 2>&1
done
for r in 1 2 3; do for L in old new; do for v in l h; do
  echo -n "$L $v: " >> $O
  VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/ab/$L.so timeout 300 python bench.py --variant $v --batch 128 --steps 15 --warmup 3 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o=' '.join(f'{k[5:9] if k.startswith(\"gemm\") else k[:5]}={v[\"ms_per_step\"]:.3f}' for k,v in d['breakdown'].items())
        else: print(d['value'], d['ms_per_step'], o)
" >> $O
done; done; done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "outlier or golden" 2>&1 | tail -5 >> $O
cat $O
