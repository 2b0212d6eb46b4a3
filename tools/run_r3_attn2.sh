#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/attn2.txt; rm -f $O
for cfg in "--variant h --dataset wholebody --batch 128" "--variant s --dataset coco --batch 256"; do
  echo "== $cfg" >> $O
  for r in 1 2; do for L in ord0 ord1 ord2 ord4; do
  echo -n "$L: " >> $O
  VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/ab/$L.so timeout 300 python bench.py $cfg --steps 20 --warmup 4 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o='attention %.3f ms/step' % d['breakdown']['attention']['ms_per_step']
        else: print(d['value'], d['ms_per_step'], o)
" >> $O
  done; done
done
VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/ab/ord2.so timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" 2>&1 | grep -E "passed|failed|rror" | tail -2 >> $O
VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/ab/ord4.so timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" 2>&1 | grep -E "passed|failed|rror" | tail -2 >> $O
cat $O
