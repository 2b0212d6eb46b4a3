#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/lnf.txt; rm -f $O
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or outlier" 2>&1 | grep -E "passed|failed|rror" | tail -2 >> $O
for cfg in "--variant h --dataset wholebody --batch 128" "--variant b --dataset coco --batch 256" "--variant l --dataset coco_25 --batch 16 --input u8"; do
  echo "== $cfg" >> $O
  for r in 1 2; do for L in lnf_old lnf_new; do
  echo -n "$L: " >> $O
  VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/ab/$L.so timeout 300 python bench.py $cfg --steps 20 --warmup 4 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o='layernorm family %.3f ms/step' % d['breakdown']['layernorm']['ms_per_step']
        else: print(d['value'], d['ms_per_step'], o)
" >> $O
  done; done
done
cat $O
