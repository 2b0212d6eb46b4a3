#!/bin/bash
# small-batch operating point (one GPU's share of a 64-person frame on 8 GPUs): per-family breakdown
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/small.txt; rm -f $O
for cfg in "--variant l --dataset coco_25 --batch 8 --input u8" "--variant l --dataset coco_25 --batch 16 --input u8" "--variant b --dataset coco --batch 8 --input u8" "--variant l --dataset coco_25 --batch 1 --input u8"; do
  echo "== $cfg" >> $O
  for g in 1 0; do
  echo -n "VP_GRAPH=$g: " >> $O
  VP_GRAPH=$g timeout 300 python bench.py $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o=' '.join(f'{k[5:9] if k.startswith(\"gemm\") else k[:5]}={v[\"ms_per_step\"]:.3f}/{v[\"launches_per_step\"]}' for k,v in d['breakdown'].items())
        else: print(d['value'], d['ms_per_step'], o, ' sum', )
" >> $O
  done
done
cat $O
