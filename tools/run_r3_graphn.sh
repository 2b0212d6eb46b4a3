#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/graphn.txt; rm -f $O
for cfg in "--variant b --dataset coco --batch 32" "--variant b --dataset coco --batch 64" "--variant l --dataset coco_25 --batch 32 --input u8" "--variant l --dataset coco_25 --batch 64 --input u8" "--variant b --dataset coco --batch 128"; do
  echo "== $cfg" >> $O
  for r in 1 2; do for g in 1 256; do
  echo -n "VP_GRAPH=$g: " >> $O
  VP_GRAPH=$g timeout 300 python bench.py $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-host-path --no-clock 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['timed'][:12])
" >> $O
  done; done
done
cat $O
