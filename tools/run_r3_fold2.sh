#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/fold2.txt; rm -f $O
timeout 600 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "fold or graph or config5 or plain" 2>&1 | grep -E "passed|failed|Error|error" | tail -5 >> $O
for cfg in "--variant l --dataset coco_25 --batch 8 --input u8" "--variant l --dataset coco_25 --batch 16 --input u8" "--variant l --dataset coco_25 --batch 1 --input u8" "--variant b --dataset coco --batch 8 --input u8" "--variant b --dataset coco --batch 16 --input u8" "--variant s --dataset coco --batch 8" "--variant h --dataset wholebody --batch 4" \
           "--variant l --dataset coco_25 --batch 24 --input u8" "--variant b --dataset coco --batch 32"; do
  echo "== $cfg" >> $O
  for r in 1 2; do for f in 0 64; do
  echo -n "VP_FOLD_STATS=$f: " >> $O
  VP_FOLD_STATS=$f timeout 300 python bench.py $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-host-path --no-clock 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])
" >> $O
  done; done
done
cat $O
