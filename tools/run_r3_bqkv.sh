#!/bin/bash
# qkv in the 64 x 64-blocked layout (head dim 64): tests + same-box A/B through the product switch VP_BLOCKED_QKV
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/bqkv.txt; rm -f $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3 >> $O
for cfg in "--variant b --dataset coco --batch 256" "--variant l --dataset coco_25 --batch 64" "--variant l --dataset coco_25 --batch 8 --input u8"; do
  echo "== $cfg" >> $O
  for r in 1 2 3; do for f in 0 1 2; do
  echo -n "VP_BLOCKED_QKV=$f: " >> $O
  VP_BLOCKED_QKV=$f timeout 300 python bench.py $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o='attention %.3f qkv %.3f proj %.3f ms/step' % (d['breakdown']['attention']['ms_per_step'], d['breakdown']['gemm_qkv']['ms_per_step'], d['breakdown']['gemm_proj']['ms_per_step'])
        else: print(d['value'], d['ms_per_step'], o)
" >> $O
  done; done
done
cat $O
