#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/bqkv_s.txt; rm -f $O
timeout 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_ops.py -x -q -m gpu -k "blocked or attention or fold or golden" 2>&1 | grep -E "passed|failed|rror" | tail -3 >> $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "s-coco or s_coco or golden or peaked" 2>&1 | grep -E "passed|failed|rror" | tail -3 >> $O
for cfg in "--variant s --dataset coco --batch 256" "--variant s --dataset coco --batch 8"; do
  echo "== $cfg" >> $O
  for r in 1 2 3; do for f in 0 1; do
  echo -n "VP_BLOCKED_QKV=$f: " >> $O
  VP_BLOCKED_QKV=$f timeout 300 python bench.py $cfg --steps 40 --warmup 5 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o='attention %.3f qkv %.3f ms/step' % (d['breakdown']['attention']['ms_per_step'], d['breakdown']['gemm_qkv']['ms_per_step'])
        else: print(d['value'], d['ms_per_step'], o)
" >> $O
  done; done
done
cat $O
