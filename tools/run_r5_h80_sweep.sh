#!/bin/bash
# round 5: batch threshold of the head-dim-80 fused tile (VP_QA80_MIN_TILES): ViTPose-H at small / mid batches, fused against two launches on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for b in 8 12 16 24 32 48 64 96; do for f in 0 1; do
  echo -n "batch $b VP_FUSE_QKV_ATTN=$f: "
  VP_QA80_MIN_TILES=1 VP_FUSE_QKV_ATTN=$f timeout 300 python bench.py --variant h --dataset wholebody --batch $b --steps 30 --warmup 6 --no-cpu-baseline --no-host-path --no-clock 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['encoder_gemms']['gemm_qkv']['kernel'])
"
done; done
