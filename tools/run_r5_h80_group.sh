#!/bin/bash
# round 5: tile order of the head-dim-80 fused kernel (VP_QA80_GROUP = crops per group, crop fastest inside a group; 0 = head fastest), ViTPose-H @ 128, one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for g in 0 4 8 16; do
  echo -n "VP_QA80_GROUP=$g: "
  VP_QA80_GROUP=$g timeout 300 python bench.py --variant h --dataset wholebody --batch 128 --steps 30 --warmup 6 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o=' '.join(f'{k[5:9] if k.startswith(\"gemm\") else k[:5]}={v[\"ms_per_step\"]:.3f}' for k,v in d['breakdown'].items())
        else: print(d['value'], d['ms_per_step'], o)
"
done; done
