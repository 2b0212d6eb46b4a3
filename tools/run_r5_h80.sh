#!/bin/bash
# round 5: fused qkv + attention for head dim 80 -- end-to-end bit identity + same-box A/B on BASELINE configs[2] (ViTPose-H / wholebody, 128 crops)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_h80.txt
rm -f $O
timeout 900 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "head_dim_80" 2>&1 | tail -5 >> $O
for r in 1 2 3; do for f in 0 1; do
  echo -n "VP_FUSE_QKV_ATTN=$f: " >> $O
  VP_FUSE_QKV_ATTN=$f timeout 300 python bench.py --variant h --dataset wholebody --batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o=' '.join(f'{k[5:9] if k.startswith(\"gemm\") else k[:5]}={v[\"ms_per_step\"]:.3f}' for k,v in d['breakdown'].items())
        else: print(d['value'], d['ms_per_step'], o, d['encoder_gemms']['gemm_qkv']['kernel'])
" >> $O
done; done
cat $O
