#!/bin/bash
# round 5, GPU call 2: attn.proj whole-CU start stagger + fc1 / qkv / residual timelines of the shipped kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_call2.txt
rm -f $O
echo "== attn.proj start stagger of whole CUs, in situ (tools build)" >> $O
for s in 0 3008 3016 3024 4008 4016 4024 5016 0; do
  echo -n "VP_PROJ_STAGGER=$s: " >> $O
  VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/libvitpose_hip_tools.so VP_PROJ_STAGGER=$s timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o=' '.join(f'{k[5:9] if k.startswith(\"gemm\") else k[:5]}={v[\"ms_per_step\"]:.3f}' for k,v in d['breakdown'].items())
        else: print(d['value'], d['ms_per_step'], o)
" >> $O
done
echo "== gemm8 timelines (tools build)" >> $O
timeout 300 python tools/gemm8_timeline.py >> $O 2>&1
timeout 300 python tools/gemm8_timeline.py --resid >> $O 2>&1
echo "== gemm8 timelines, K-tile sections (-DVP_G8_ABL=16 build)" >> $O
VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/ab/tl16.so timeout 300 python tools/gemm8_timeline.py >> $O 2>&1
cat $O
