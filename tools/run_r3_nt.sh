#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
T=$PWD/easy_vitpose_amd/_lib/libvitpose_hip_tools.so
for r in 1 2; do
  for cfg in "none" "2:64" "1:64" "2:64,1:64"; do
    echo -n "nt stores [$cfg]: " >> gpurun_out/r3_nt.txt
    VP_HIP_LIB=$T VP_ABLATE_FAM=$cfg timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o=' '.join(f'{k[5:9] if k.startswith(\"gemm\") else k[:5]}={v[\"ms_per_step\"]:.3f}' for k,v in d['breakdown'].items())
        else: print(d['value'], d['ms_per_step'], o)
" >> gpurun_out/r3_nt.txt
  done
done
cat gpurun_out/r3_nt.txt
