#!/bin/bash
# same-box A/B of two product builds under easy_vitpose_amd/_lib/ab/: tools/run_ab.sh old new [rounds]
cd ${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2; R=${3:-3}
rm -f gpurun_out/ab_${A}_${B}.txt
for r in $(seq $R); do for L in $A $B; do
  echo -n "$L: " >> gpurun_out/ab_${A}_${B}.txt
  VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/ab/$L.so timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o=' '.join(f'{k[5:9] if k.startswith(\"gemm\") else k[:5]}={v[\"ms_per_step\"]:.3f}' for k,v in d['breakdown'].items())
        else: print(d['value'], d['ms_per_step'], o)
" >> gpurun_out/ab_${A}_${B}.txt
done; done
cat gpurun_out/ab_${A}_${B}.txt
