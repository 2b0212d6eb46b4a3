#!/bin/bash
# A/B two builds of the library on the same box (box-to-box variance is ~5 %): tools/ab.sh libA.so libB.so [rounds]
# The libraries live under easy_vitpose_amd/_lib/ab/ (git-ignored, shipped by gpurun).
A=$1; B=$2; R=${3:-2}
for r in $(seq $R); do for L in $A $B; do
  echo -n "$(basename $L): "
  VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/ab/$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o=' '.join(f'{k[:9]}={v[\"ms_per_step\"]:.3f}' for k,v in d['breakdown'].items())
        else: print(d['value'], d['ms_per_step'], o)
"; done; done
