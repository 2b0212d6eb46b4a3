#!/bin/bash
# A/B several values of one environment switch on the same box: tools/ab_multi.sh VAR "v1 v2 v3" [rounds]
V=$1; VALS=$2; R=${3:-2}
for r in $(seq $R); do for x in $VALS; do
  echo -n "$V=$x: "
  env $V=$x timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o=' '.join(f'{k[:9]}={v[\"ms_per_step\"]:.3f}' for k,v in d['breakdown'].items())
        else: print(d['value'], d['ms_per_step'], o)
"; done; done
