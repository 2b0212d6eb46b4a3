#!/bin/bash
# same-box A/B of product builds under easy_vitpose_amd/_lib/ab/ on any bench configuration: tools/run_ab_cfg.sh "<bench args>" rounds libA libB [libC ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
ARGS=$1; R=$2; shift 2
for r in $(seq $R); do for L in "$@"; do
  echo -n "$L: "
  VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/ab/$L.so timeout 300 python bench.py $ARGS --steps 30 --warmup 5 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
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
