#!/bin/bash
# tile-boundary waits of the 16-bit epilogues (VP_G8_BND 0 / 1 / 2): identity, timeline, same-box A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/bnd.txt; rm -f $O
for v in b s h; do
  echo "== gemm8_check --variant $v" >> $O
  timeout 300 python tools/gemm8_check.py --variant $v --batch $([ $v = b ] && echo 256 || echo 128) --reps 3 >> $O 2>&1
done
echo "== timeline (tools build = BND 2)" >> $O
timeout 200 python tools/gemm8_timeline.py >> $O 2>&1
A=bnd0; B=bnd1; Cc=bnd2
for r in 1 2 3; do for L in bnd0 bnd1 bnd2; do
  echo -n "$L: " >> $O
  VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/ab/$L.so timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o=' '.join(f'{k[5:9] if k.startswith(\"gemm\") else k[:5]}={v[\"ms_per_step\"]:.3f}' for k,v in d['breakdown'].items())
        else: print(d['value'], d['ms_per_step'], o)
" >> $O
done; done
cat $O
