#!/bin/bash
# round 5, GPU call 1: calibration probes + attn.proj start-stagger sweep (tools build, in situ) + reference bench of the product build
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_call1.txt
rm -f $O
timeout 600 python tools/r5_probes.py >> $O 2>&1
echo "== attn.proj start stagger, in situ (tools build): persons/s, ms/step, per-family ms/step" >> $O
for s in 0 4 8 12 16 24 32 1008 1016 2008 0; do
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
echo "== product build, reference bench" >> $O
timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/r5_bench_ref.json 2>> $O
cat gpurun_out/r5_bench_ref.json >> $O
cat $O
