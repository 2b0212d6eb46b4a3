#!/bin/bash
# round 3, first GPU call: full GPU suite, bench, product-vs-measurement-build A/B, item-4 measurements
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r3_pytest1.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3_pytest1.log
tail -5 gpurun_out/r3_pytest1.log
timeout 400 python bench.py > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err; echo "bench rc=$?"
cat gpurun_out/r3_bench1.json | head -c 1500
T=$PWD/easy_vitpose_amd/_lib/libvitpose_hip_tools.so
for r in 1 2; do
  for L in product tools; do
    if [ $L = tools ]; then export VP_HIP_LIB=$T; else unset VP_HIP_LIB; fi
    echo -n "$L: " >> gpurun_out/r3_ab_product_tools.txt
    timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o=' '.join(f'{k[5:9] if k.startswith(\"gemm\") else k[:5]}={v[\"ms_per_step\"]:.3f}' for k,v in d['breakdown'].items())
        else: print(d['value'], d['ms_per_step'], o)
" >> gpurun_out/r3_ab_product_tools.txt
  done
done
unset VP_HIP_LIB
cat gpurun_out/r3_ab_product_tools.txt
timeout 300 python tools/item4_compare.py > gpurun_out/r3_item4.txt 2>&1; cat gpurun_out/r3_item4.txt
