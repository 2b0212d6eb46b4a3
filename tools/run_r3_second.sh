#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 200 python tools/fp8_probe_rows.py > gpurun_out/r3_fp8_rows.txt 2>&1; cat gpurun_out/r3_fp8_rows.txt
timeout 300 python tools/touch_ab.py > gpurun_out/r3_touch_ab.txt 2>&1; cat gpurun_out/r3_touch_ab.txt
T=$PWD/easy_vitpose_amd/_lib/libvitpose_hip_tools.so
for r in 1 2; do
  for cfg in "7 0" "7 1" "15 0" "15 1"; do
    set -- $cfg
    echo -n "GEMM8=$1 TOUCH=$2: " >> gpurun_out/r3_touch_bench.txt
    VP_HIP_LIB=$T VP_GEMM8=$1 VP_TOUCH=$2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o=' '.join(f'{k[5:9] if k.startswith(\"gemm\") else k[:5]}={v[\"ms_per_step\"]:.3f}' for k,v in d['breakdown'].items())
        else: print(d['value'], d['ms_per_step'], o)
" >> gpurun_out/r3_touch_bench.txt
  done
done
cat gpurun_out/r3_touch_bench.txt
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/r3_pytest2.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|\[fp8|outliers|ap10k" gpurun_out/r3_pytest2.log | tail -40
