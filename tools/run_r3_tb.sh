#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
AB=$PWD/easy_vitpose_amd/_lib/ab
rm -f gpurun_out/r3_tb_check.txt gpurun_out/r3_tb_bench.txt
for L in tb2 tb1; do
  echo "##### $L" >> gpurun_out/r3_tb_check.txt
  VP_HIP_LIB=$AB/$L.so timeout 300 python tools/gemm8_check.py --reps 4 >> gpurun_out/r3_tb_check.txt 2>&1
  VP_HIP_LIB=$AB/$L.so timeout 300 python tools/gemm8_check.py --variant h --batch 128 --reps 3 >> gpurun_out/r3_tb_check.txt 2>&1
  VP_HIP_LIB=$AB/$L.so timeout 300 python tools/gemm8_check.py --variant s --batch 256 --reps 3 --no-bench >> gpurun_out/r3_tb_check.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/r3_tb_check.txt
for r in 1 2 3; do for L in tb1 tb2; do
  echo -n "$L: " >> gpurun_out/r3_tb_bench.txt
  VP_HIP_LIB=$AB/$L.so timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o=' '.join(f'{k[5:9] if k.startswith(\"gemm\") else k[:5]}={v[\"ms_per_step\"]:.3f}' for k,v in d['breakdown'].items())
        else: print(d['value'], d['ms_per_step'], o)
" >> gpurun_out/r3_tb_bench.txt
done; done
cat gpurun_out/r3_tb_bench.txt
