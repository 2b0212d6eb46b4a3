#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
AB=$PWD/easy_vitpose_amd/_lib/ab
python tools/resid_store_probe.py 2>&1 | grep -v amdgpu
for L in new old; do echo "## $L"; VP_HIP_LIB=$AB/$L.so timeout 300 python tools/gemm8_check.py --reps 4 2>&1 | grep -E "proj|fc2"; VP_HIP_LIB=$AB/$L.so timeout 300 python tools/gemm8_check.py --variant s --batch 256 --reps 3 --no-bench 2>&1 | grep -E "proj|fc2"; done
bash tools/run_ab.sh old new 3
