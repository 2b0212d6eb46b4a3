#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
AB=$PWD/easy_vitpose_amd/_lib/ab
for L in new old; do echo "## $L"; VP_HIP_LIB=$AB/$L.so timeout 300 python tools/gemm8_check.py --reps 4 2>&1 | grep -E "compare|gemm8"; VP_HIP_LIB=$AB/$L.so timeout 300 python tools/gemm8_check.py --variant h --batch 128 --reps 3 2>&1 | grep -E "compare|gemm8"; done
bash tools/run_ab.sh old new 3
