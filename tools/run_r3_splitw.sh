#!/bin/bash
# 192-wide tiles of the 8-phase kernel: W restage inside the MFMA section (VP_G8_SPLITW): identity + timing + same-box A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/splitw.txt; rm -f $O
for v in b s; do
  echo "== gemm8_check --variant $v" >> $O
  timeout 300 python tools/gemm8_check.py --variant $v --batch 256 --reps 3 >> $O 2>&1
done
timeout 100 python tools/gemm8_timeline.py --resid >> $O 2>&1
bash tools/run_ab.sh old new 3 >> $O 2>&1
cat $O
